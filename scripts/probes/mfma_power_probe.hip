// Probe (developer, standalone; round 4): what the matrix pipe sustains under the chip's power envelope.  Every SIMD of the chip runs nothing
// but v_mfma_f32_32x32x16_f16 (and, second case, v_mfma_scale_f32_32x32x64_f8f6f4 on e4m3) on register operands — no memory, no LDS —
// for a few milliseconds, with all-zero operands and with random ones.  The ratio is the envelope; the random-operand rate is the ceiling a
// real GEMM's matrix pipe has on this part, whatever its schedule.   hipcc --offload-arch=gfx950 -O2 mfma_power_probe.hip -o mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// KIND 6 / 7: the f16 32x32x16 stream with LDS fragment traffic beside it — 1.5 / 1.0 sixteen-byte reads per lane and MFMA (what a 128 x 64 / a 128 x 128
// wave tile of the projections reads), random data in LDS: how much of the envelope do the fragment reads take from the matrix pipe?
template <int KIND>
__global__ __launch_bounds__(256) void burn(const unsigned* __restrict__ seed, float* __restrict__ out, int iters, unsigned long long* clk) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    unsigned s = seed[tid & 4095];
    f16x8 a[4], b[4];
    i32x8 xa[2], xb[2];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u; a[i][e] = (_Float16)(seed[4096] ? ((float)(s >> 8) / 8388608.f - 1.f) : 0.f);
            s = s * 1664525u + 1013904223u; b[i][e] = (_Float16)(seed[4096] ? ((float)(s >> 8) / 8388608.f - 1.f) * 0.05f : 0.f);
        }
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u; xa[i][e] = seed[4096] ? (int)(s & 0x77777777u) : 0;      // (e4m3 bytes without the NaN pattern)
            s = s * 1664525u + 1013904223u; xb[i][e] = seed[4096] ? (int)(s & 0x77777777u) : 0;
        }
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    if constexpr (KIND >= 6) {
        for (int i = threadIdx.x; i < 65536 / 4; i += 256) { s = s * 1664525u + 1013904223u; reinterpret_cast<unsigned*>(lds)[i] = seed[4096] ? (s & 0x3BFF3BFFu) : 0u; }
        __syncthreads();
    }
    const unsigned lbase = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 16384;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (KIND >= 6) {
                constexpr int NR = KIND == 6 ? 6 : 4;        // reads per 4 MFMAs
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const f16x8 t = *reinterpret_cast<const f16x8*>(lds + ((lbase + (unsigned)(it * 4 + u) * 1024u * NR + r * 1024u) & 65535u & ~15u));
                    if (r < 4) a[r] = t; else b[r - 4] = t;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], c[i], 0, 0, 0);
            } else
            if constexpr (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], c[i], 0, 0, 0);
            } else if constexpr (KIND == 5) {        // block-scaled 16x16x128 on e4m3
                typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 t = {c[i][0], c[i][1], c[i][2], c[i][3]};
                    t = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xa[(i + u) & 1], xb[i & 1], t, 0, 0, 0, 100, 0, 100);
                    c[i][0] = t[0]; c[i][1] = t[1]; c[i][2] = t[2]; c[i][3] = t[3];
                }
            } else if constexpr (KIND == 4) {        // 16x16x32 f16: half the accumulator registers per MAC, twice the operand registers
                typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 t = {c[i][0], c[i][1], c[i][2], c[i][3]};
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[i], t, 0, 0, 0);
                    c[i][0] = t[0]; c[i][1] = t[1]; c[i][2] = t[2]; c[i][3] = t[3];
                }
            } else if constexpr (KIND == 2) {        // the same bits read as bf16
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + u) & 3]), __builtin_bit_cast(bf16x8, b[i]), c[i], 0, 0, 0);
            } else if constexpr (KIND == 3) {        // int8 (32x32x32): the fp8 test's bytes
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const i32x4 ia = {xa[(i + u) & 1][0], xa[(i + u) & 1][1], xa[(i + u) & 1][2], xa[(i + u) & 1][3]}, ib = {xb[i & 1][0], xb[i & 1][1], xb[i & 1][2], xb[i & 1][3]};
                    c[i] = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, __builtin_bit_cast(i32x16, c[i]), 0, 0, 0));
                }
            } else if constexpr (KIND == 8) {        // round 5: the block-scaled MFMA on e2m3 (fp6) operands — 6 of the 8 operand registers, half the cycles of e4m3
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[(i + u) & 1], xb[i & 1], c[i], 2, 2, 0, 100, 0, 100);
            } else if constexpr (KIND == 9) {        // ... and on e2m1 (fp4): 4 of the 8 operand registers
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[(i + u) & 1], xb[i & 1], c[i], 4, 4, 0, 100, 0, 100);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[(i + u) & 1], xb[i & 1], c[i], 0, 0, 0, 100, 0, 100);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc += c[i][e];
    out[tid] = acc;
    if (tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    int ncu = 256;
    hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
    const int blocks = ncu * 2, threads = blocks * 256;             // 8 waves per CU = 2 per SIMD
    unsigned* dseed; float* dout; unsigned long long* dclk;
    hipMalloc(&dseed, 4097 * 4); hipMalloc(&dout, threads * 4); hipMalloc(&dclk, 16);
    std::vector<unsigned> hs(4097);
    for (int i = 0; i < 4096; ++i) hs[i] = 12345u + 7919u * i;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 10; ++kind)
        for (int rnd = 0; rnd < 2; ++rnd) {
            hs[4096] = rnd;
            hipMemcpy(dseed, hs.data(), 4097 * 4, hipMemcpyHostToDevice);
            float best = 1e30f; unsigned long long hc[2] = {0, 0};
            for (int rep = 0; rep < 6; ++rep) {                     // back to back: the later repetitions are the sustained figure
                hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else if (kind == 1) hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else if (kind == 2) hipLaunchKernelGGL(burn<2>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else if (kind == 3) hipLaunchKernelGGL(burn<3>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else if (kind == 4) hipLaunchKernelGGL(burn<4>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else if (kind == 5) hipLaunchKernelGGL(burn<5>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else if (kind == 6) hipLaunchKernelGGL(burn<6>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else if (kind == 7) hipLaunchKernelGGL(burn<7>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else if (kind == 8) hipLaunchKernelGGL(burn<8>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                else hipLaunchKernelGGL(burn<9>, dim3(blocks), dim3(256), 0, 0, dseed, dout, iters, dclk);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 3 && ms < best) best = ms;
                hipMemcpy(hc, dclk, 16, hipMemcpyDeviceToHost);
            }
            const double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 /*mfma per iteration*/ * (kind == 0 || kind == 2 || kind == 6 || kind == 7 ? 2.0 * 32 * 32 * 16 : kind == 3 ? 2.0 * 32 * 32 * 32 : kind == 4 ? 2.0 * 16 * 16 * 32 : kind == 5 ? 2.0 * 16 * 16 * 128 : 2.0 * 32 * 32 * 64);
            printf("%s, %s operands: %.3f ms for %d x 16 MFMAs per wave, 2 waves per SIMD on %d CUs: %.0f TFLOP/s; shader clock %.0f MHz\n",
                   kind == 0 ? "v_mfma_f32_32x32x16_f16" : kind == 6 ? "v_mfma_f32_32x32x16_f16 + 1.5 LDS reads per MFMA" : kind == 7 ? "v_mfma_f32_32x32x16_f16 + 1.0 LDS read per MFMA" : kind == 2 ? "v_mfma_f32_32x32x16_bf16" : kind == 3 ? "v_mfma_i32_32x32x32_i8" : kind == 4 ? "v_mfma_f32_16x16x32_f16" : kind == 5 ? "v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)" : kind == 8 ? "v_mfma_scale_f32_32x32x64_f8f6f4 (e2m3)" : kind == 9 ? "v_mfma_scale_f32_32x32x64_f8f6f4 (e2m1)" : "v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)", rnd ? "random" : "zero", best, iters, ncu, flops / best / 1e9,
                   hc[1] ? (double)hc[0] * 100.0 / (double)hc[1] : 0.0);
        }
    return 0;
}
