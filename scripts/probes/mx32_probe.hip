// Developer probe (GPU, round 3): the 32x32x64 block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, fp8 e4m3 operands) as the carrier of
// the split products' cross terms: operand layout, per-lane e8m0 scales, issue rate and the clock the chip holds, next to the f16 forms.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mx32_probe.hip -o /tmp/mx32_probe && /tmp/mx32_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__global__ void k_once(const uint8_t* A, const uint8_t* B, float* C, const int* sa, const int* sb) {
    const int lane = threadIdx.x;
    i32x8 a = *reinterpret_cast<const i32x8*>(A + lane * 32);
    i32x8 b = *reinterpret_cast<const i32x8*>(B + lane * 32);
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
    for (int r = 0; r < 16; ++r) C[lane * 16 + r] = c[r];
}

// MODE 0: 16x16x32 f16 x4 accs; 1: 32x32x16 f16 x2 accs; 2: 32x32x64 fp8 scaled x2 accs; 3: per iteration 4 x 32x32x16 f16 + 2 x 32x32x64 (the MX product mix)
// 4: 6 x 32x32x16 (the 3-MFMA split mix at equal product count)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)(rnd() & 0x7f7f7f7f) ^ (int)(rnd() & 0x80808080); b[i] = (int)(rnd() & 0x7f7f7f7f) ^ (int)(rnd() & 0x80808080); }
    for (int i = 0; i < 8; ++i) { a[i] &= ~0x40404040; b[i] &= ~0x40404040; }      // keep |x| small: no inf / nan
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(((int)(rnd() >> 8) % 2001 - 1000) * 0.001f); hb[i] = (_Float16)(((int)(rnd() >> 8) % 2001 - 1000) * 0.001f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c3, 0, 0, 0);
        } else if (MODE == 1) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, d1, 0, 0, 0);
        } else if (MODE == 2) {
            d0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d0, 0, 0, 0, 0x7f, 0, 0x75);
            d1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d1, 0, 0, 0, 0x7f, 0, 0x75);
        } else if (MODE == 3) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d0, 0, 0, 0, 0x7f, 0, 0x75);
            d1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d1, 0, 0, 0, 0x7f, 0, 0x75);
        } else {
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ha, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, hb, d1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[5] + d1[7];
    if (threadIdx.x == 0 && blockIdx.x == 17) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

static const float kFp8[8] = {0.f, 1.f, 2.f, 3.f, 4.f, 0.5f, 1.5f, -2.f};
static uint8_t fp8_of(int v) { static const uint8_t t[8] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x30, 0x3c, 0xc0}; return t[v]; }

int main() {
    int Am[32][64], Bm[64][32];
    srand(11);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) Am[i][k] = rand() % 8;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) Bm[k][j] = rand() % 8;
    uint8_t hA[64 * 32], hB[64 * 32];
    uint8_t *dA, *dB; float* dC; int *dsa, *dsb;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 64 * 16 * sizeof(float)); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    for (int hyp = 0; hyp < 2; ++hyp) {
        // H0: lane l = (row l & 31, half l >> 5), byte y <-> k = 32 * half + y;   H1: k = 16 * half + (y & 15) + 32 * (y >> 4)
        for (int l = 0; l < 64; ++l) for (int y = 0; y < 32; ++y) {
            const int r = l & 31, h = l >> 5;
            const int k = hyp == 0 ? 32 * h + y : 16 * h + (y & 15) + 32 * (y >> 4);
            hA[l * 32 + y] = fp8_of(Am[r][k]);
            hB[l * 32 + y] = fp8_of(Bm[k][r]);
        }
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        for (int sc = 0; sc < 3; ++sc) {
            // sc 0: all scales 1; sc 1: A lanes of half 1 carry 2^-3; sc 2: B lanes of half 0 carry 2^2
            int sa[64], sb[64];
            for (int l = 0; l < 64; ++l) { sa[l] = (sc == 1 && l >= 32) ? 127 - 3 : 127; sb[l] = (sc == 2 && l < 32) ? 129 : 127; }
            hipMemcpy(dsa, sa, sizeof sa, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, sizeof sb, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
            float hC[64 * 16]; hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
            int bad = 0; double maxd = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {     // C/D: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
                double want = 0;
                for (int k = 0; k < 64; ++k) {
                    double w = (double)kFp8[Am[row][k]] * kFp8[Bm[k][col]];
                    if (sc == 1 && k >= 32) w *= 0.125;
                    if (sc == 2 && k < 32) w *= 4.0;
                    want += w;
                }
                const double d = fabs(hC[l * 16 + r] - want); if (d > 1e-3) ++bad; if (d > maxd) maxd = d;
            }
            printf("32x32x64 fp8: layout hypothesis %d, scale case %d: %d / 1024 wrong, max diff %.3f\n", hyp, sc, bad, maxd);
        }
    }
    float* dOut; unsigned long long* dCyc; hipMalloc(&dOut, 1024 * 256 * sizeof(float)); hipMalloc(&dCyc, 16);
    const char* names[5] = {"16x16x32 f16 (x4 accs)", "32x32x16 f16 (x2 accs)", "32x32x64 fp8 scaled (x2 accs)", "mix: 4 x 32x32x16 f16 + 2 x 32x32x64 fp8 (two MX products of 32x32x32... per iteration)", "mix: 6 x 32x32x16 f16 (two 3-MFMA products per iteration)"};
    const int per_it[5] = {4, 2, 2, 6, 6};
    for (int wps = 1; wps <= 2; ++wps)           // waves per SIMD
    for (int mode = 0; mode < 5; ++mode) {
        const int iters = 200000;
        unsigned long long cyc[2] = {0, 0};
        float ms = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            const dim3 grid(256 * wps), blk(256);
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, grid, blk, 0, 0, dOut, dCyc, iters, 1234u + rep);
            else if (mode == 1) hipLaunchKernelGGL(k_rate<1>, grid, blk, 0, 0, dOut, dCyc, iters, 1234u + rep);
            else if (mode == 2) hipLaunchKernelGGL(k_rate<2>, grid, blk, 0, 0, dOut, dCyc, iters, 1234u + rep);
            else if (mode == 3) hipLaunchKernelGGL(k_rate<3>, grid, blk, 0, 0, dOut, dCyc, iters, 1234u + rep);
            else hipLaunchKernelGGL(k_rate<4>, grid, blk, 0, 0, dOut, dCyc, iters, 1234u + rep);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(cyc, dCyc, 16, hipMemcpyDeviceToHost);
        const double per = (double)cyc[0] / ((double)per_it[mode] * iters);
        printf("%d wave(s)/SIMD  %-60s %6.1f ticks per MFMA per wave, in-kernel clock %.0f MHz, %.2f ms per launch (%.2f us per iteration per SIMD)\n", wps, names[mode], per,
               cyc[1] ? (double)cyc[0] / cyc[1] * 100.0 : 0.0, ms, ms * 1e3 / iters);
    }
    return 0;
}
