// Developer probe (GPU, round 3): do the matrix pipe and the VALU of one SIMD work at the same time for two DIFFERENT waves?
// 8 waves per workgroup (waves w and w + 4 share a SIMD), one workgroup per CU: waves 0-3 run an MFMA loop, waves 4-7 a VALU loop
// (independent v_fma_f32 chains, or v_exp_f32).  Each role alone, then both together: "together ~ max" = the pipes overlap across
// waves, "together ~ sum" = they exclude each other.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/coexec_probe.hip -o /tmp/coexec_probe && /tmp/coexec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// mode bits: 1 = MFMA waves work, 2 = VALU waves work; mf: 0 = 32x32x16 f16, 1 = 32x32x64 fp8 scaled; vk: 0 = fma chains, 1 = exp
template <int MF, int VK>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mf_wave = wave < 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float res = 0.f;
    if (mf_wave) {
        if (mode & 1) {
            f16x8 a, b; i32x8 xa, xb;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); xa[i] = 0x38383838 + lane; xb[i] = 0x34343434 - lane; }
            f32x16 d0, d1, d2, d3;
            for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; d2[i] = 0; d3[i] = 0; }
            for (int it = 0; it < iters; ++it) {
                if (MF == 0) {
                    d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d3, 0, 0, 0);
                } else {
                    d0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, d0, 0, 0, 0, 127, 0, 127); d1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, d1, 0, 0, 0, 127, 0, 127);
                }
            }
            for (int i = 0; i < 16; ++i) res += d0[i] + d1[i] + d2[i] + d3[i];
        }
    } else {
        if (mode & 2) {
            float x0 = 1.0f + lane * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
            const float m = 0.999f, c = 1e-4f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (VK == 0) {
                        x0 = __builtin_fmaf(x0, m, c); x1 = __builtin_fmaf(x1, m, c); x2 = __builtin_fmaf(x2, m, c); x3 = __builtin_fmaf(x3, m, c);
                        x4 = __builtin_fmaf(x4, m, c); x5 = __builtin_fmaf(x5, m, c); x6 = __builtin_fmaf(x6, m, c); x7 = __builtin_fmaf(x7, m, c);
                    } else {
                        x0 = __builtin_amdgcn_exp2f(x0 * m); x1 = __builtin_amdgcn_exp2f(x1 * m); x2 = __builtin_amdgcn_exp2f(x2 * m); x3 = __builtin_amdgcn_exp2f(x3 * m);
                    }
                }
            }
            res = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (res == 12345.678f) out[threadIdx.x] = res;
}

template <int MF, int VK> static void run(const char* what, float* out, unsigned long long* cyc, int iters) {
    unsigned long long h[256 * 8];
    double r[4][2];
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MF, VK>), dim3(256), dim3(512), 0, 0, out, cyc, iters, mode); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int g = 0; g < 256; ++g) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += (double)h[g * 8 + w];
        r[mode][0] = a / (256 * 4) / iters; r[mode][1] = b / (256 * 4) / iters;
    }
    printf("%-46s MFMA wave alone %7.1f ticks/iter | VALU wave alone %7.1f | together: MFMA wave %7.1f, VALU wave %7.1f\n", what, r[1][0], r[2][1], r[3][0], r[3][1]);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc((void**)&out, 4096); (void)hipMalloc((void**)&cyc, 256 * 8 * 8);
    const int iters = 20000;
    printf("per iteration: MFMA wave = 4 x 32x32x16 f16 (or 2 x 32x32x64 fp8) ; VALU wave = 32 v_fma_f32 (or 16 v_mul + 16 v_exp); s_memtime ticks\n");
    run<0, 0>("f16 MFMAs  |  fma chains", out, cyc, iters);
    run<1, 0>("fp8 scaled MFMAs  |  fma chains", out, cyc, iters);
    run<0, 1>("f16 MFMAs  |  mul + exp", out, cyc, iters);
    return 0;
}
