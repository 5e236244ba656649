// Developer probe (GPU, round 5): does ONE wave overlap its own MFMAs with its own VALU instructions, and what do TWO waves of a SIMD gain?
// Per loop iteration a wave issues NM v_mfma_f32_32x32x16_f16 (four independent accumulators, dependent back to back inside an accumulator) and / or NV
// independent VALU instructions (v_fma_f32 or v_exp_f32), interleaved one MFMA : NV / NM VALU in program order (asm volatile keeps the order).
// Modes: M = MFMAs only, V = VALU only, MV = both in one instruction stream.  Launched with 4 waves per CU (one per SIMD) and with 8 (two per SIMD; in
// the "split" run wave k issues only MFMAs and wave k + 4 only VALU — the role split of attention_mxs.hip).  Cycles per iteration from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_overlap_probe.hip -o scripts/probes/mfma_valu_overlap_probe && scripts/probes/mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// MODE: 1 = MFMA only, 2 = VALU only, 3 = both interleaved, 4 = role split (waves 0-3 MFMA only, waves 4-7 VALU only; needs 8 waves)
template <int MODE, int TRANS, int MK = 0>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + 1e-3f * (lane + i);
    i32x8 ax, bx; int w[8]; f32x2 p2[8];
    for (int i = 0; i < 8; ++i) { ax[i] = 0x38383838 + lane + i; bx[i] = 0x30303030 + lane * 3 + i; w[i] = lane + i; p2[i] = (f32x2){x[i], x[i] + 1.f}; }
    const int sc = 127;
    const bool do_m = MODE == 1 || MODE == 3 || (MODE == 4 && wave < 4);
    const bool do_v = MODE == 2 || MODE == 3 || (MODE == 4 && wave >= 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {          // 8 MFMAs and 64 VALU instructions per iteration: 1 : 8
            if (do_m) {
                if (MK == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3" : "+v"(acc[u & 3]) : "v"(ax), "v"(bx), "v"(sc));
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (TRANS == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                    else if (TRANS == 2) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(w[i]) : "v"(x[i]), "v"(x[(i + 1) & 7]));
                    else if (TRANS == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2[i]) : "v"(p2[(i + 1) & 7]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(0.999f), "v"(1e-4f));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += x[i] + (float)w[i] + p2[i][0];
    for (int j = 0; j < 4; ++j) r += acc[j][0];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE, int TRANS, int MK = 0> static double run(int waves, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    (void)hipMemset(cyc, 0, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, TRANS, MK>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    unsigned long long h[2048];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int i = 0; i < 2048; ++i) if (h[i]) { s += (double)h[i]; ++n; }
    return s / n / iters;
}

template <int TR, int MK> static void table(const char* vn, const char* mn, float* out, unsigned long long* cyc) {
    printf("== per iteration: 8 x %s (4 accumulators) and / or 64 x %s; s_memtime ticks per iteration and wave ==\n", mn, vn);
    const double m1 = run<1, TR, MK>(4, out, cyc), v1 = run<2, TR, MK>(4, out, cyc), b1 = run<3, TR, MK>(4, out, cyc);
    printf("one wave per SIMD:   MFMA only %7.1f | VALU only %7.1f | both in one stream %7.1f   (sum %7.1f, max %7.1f)\n", m1, v1, b1, m1 + v1, m1 > v1 ? m1 : v1);
    const double m2 = run<1, TR, MK>(8, out, cyc), v2 = run<2, TR, MK>(8, out, cyc), b2 = run<3, TR, MK>(8, out, cyc), s2 = run<4, TR, MK>(8, out, cyc);
    printf("two waves per SIMD:  MFMA only %7.1f | VALU only %7.1f | both, each wave both %7.1f (= %7.1f per wave's work) | role split (one wave MFMA, one VALU) %7.1f\n", m2, v2, b2, b2 / 2, s2);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 256 * 8 * 8);
    table<0, 0>("v_fma_f32", "v_mfma_f32_32x32x16_f16", out, cyc);
    table<1, 0>("v_exp_f32", "v_mfma_f32_32x32x16_f16", out, cyc);
    table<2, 0>("v_cvt_pk_fp8_f32", "v_mfma_f32_32x32x16_f16", out, cyc);
    table<3, 0>("v_pk_add_f32", "v_mfma_f32_32x32x16_f16", out, cyc);
    table<0, 1>("v_fma_f32", "v_mfma_scale_f32_32x32x64_f8f6f4", out, cyc);
    table<1, 1>("v_exp_f32", "v_mfma_scale_f32_32x32x64_f8f6f4", out, cyc);
    table<2, 1>("v_cvt_pk_fp8_f32", "v_mfma_scale_f32_32x32x64_f8f6f4", out, cyc);
    return 0;
}
