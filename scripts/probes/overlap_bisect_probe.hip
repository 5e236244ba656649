// Developer probe (GPU, round 6): from the clean MFMA / VALU overlap loop of mfma_valu_overlap_probe.hip towards the band attention kernel, ONE
// construct at a time — where does a wave stop overlapping its matrix instructions with its vector instructions?
// A "step" = 8 MFMAs (4 accumulators) + 64 VALU instructions, program order pinned by sched_barrier(0) after every group; MFMAs and VALU are compiler
// builtins so that hipcc's hazard padding (s_nop) is the one a real kernel gets.  s_memtime ticks per step and wave, one wave per SIMD (256-thread
// workgroups, one per CU) and two (512-thread).  Modes:
//   M        MFMAs only                                   V        VALU only (v_fma_f32)
//   MV       1 MFMA : 8 VALU interleaved, all independent (the round-5 probe's "both in one stream")
//   BUNCH    8 MFMAs back to back, then the 64 VALU (what hipcc emits for a product: a burst of 4-6 MFMAs)
//   DEPT     interleaved; the VALU of step t read the accumulators written in step t - 1 (two accumulator sets: software pipelining by one tile)
//   DEP1/2/4 interleaved; VALU group k reads the accumulator written by MFMA k - 1 / k - 2 / k - 4 of the same step
//   PROD     interleaved; the B operand of MFMA k is written by VALU group k - 1 (v_cvt_pk_f16: P -> P.V)
//   SW_MIX   MFMAs only, per accumulator f16 f16 f16 f16 fp8 fp8 (mm_lh_hl of attention_mx.hip: the type switches on ONE accumulator)
//   SW_SEP   MFMAs only, the f16 MFMAs of all accumulators first, then the scaled ones (same instructions)
//   LDS      MV + one ds_read_b128 per MFMA gap, waited for (lgkmcnt) in the next gap
//   CVTS     MV with v_cvt_scalef32_pk_fp8_f32 as the VALU      FMIX     MV with v_fma_mix_f32 op_sel
//   AGPR     MV, accumulators in the accumulation registers ("a" constraints)
// Built with -mllvm -amdgpu-mfma-vgpr-form so that the accumulators are architectural registers at BOTH occupancies (without it hipcc moves them to
// AGPRs as soon as the kernel may use 512 registers); the AGPR mode asks for them explicitly.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form scripts/probes/overlap_bisect_probe.hip -o /tmp/overlap_bisect_probe && /tmp/overlap_bisect_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

enum { M_ONLY, V_ONLY, MV, BUNCH, DEPT, DEP1, DEP2, DEP4, PROD, SW_MIX, SW_SEP, LDS_MV, CVTS, FMIX, AGPR, LDS_M, NMODES };
static const char* NAMES[NMODES] = {"M", "V", "MV", "BUNCH", "DEPT", "DEP1", "DEP2", "DEP4", "PROD", "SW_MIX", "SW_SEP", "LDS", "CVTS", "FMIX", "AGPR", "LDS_M"};

// SB: nothing crosses in the machine scheduler; FX: the VALU values pass through an empty asm, so no IR pass merges, re-associates or moves the groups
#define SB() __builtin_amdgcn_sched_barrier(0)
#define FX() asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(wacc))

template <int MODE, int NT>
__global__ __launch_bounds__(NT, 1) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a[2], b[4];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 8; ++i) a[j][i] = (_Float16)(0.001f * (lane + i + j));
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) b[j][i] = (_Float16)(0.002f * (lane - i + j));
    f32x16 acc[2][4];
    for (int s = 0; s < 2; ++s) for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[s][j][i] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + 1e-3f * (lane + i);
    i32x8 ax, bx;
    for (int i = 0; i < 8; ++i) { ax[i] = 0x38383838 + lane + i; bx[i] = 0x30303030 + lane * 3 + i; }
    int sc = 127;
    asm volatile("" : "+v"(sc));
    float cA = 0.999f, cB = 1e-4f, one_f = 1.0f;
    asm volatile("" : "+v"(cA), "+v"(cB), "+s"(one_f));
    for (int i = threadIdx.x; i < 4096; i += NT) reinterpret_cast<int*>(lds)[i] = i;
    i32x4 lq = {0, 0, 0, 0};
    int wacc = lane;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {         // (two steps per trip: DEPT's accumulator set is a compile-time index)
        const int cur = (MODE == DEPT) ? half : 0;
        auto valu8 = [&](int g, const f32x16* src) __attribute__((always_inline)) {      // 8 VALU instructions; src: an accumulator they read (or nullptr)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MODE == CVTS) {
                    typedef short v2i16 __attribute__((ext_vector_type(2)));
                    v2i16 w2 = __builtin_bit_cast(v2i16, wacc);
                    if (i & 1) w2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w2, x[i], x[(i + 1) & 7], 0.5f, true);
                    else w2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w2, x[i], x[(i + 1) & 7], 0.5f, false);
                    wacc = __builtin_bit_cast(int, w2);
                } else if constexpr (MODE == FMIX) {
                    asm volatile("v_fma_mix_f32 %0, %0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(x[i]) : "s"(one_f), "v"(wacc));
                } else if (src) {
                    x[i] = __builtin_fmaf((*src)[(8 * g + i) & 15], cA, x[i]);
                } else {
                    x[i] = __builtin_fmaf(x[i], cA, cB);
                }
            }
        };
        auto f16mm = [&](int j, int s, int u) __attribute__((always_inline)) {
            if constexpr (MODE == AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[s][j]) : "v"(a[u & 1]), "v"(b[j]));
            else acc[s][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 1], b[j], acc[s][j], 0, 0, 0);
        };
        auto f8mm = [&](int j, int s) __attribute__((always_inline)) {
            acc[s][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ax, bx, acc[s][j], 0, 0, 0, sc, 0, sc);
        };
        if constexpr (MODE == SW_MIX) {            // 4 accumulators x (4 f16 + 2 fp8): 1024 matrix-pipe cycles
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { f16mm(j, 0, u); SB(); }
                f8mm(j, 0); SB(); f8mm(j, 0); SB();
            }
        } else if constexpr (MODE == SW_SEP) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) { f16mm(j, 0, u); SB(); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { f8mm(j, 0); SB(); f8mm(j, 0); SB(); }
        } else if constexpr (MODE == BUNCH) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { f16mm(u & 3, 0, u); SB(); }
#pragma unroll
            for (int u = 0; u < 8; ++u) { valu8(u, nullptr); FX(); SB(); }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (MODE != V_ONLY) {
                    if constexpr (MODE == PROD) {      // B operand of this MFMA = the previous group's VALU results
                        f16x8 p;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f16x2 h2 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x[2 * i], x[2 * i + 1]));
                            p[2 * i] = h2[0]; p[2 * i + 1] = h2[1];
                        }
                        acc[0][u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 1], p, acc[0][u & 3], 0, 0, 0);
                    } else f16mm(u & 3, cur, u);
                }
                if constexpr (MODE == LDS_MV || MODE == LDS_M) {
                    // the read issued in the previous gap is consumed here (its wait), the next one is issued
                    x[u & 7] += __builtin_bit_cast(float, lq[0]);
                    lq = *reinterpret_cast<const i32x4*>(lds + ((lane * 16 + u * 1024 + it * 64) & 16383 & ~15));
                }
                SB();
                if constexpr (MODE != M_ONLY && MODE != LDS_M) {
                    const f32x16* src = nullptr;
                    if constexpr (MODE == DEPT) src = &acc[cur ^ 1][u & 3];
                    if constexpr (MODE == DEP1) src = &acc[0][(u + 3) & 3];       // written by MFMA u - 1 (previous step's last for u = 0)
                    if constexpr (MODE == DEP2) src = &acc[0][(u + 2) & 3];
                    if constexpr (MODE == DEP4) src = &acc[0][u & 3];             // MFMA u - 4 wrote it ... and MFMA u (just issued) rewrites it: WAR + RAW
                    valu8(u, src);
                    FX();
                    SB();
                }
            }
        }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = (float)wacc + __builtin_bit_cast(float, lq[1]);
    for (int i = 0; i < 8; ++i) r += x[i];
    for (int s = 0; s < 2; ++s) for (int j = 0; j < 4; ++j) r += acc[s][j][0] + acc[s][j][7];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE, int NT> static double run1(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    (void)hipMemset(cyc, 0, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
    unsigned long long h[2048];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; int n = 0;
    for (int i = 0; i < 2048; ++i) if (h[i]) { s += (double)h[i]; ++n; }
    return n ? s / n / iters : 0.0;
}
template <int MODE> static void row(float* out, unsigned long long* cyc, const char* note) {
    const double w1 = run1<MODE, 256>(out, cyc), w2 = run1<MODE, 512>(out, cyc);
    printf("%-7s one wave per SIMD %7.1f | two waves per SIMD %7.1f (= %6.1f per wave's work)   %s\n", NAMES[MODE], w1, w2, w2 / 2, note);
    fflush(stdout);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 256 * 8 * 8);
    printf("s_memtime ticks per step (8 MFMAs 32x32x16 f16 on 4 accumulators + 64 VALU) and wave\n");
    row<M_ONLY>(out, cyc, "8 MFMAs alone: 256 matrix-pipe cycles");
    row<V_ONLY>(out, cyc, "64 v_fma_f32 alone");
    row<MV>(out, cyc, "1 MFMA : 8 v_fma, independent");
    row<BUNCH>(out, cyc, "8 MFMAs back to back, then 64 v_fma");
    row<DEPT>(out, cyc, "v_fma read the accumulators of the previous step (two accumulator sets)");
    row<DEP4>(out, cyc, "v_fma group k reads the accumulator MFMA k - 4 wrote (and MFMA k rewrites)");
    row<DEP2>(out, cyc, "v_fma group k reads the accumulator MFMA k - 2 wrote");
    row<DEP1>(out, cyc, "v_fma group k reads the accumulator MFMA k - 1 wrote");
    row<PROD>(out, cyc, "MFMA k's B operand = v_cvt_pkrtz of group k - 1's results");
    row<LDS_M>(out, cyc, "8 MFMAs + one ds_read_b128 per gap, consumed in the next gap");
    row<LDS_MV>(out, cyc, "MV + one ds_read_b128 per gap, consumed in the next gap");
    row<CVTS>(out, cyc, "MV with v_cvt_scalef32_pk_fp8_f32");
    row<FMIX>(out, cyc, "MV with v_fma_mix_f32 op_sel");
    row<AGPR>(out, cyc, "MV, accumulators in AGPRs");
    printf("type switches on one accumulator (MFMAs only; 16 x 32x32x16 f16 + 8 x scaled 32x32x64 = 1024 matrix-pipe cycles per step)\n");
    row<SW_MIX>(out, cyc, "per accumulator: f16 f16 f16 f16 fp8 fp8");
    row<SW_SEP>(out, cyc, "all f16 chains first, then all fp8 chains");
    return 0;
}
