// 256x256x64 MFMA GEMM for the 16-bit encoder projections: C = A * W^T (+ fused epilogue).
//
// Same contract as gemm.hip (A [Mpad,K], W [N,K] row-major, fp32 accumulate, epilogues
// BIAS / GELU / RESID / QKV) but built for the large shapes of the hot path (M = B*Sp >= 256):
//   * 512 threads = 8 waves as 2(M) x 4(N); each wave owns 128x64 = 8x4 MFMA 16x16x32 tiles
//     (128 accumulator VGPRs), so one fragment read feeds 4-8 MFMAs;
//   * both operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no
//     ds_write); one wave instruction lands 8 rows x 128 B.  The LDS image is unpadded [256][128 B];
//     bank conflicts are removed by an XOR swizzle of the 16-B chunk index, chunk ^= (row>>1)&7,
//     applied to the per-lane SOURCE address (the DMA destination is lane-linear) and to the
//     ds_read_b128 address — the same involution on both sides;
//   * two 64 KiB stages: the DMA of K-tile t+1 is in flight while the MFMAs of tile t run; its 8 issue
//     slots per wave are spread between the MFMA groups (a burst at the loop top starves the matrix pipe:
//     +1..7 % measured); one s_waitcnt vmcnt(0) + s_barrier per K-tile.  Tried and rejected on MI355X
//     (same-box A/B): a 4-stage BK=32 ring with counted vmcnt (-10 %, barrier per 32-deep step), a 128x256
//     tile with two resident workgroups (-10..-25 %, 1.5x DMA bytes per FLOP), a software L2 prefetch (-8 %);
//   * workgroup ids are remapped so that the tiles of one XCD (blockIdx % 8) are consecutive in
//     (m, n) order: the n-tiles of an A panel share one L2.
#include <stdlib.h>
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int ROWB = 128;                  // bytes of K per row per stage (64 x 16-bit)
constexpr int STAGE = (TM + TN) * ROWB;    // 64 KiB
constexpr int EPI_PATCH = 9216;            // bytes of wave-private fp32 epilogue staging (8 x 9 KiB < 2 stages)
extern __shared__ __attribute__((aligned(16))) unsigned char smem256[];

__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}

template <typename T, int EPI, bool VMODE>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p, int n_tile0, int ntn) {
    typedef typename Frag<T>::type frag_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15, g = lane >> 4;
    const int K = p.K, N = p.N;

    // XCD-aware tile order (bijective for any grid size)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int mt = tile / ntn, nt = tile % ntn;
    const int m0 = mt * TM, n0 = (n_tile0 + nt) * TN;   // this launch covers n-tiles [n_tile0, n_tile0 + ntn)

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W);

    // DMA map: wave w moves rows [32w + 8i, +8) of the A tile and of the W tile, i = 0..3.
    // lane L lands at (row 8i + (L>>3), chunk L&7) and therefore FETCHES chunk (L&7) ^ ((row>>1)&7).
    const int lrow = lane >> 3, lch = lane & 7;
    const T* ga[4];
    const T* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + lrow;
        const int ch = lch ^ ((row >> 1) & 7);
        ga[i] = A + (size_t)(m0 + row) * K + ch * 8;
        gw[i] = W + (size_t)(n0 + row) * K + ch * 8;
    }
    auto stage = [&](int kt, int buf) {
        unsigned char* sa = smem256 + buf * STAGE + (wave * 32) * ROWB;
        unsigned char* sw = sa + TM * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(ga[i] + (size_t)kt * 64, sa + i * 8 * ROWB);
            glds16(gw[i] + (size_t)kt * 64, sw + i * 8 * ROWB);
        }
    };
    // one DMA piece (q = 0..7) of the stage: lets the main loop spread the 8 issues between its MFMA groups
    auto stage_piece = [&](int kt, int buf, int q) {
        unsigned char* sa = smem256 + buf * STAGE + (wave * 32) * ROWB;
        if (q < 4) glds16(ga[q] + (size_t)kt * 64, sa + q * 8 * ROWB);
        else glds16(gw[q - 4] + (size_t)kt * 64, sa + TM * ROWB + (q - 4) * 8 * ROWB);
    };
    const bool spread = p.spread_dma != 0;

    constexpr bool vmode = VMODE;          // V third of the fused QKV projection: transposed output
    f32x4 acc[8][4];      // [mi][ni] (lane = m) or, in vmode, the same slots with lane = n
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read addresses: row r, logical chunk 4*ks + g  ->  physical chunk ^ ((r>>1)&7)
    const int arow = wm * 128 + r16, brow = wn * 64 + r16;
    const int asw = (arow >> 1) & 7, bsw = (brow >> 1) & 7;   // (row + 16*i) keeps (row>>1)&7
    const int nk = K / 64;

    // diagnostic stamps (p.stamps != nullptr only in glc_debug_gemm_bench's stamp launch; never in the product path)
    const bool stamp = p.stamps != nullptr;
    unsigned long long t_cmp = 0, t_dma = 0, t_bar = 0, t_a = 0, t_b = 0, t_c = 0, t_begin = 0;
    auto now = [&]() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    if (stamp) t_begin = now();

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (stamp) t_a = now();
        const bool more = kt + 1 < nk;
        if (more && !spread) stage(kt + 1, buf ^ 1);
        const unsigned char* sa = smem256 + buf * STAGE + arow * ROWB;
        const unsigned char* sw = smem256 + buf * STAGE + TM * ROWB + brow * ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t bf[4];
            const int bch = ((4 * ks + g) ^ bsw) * 16, ach = ((4 * ks + g) ^ asw) * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const frag_t*>(sw + j * 16 * ROWB + bch);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const frag_t af = *reinterpret_cast<const frag_t*>(sa + i * 16 * ROWB + ach);
                if (more && ((p.spread_dma == 1 && (i & 1) == 0) || (p.spread_dma == 2 && ks == 0))) {
                    // spread_dma 1: one DMA issue per 8 MFMAs over the whole K-tile; 2: one per 4 MFMAs over its FIRST half,
                    // so that the youngest piece still has half a K-tile of MFMAs to land behind
                    stage_piece(kt + 1, buf ^ 1, p.spread_dma == 1 ? ks * 4 + (i >> 1) : i);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!vmode) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(bf[j], af, acc[i][j]);   // D[n = 4g+r][m = r16]
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(af, bf[j], acc[i][j]);   // D[m = 4g+r][n = r16]
                }
            }
        }
        if (stamp) { t_b = now(); t_cmp += t_b - t_a; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my DMA pieces of tile kt+1 have landed
        if (stamp) { t_c = now(); t_dma += t_c - t_b; }
        __builtin_amdgcn_s_barrier();                        // everyone's have; everyone is done reading `buf`
        if (stamp) { t_bar += now() - t_c; }
    }
    if (stamp && blockIdx.x < 64 && lane == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = t_cmp; o[1] = t_dma; o[2] = t_bar; o[3] = now() - t_begin;
    }

    // ---------------- epilogue ----------------
    // The accumulator layout gives a lane 4 consecutive columns of one row (8-byte pieces).  Storing that
    // directly costs 32 narrow, scattered stores per lane — as much time as a K=768 main loop.  Instead each
    // wave stages its 128x64 sub-tile through a PRIVATE fp32 LDS patch, 32 rows at a time (the stage ring
    // is dead after the last barrier), and writes it back as 16-byte pieces: one wave instruction = 8 rows
    // x 128 contiguous bytes (row-major outputs) or whole 16-byte fragment units (Q / K / V^T).  Bias and
    // GELU are applied before staging, the residual is added in fp32 at the store, so the value is rounded
    // once, exactly as before.  Wave-local LDS ordering only; no workgroup barrier.
    typedef __attribute__((ext_vector_type(8))) T vec8T;
    const float* __restrict__ bias = p.bias;
    float* stg = reinterpret_cast<float*>(smem256 + wave * EPI_PATCH);
    const int qkv_b0 = (EPI == EPI_QKV) ? m0 / p.Sp : 0;
    if (!vmode) {
        // D[n = 16j + 4g + r][m = 16i + r16]; patch [32 rows m][64 cols n], row stride 68 floats
        const int which = (EPI == EPI_QKV) ? n0 / p.H : 0;
        float bj[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (bias) { const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0 + wn * 64 + j * 16 + 4 * g); bj[j][0] = bv[0]; bj[j][1] = bv[1]; bj[j][2] = bv[2]; bj[j][3] = bv[3]; }
            else { bj[j][0] = bj[j][1] = bj[j][2] = bj[j][3] = 0.f; }
        }
        // residual rows are fetched one 32-row chunk AHEAD of their use (16-byte coalesced loads): without this each
        // chunk exposed a full HBM round trip between its LDS read-back and its store (+4.7 us per tile measured)
        vec8T rpre[4];
        auto load_resid = [&](int c, vec8T (&r)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, row = idx >> 3, g8 = idx & 7;
                r[k] = *reinterpret_cast<const vec8T*>(reinterpret_cast<const T*>(p.resid) +
                                                       (size_t)(m0 + wm * 128 + c * 32 + row) * N + n0 + wn * 64 + g8 * 8);
            }
        };
        if (EPI == EPI_RESID) load_resid(0, rpre);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            vec8T rcur[4];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int k = 0; k < 4; ++k) rcur[k] = rpre[k];
                if (c + 1 < 4) load_resid(c + 1, rpre);
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[2 * c + ii][j];
                    v[0] += bj[j][0]; v[1] += bj[j][1]; v[2] += bj[j][2]; v[3] += bj[j][3];
                    if (EPI == EPI_GELU) {
                        const f32x2 g0 = glc_gelu2((f32x2){v[0], v[1]}), g1 = glc_gelu2((f32x2){v[2], v[3]});
                        v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
                    }
                    *reinterpret_cast<f32x4*>(stg + (ii * 16 + r16) * 68 + j * 16 + 4 * g) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, row = idx >> 3, g8 = idx & 7;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * 68 + g8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * 68 + g8 * 8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const int m = m0 + wm * 128 + c * 32 + row;
                const int n = n0 + wn * 64 + g8 * 8;
                if (EPI == EPI_RESID) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rcur[k][e];
                }
                vec8T o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (T)v[e];
                if (EPI == EPI_QKV) {
                    if (m < p.Mvalid) {
                        int b = qkv_b0, sq = m - qkv_b0 * p.Sp;
                        while (sq >= p.Sp) { sq -= p.Sp; ++b; }          // a 256-row tile spans <= 5 sequences (Sp >= 64)
                        const int nn = n - which * p.H, hh = nn >> 6, dd = nn & 63;     // dd is a multiple of 8: one 16-B unit
                        const int bh = b * p.nh + hh;
                        T* dst = which == 0 ? reinterpret_cast<T*>(p.Qh) + glc_qoff(p.Sp, bh, sq, dd)
                                            : reinterpret_cast<T*>(p.Kh) + glc_koff(p.Sp, bh, sq, dd);
                        *reinterpret_cast<vec8T*>(dst) = o;
                    }
                } else {
                    *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.C) + (size_t)m * N + n) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // V third: D[m = 16i + 4g + r][n = 16j + r16]; patch [64 rows dd][32 cols key], row stride 36 floats
        float bn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bn[j] = bias ? bias[n0 + wn * 64 + j * 16 + r16] : 0.f;
        const int hh = (n0 + wn * 64 - 2 * p.H) >> 6;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[2 * c + ii][j];
                    v[0] += bn[j]; v[1] += bn[j]; v[2] += bn[j]; v[3] += bn[j];
                    *reinterpret_cast<f32x4*>(stg + (j * 16 + r16) * 36 + ii * 16 + 4 * g) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, dd = idx >> 2, kg = idx & 3;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + dd * 36 + kg * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + dd * 36 + kg * 8 + 4);
                vec8T o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (T)lo[e]; o[4 + e] = (T)hi[e]; }
                const int m = m0 + wm * 128 + c * 32 + kg * 8;           // first of 8 consecutive keys
                if (m < p.Mvalid) {
                    int b = qkv_b0, sq = m - qkv_b0 * p.Sp;
                    while (sq >= p.Sp) { sq -= p.Sp; ++b; }
                    *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.Vt) + glc_voff(p.Sp, b * p.nh + hh, dd, sq)) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename T, int EPI, bool VMODE> const char* launch_e(hipStream_t st, const GemmArgs& a, int n_tile0, int ntn) {
    static std::atomic<unsigned> lds_ok{0};        // per device: several engines of one process may sit on different GPUs
    if (!glc_raise_lds_limit(gemm256_kernel<T, EPI, VMODE>, 2 * STAGE, lds_ok)) return "gemm256: cannot raise the dynamic LDS limit";
    const int grid = (a.Mpad / TM) * ntn;
    hipLaunchKernelGGL((gemm256_kernel<T, EPI, VMODE>), dim3(grid), dim3(512), 2 * STAGE, st, a, n_tile0, ntn);
    return nullptr;
}
template <typename T> const char* launch_t(hipStream_t st, int epi, const GemmArgs& a) {
    const int ntn = a.N / TN;
    switch (epi) {
        case EPI_BIAS: return launch_e<T, EPI_BIAS, false>(st, a, 0, ntn);
        case EPI_GELU: return launch_e<T, EPI_GELU, false>(st, a, 0, ntn);
        case EPI_RESID: return launch_e<T, EPI_RESID, false>(st, a, 0, ntn);
        case EPI_QKV: {   // Q|K columns in row orientation, V columns transposed: two grids, one stream
            const int nqk = 2 * a.H / TN, nq = a.qkv_skip_q ? a.H / TN : 0;     // skip the Q columns when asked
            const char* m = launch_e<T, EPI_QKV, false>(st, a, nq, nqk - nq);
            return m ? m : launch_e<T, EPI_QKV, true>(st, a, nqk, ntn - nqk);
        }
    }
    return "gemm256: bad epilogue";
}

}  // namespace

bool glc_gemm256_supported(int dtype, const GemmArgs& a) {
    return (dtype == GLC_DT_BF16 || dtype == GLC_DT_F16) && a.Mpad > 0 && a.Mpad % TM == 0 && a.N > 0 && a.N % TN == 0 &&
           a.K > 0 && a.K % 64 == 0;
}

// Host-side shape contract: 16-bit T; Mpad % 256 == 0 (buffers allocated with Mpad rows), N % 256 == 0,
// K % 64 == 0; EPI_QKV: H % 256 == 0 (a tile never straddles Q|K|V), Sp % 64 == 0.
const char* glc_launch_gemm256(hipStream_t st, int dtype, int epi, const GemmArgs& a_in) {
    if (a_in.a_stats || a_in.r_stats || a_in.ln_part) return "gemm: the LayerNorm-fold arguments exist for the staggered 256-tile kernel only";
    GemmArgs a = a_in;
    static const int env_spread = getenv("GLC_GEMM_SPREAD") ? atoi(getenv("GLC_GEMM_SPREAD")) : 1;   // default on (A/B switch)
    a.spread_dma = env_spread;
    if (!glc_gemm256_supported(dtype, a)) return "gemm256: unsupported shape";
    if (!a.A || !a.W) return "gemm256: null operand";
    if (epi == EPI_QKV) {
        if (a.H % 256 || a.N != 3 * a.H || a.Sp % 64 || a.Sp < 64 || !a.Qh || !a.Kh || !a.Vt || a.nh * 64 != a.H) return "gemm256: bad QKV epilogue shape";
    } else if (!a.C) return "gemm256: null output";
    if (epi == EPI_RESID && !a.resid) return "gemm256: null residual";
    return dtype == GLC_DT_BF16 ? launch_t<bf16_t>(st, epi, a) : launch_t<f16_t>(st, epi, a);
}
