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
//   * two 64 KiB stages: the DMA of K-tile t+1 is in flight while the MFMAs of tile t run; one
//     s_waitcnt vmcnt(0) + s_barrier per K-tile;
//   * workgroup ids are remapped so that the tiles of one XCD (blockIdx % 8) are consecutive in
//     (m, n) order: the n-tiles of an A panel share one L2.
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int ROWB = 128;                  // bytes of K per row per stage (64 x 16-bit)
constexpr int STAGE = (TM + TN) * ROWB;    // 64 KiB
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
    const int m0 = (tile / ntn) * TM, n0 = (n_tile0 + tile % ntn) * TN;   // this launch covers n-tiles [n_tile0, n_tile0 + ntn)

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

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
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
                if (!vmode) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(bf[j], af, acc[i][j]);   // D[n = 4g+r][m = r16]
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(af, bf[j], acc[i][j]);   // D[m = 4g+r][n = r16]
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my DMA pieces of tile kt+1 have landed
        __builtin_amdgcn_s_barrier();                        // everyone's have; everyone is done reading `buf`
    }

    // ---------------- epilogue ----------------
    const float* __restrict__ bias = p.bias;
    const int qkv_b0 = (EPI == EPI_QKV) ? m0 / p.Sp : 0;
    if (!vmode) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * g;
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
            if (bias) { const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n); b0 = bv[0]; b1 = bv[1]; b2 = bv[2]; b3 = bv[3]; }
            // QKV scatter: (which, head, dd) are uniform over i
            const int which = (EPI == EPI_QKV) ? n0 / p.H : 0;
            const int nn = n - which * p.H, hh = nn >> 6, dd = nn & 63;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wm * 128 + i * 16 + r16;
                float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
                if (EPI == EPI_GELU) { v0 = glc_gelu(v0); v1 = glc_gelu(v1); v2 = glc_gelu(v2); v3 = glc_gelu(v3); }
                if (EPI == EPI_RESID) {
                    float r0, r1, r2, r3;
                    load4<T>(reinterpret_cast<const T*>(p.resid) + (size_t)m * N + n, r0, r1, r2, r3);
                    v0 += r0; v1 += r1; v2 += r2; v3 += r3;
                }
                if (EPI == EPI_QKV) {
                    if (m < p.Mvalid) {
                        int b = qkv_b0, s = m - qkv_b0 * p.Sp;
                        while (s >= p.Sp) { s -= p.Sp; ++b; }          // a 256-row tile spans <= 5 sequences (Sp >= 64)
                        const int bh = b * p.nh + hh;
                        T* dst = which == 0 ? reinterpret_cast<T*>(p.Qh) + glc_qoff(p.Sp, bh, s, dd)
                                            : reinterpret_cast<T*>(p.Kh) + glc_koff(p.Sp, bh, s, dd);
                        store4<T>(dst, v0, v1, v2, v3);
                    }
                } else {
                    store4<T>(reinterpret_cast<T*>(p.C) + (size_t)m * N + n, v0, v1, v2, v3);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + r16;
            const float bv = bias ? bias[n] : 0.f;
            const int nn = n - 2 * p.H, hh = nn >> 6, dd = nn & 63;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + wm * 128 + i * 16 + 4 * g;
                if (m < p.Mvalid) {
                    int b = qkv_b0, s = m - qkv_b0 * p.Sp;
                    while (s >= p.Sp) { s -= p.Sp; ++b; }
                    T* dst = reinterpret_cast<T*>(p.Vt) + glc_voff(p.Sp, b * p.nh + hh, dd, s);
                    store4<T>(dst, acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
                }
            }
        }
    }
}

template <typename T, int EPI, bool VMODE> const char* launch_e(hipStream_t st, const GemmArgs& a, int n_tile0, int ntn) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<T, EPI, VMODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * STAGE) != hipSuccess)
            return "gemm256: cannot raise the dynamic LDS limit";
        attr_set = true;
    }
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
            const int nqk = 2 * a.H / TN;
            const char* m = launch_e<T, EPI_QKV, false>(st, a, 0, nqk);
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
const char* glc_launch_gemm256(hipStream_t st, int dtype, int epi, const GemmArgs& a) {
    if (!glc_gemm256_supported(dtype, a)) return "gemm256: unsupported shape";
    if (!a.A || !a.W) return "gemm256: null operand";
    if (epi == EPI_QKV) {
        if (a.H % 256 || a.N != 3 * a.H || a.Sp % 64 || a.Sp < 64 || !a.Qh || !a.Kh || !a.Vt || a.nh * 64 != a.H) return "gemm256: bad QKV epilogue shape";
    } else if (!a.C) return "gemm256: null output";
    if (epi == EPI_RESID && !a.resid) return "gemm256: null residual";
    return dtype == GLC_DT_BF16 ? launch_t<bf16_t>(st, epi, a) : launch_t<f16_t>(st, epi, a);
}
