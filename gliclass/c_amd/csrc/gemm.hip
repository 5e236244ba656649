// C = A * W^T (+bias, +epilogue) on MFMA, for the encoder's dense projections
// (modeling_deberta_v2.py:231-233 QKV, :49-53 attention output, :393-396 intermediate, :408-412 output)
// and the GLiClass head's projectors.
//
// Layout: A [Mpad, K] row-major (activations), W [N, K] row-major (nn.Linear weight), both element
// type T in {float, bf16, f16}; fp32 accumulate.  128x128 block tile, 4 waves (2x2), each wave
// 64x64 = 4x4 MFMA 16x16 tiles.  K is consumed in stages of 128 BYTES per row (64 16-bit / 32 f32
// elements) so the staging/LDS code is identical for every T.  Stages are double-buffered in LDS
// (144-B padded rows -> conflict-free ds_read_b128) with the global loads of stage t+1 issued
// before the MFMAs of stage t and written to LDS after them (issue-early / write-late).
//
// Output orientation: the wave computes D^T = W_tile * A_tile^T (lane = output row m, 4 regs = 4
// consecutive columns n) so each lane stores 4 packed elements; the V third of the fused QKV GEMM
// flips the operands (lane = n, regs = 4 consecutive m) to write V transposed.  Q, K and V^T leave
// this kernel in the fragment-major layouts of glc_layout.h (what the attention MFMAs load).
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

// Split-f16 unit store (GemmArgs::qkv_split): `dst` addresses 4 consecutive elements (e & 7 in {0, 4}) of an 8-element fp32 unit;
// the unit's 32 bytes become [8 hi halves | 8 lo halves] (glc_common.h f16x8s), so the 4 values go to halves (e&7).. of both parts.
template <typename T> __device__ __forceinline__ void store4_split(T* dst, float a, float b, float c, float d) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(dst);
    f16_t* unit = reinterpret_cast<f16_t*>(addr & ~(uintptr_t)31);
    const int sub = (int)((addr & 31) >> 2);                               // 0 or 4
    const f16_t ha = (f16_t)a, hb = (f16_t)b, hc = (f16_t)c, hd = (f16_t)d;
    store4<f16_t>(unit + sub, (float)ha, (float)hb, (float)hc, (float)hd);
    store4<f16_t>(unit + 8 + sub, a - (float)ha, b - (float)hb, c - (float)hc, d - (float)hd);
}

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;          // bytes of K per row per stage
constexpr int ROWP = ROWB + 16;    // padded LDS row stride (bytes)

// SPLIT (T = float only): the fp32 operands are split on their way into LDS into two f16 halves, x = hi + lo with hi = f16(x),
// lo = f16(x - hi), and every product runs as three f16 MFMAs a_lo*w_hi + a_hi*w_lo + a_hi*w_hi with fp32 accumulation
// (the lo*lo term is below 2^-22 of the product).  That keeps ~21-22 mantissa bits of each operand — the parity-grade
// accuracy of the fp32 mode (tests assert the same 1e-4 envelope) — at 16 cycles x 3 per 16x16x32 block instead of
// 32 cycles x 8 on the fp32 16x16x4 MFMA, i.e. up to 5.3x the fp32 matrix rate.  Range: operands must stay below the f16
// maximum (65504); the engine checks nothing at run time, GLICLASS_F32_GEMM=native selects the plain fp32-MFMA kernel.
// LDS image per row and stage (32 k): [32 hi halves | 32 lo halves] = the same 128 bytes as 32 floats.
template <typename T, int EPI, bool SPLITK = false, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs p, int ksplit) {   // two workgroups per CU: the QKV epilogue variants otherwise take 284 registers (one wave per SIMD)
    static_assert(!SPLIT || sizeof(T) == 4, "the split path takes fp32 operands");
    typedef typename Frag<T>::type frag_t;
    constexpr int BK = ROWB / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * BM * ROWP];
    unsigned char* As = smem;                       // [2][BM][ROWP]
    unsigned char* Bs = smem + 2 * BM * ROWP;       // [2][BN][ROWP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = (blockIdx.x + ((EPI == EPI_QKV && p.qkv_skip_q) ? p.H / BN : 0)) * BN;
    const int K = p.K, N = p.N;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const bool grp2 = p.W2 != nullptr && m0 >= p.m_split;                   // block-uniform: second weight group
    const T* __restrict__ W = reinterpret_cast<const T*>(grp2 ? p.W2 : p.W);

    // staging map: thread -> 16-B chunk cc of rows (tid>>3) + 32*i
    const int cc = tid & 7, srow = tid >> 3;
    const T* ga = A + (size_t)(m0 + srow) * K + cc * (16 / (int)sizeof(T));       // advanced to this block's K range below
    const T* gw = W + (size_t)(n0 + srow) * K + cc * (16 / (int)sizeof(T));
    const size_t gstep = (size_t)32 * K;
    u32x4 ra[4], rw[4];

    const bool vmode = (EPI == EPI_QKV) && (n0 >= 2 * p.H);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk_all = K / BK;
    const int per = SPLITK ? (nk_all + ksplit - 1) / ksplit : nk_all;       // K stages of this block: [kbeg, kbeg + nk)
    const int kbeg = SPLITK ? (int)blockIdx.z * per : 0;
    const int nk = SPLITK ? (kbeg + per <= nk_all ? per : (nk_all > kbeg ? nk_all - kbeg : 0)) : nk_all;
    ga += (size_t)kbeg * BK;
    gw += (size_t)kbeg * BK;
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const u32x4*>(ga + (size_t)i * gstep + (size_t)kt * BK);
            rw[i] = *reinterpret_cast<const u32x4*>(gw + (size_t)i * gstep + (size_t)kt * BK);
        }
    };
    typedef __attribute__((ext_vector_type(4))) f16_t f16x4_t;
    auto split_store = [&](unsigned char* row, const u32x4& raw) {     // 4 floats (k = 4cc..4cc+3) -> hi at cc*8, lo at 64 + cc*8
        const f32x4 xf = __builtin_bit_cast(f32x4, raw);    // (bit_cast of a single vector ELEMENT reads element 0 under hipcc 7.2)
        f16x4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f16_t h = (f16_t)xf[e];
            hi[e] = h;
            lo[e] = (f16_t)(xf[e] - (float)h);
        }
        *reinterpret_cast<f16x4_t*>(row + cc * 8) = hi;
        *reinterpret_cast<f16x4_t*>(row + 64 + cc * 8) = lo;
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned char* arow = As + (size_t)buf * BM * ROWP + (srow + 32 * i) * ROWP;
            unsigned char* brow = Bs + (size_t)buf * BN * ROWP + (srow + 32 * i) * ROWP;
            if constexpr (SPLIT) {
                split_store(arow, ra[i]);
                if (p.w_presplit) *reinterpret_cast<u32x4*>(brow + cc * 16) = rw[i];    // split once at load: the raw 128-byte group IS the row image
                else split_store(brow, rw[i]);
            } else {
                *reinterpret_cast<u32x4*>(arow + cc * 16) = ra[i];
                *reinterpret_cast<u32x4*>(brow + cc * 16) = rw[i];
            }
        }
    };

    if (nk > 0) { gload(0); sstore(0); }
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char* as = As + (size_t)cur * BM * ROWP + (wm * 64 + r16) * ROWP + g * 16;
        const unsigned char* bs = Bs + (size_t)cur * BN * ROWP + (wn * 64 + r16) * ROWP + g * 16;
        if constexpr (SPLIT) {
            f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(as + i * 16 * ROWP);
                al[i] = *reinterpret_cast<const f16x8*>(as + i * 16 * ROWP + 64);
                bh[i] = *reinterpret_cast<const f16x8*>(bs + i * 16 * ROWP);
                bl[i] = *reinterpret_cast<const f16x8*>(bs + i * 16 * ROWP + 64);
            }
            if (!vmode) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                                  // D[n][m]; small terms first
                        mma16(bh[j], al[i], acc[j][i]);
                        mma16(bl[j], ah[i], acc[j][i]);
                        mma16(bh[j], ah[i], acc[j][i]);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                                  // D[m][n]
                        mma16(al[i], bh[j], acc[i][j]);
                        mma16(ah[i], bl[j], acc[i][j]);
                        mma16(ah[i], bh[j], acc[i][j]);
                    }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const frag_t*>(as + i * 16 * ROWP + ks * 64);
                bf[i] = *reinterpret_cast<const frag_t*>(bs + i * 16 * ROWP + ks * 64);
            }
            if (!vmode) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mma16(bf[j], af[i], acc[j][i]);   // D[n][m]
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(af[i], bf[j], acc[i][j]);   // D[m][n]
            }
        }
        }
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    if constexpr (SPLITK) {
        // raw fp32 partial tile -> ws[z][m][n]; bias / activation / residual are applied by splitk_reduce_kernel
        float* wsz = p.ws + (size_t)blockIdx.z * p.Mpad * N;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 64 + i * 16 + r16, n = n0 + wn * 64 + j * 16 + 4 * g;
                *reinterpret_cast<f32x4*>(wsz + (size_t)m * N + n) = acc[j][i];
            }
        return;
    }
    // ---------------- epilogue ----------------
    const float* __restrict__ bias = grp2 ? p.bias2 : p.bias;
    const int qkv_b0 = (EPI == EPI_QKV) ? m0 / p.Sp : 0;   // block-uniform (scalar) division
    if (!vmode) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * g;
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
            if (bias) { f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n); b0 = bv[0]; b1 = bv[1]; b2 = bv[2]; b3 = bv[3]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 64 + i * 16 + r16;
                float v0 = acc[j][i][0] + b0, v1 = acc[j][i][1] + b1, v2 = acc[j][i][2] + b2, v3 = acc[j][i][3] + b3;
                if (EPI == EPI_GELU) { v0 = glc_gelu(v0); v1 = glc_gelu(v1); v2 = glc_gelu(v2); v3 = glc_gelu(v3); }
                if (EPI == EPI_RESID) {
                    float r0, r1, r2, r3;
                    load4<T>(reinterpret_cast<const T*>(p.resid) + (size_t)m * N + n, r0, r1, r2, r3);
                    v0 += r0; v1 += r1; v2 += r2; v3 += r3;
                }
                if (EPI == EPI_QKV) {
                    if (m < p.Mvalid) {
                        const int which = n0 / p.H;                 // block-uniform: 0 = Q, 1 = K
                        const int nn = n - which * p.H;
                        const int hh = nn >> 6, dd = nn & 63;
                        int b = qkv_b0, s = m - qkv_b0 * p.Sp;     // a 128-row tile spans at most two sequences (Sp >= 64)
                        if (s >= p.Sp) { s -= p.Sp; ++b; }
                        const int bh = b * p.nh + hh;
                        T* dst = which == 0 ? reinterpret_cast<T*>(p.Qh) + glc_qoff(p.Sp, bh, s, dd)
                                            : reinterpret_cast<T*>(p.Kh) + glc_koff(p.Sp, bh, s, dd);
                        if (sizeof(T) == 4 && p.qkv_split) store4_split(dst, v0, v1, v2, v3);
                        else store4<T>(dst, v0, v1, v2, v3);
                    }
                } else {
                    store4<T>(reinterpret_cast<T*>(p.C) + (size_t)m * N + n, v0, v1, v2, v3);
                }
            }
        }
    } else {
        // V third: D[m = 4g+r][n = r16] -> Vt[b][hh][dd][s .. s+3]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + r16;
            const float bv = bias ? bias[n] : 0.f;
            const int nn = n - 2 * p.H;
            const int hh = nn >> 6, dd = nn & 63;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wm * 64 + i * 16 + 4 * g;
                if (m < p.Mvalid) {
                    int b = qkv_b0, s = m - qkv_b0 * p.Sp;
                    if (s >= p.Sp) { s -= p.Sp; ++b; }
                    T* dst = reinterpret_cast<T*>(p.Vt) + glc_voff(p.Sp, b * p.nh + hh, dd, s);
                    if (sizeof(T) == 4 && p.qkv_split) store4_split(dst, acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
                    else store4<T>(dst, acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
                }
            }
        }
    }
}

// Second pass of the split-K path: C[m, n .. n+3] = epi(sum_z ws[z][m][n] + bias[n]) in a fixed summation order (deterministic).
template <typename T, int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p, int ksplit) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, nq = (size_t)p.N / 4;
    if (idx >= (size_t)p.Mpad * nq) return;
    const int m = (int)(idx / nq), n = (int)(idx - (size_t)m * nq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.ws + (size_t)m * p.N + n);
    for (int z = 1; z < ksplit; ++z) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.ws + ((size_t)z * p.Mpad + m) * p.N + n);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    const float* bias = (p.W2 != nullptr && m >= p.m_split) ? p.bias2 : p.bias;
    if (bias) { const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
    if (EPI == EPI_GELU) { v[0] = glc_gelu(v[0]); v[1] = glc_gelu(v[1]); v[2] = glc_gelu(v[2]); v[3] = glc_gelu(v[3]); }
    if (EPI == EPI_RESID) {
        float r0, r1, r2, r3;
        load4<T>(reinterpret_cast<const T*>(p.resid) + (size_t)m * p.N + n, r0, r1, r2, r3);
        v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
    }
    store4<T>(reinterpret_cast<T*>(p.C) + (size_t)m * p.N + n, v[0], v[1], v[2], v[3]);
}

// Split-K pays when the 128x128 tiles alone leave most CUs idle and the K loop is long: the reference's own batches of 8 short
// texts give M = B*Sp of ~1 k rows, i.e. 48 tiles for N = 768.  Returns the number of K parts (1 = no split).
int splitk_parts(int epi, const GemmArgs& a, int nk) {
    static const int mode = glc_dev_env("GLC_GEMM_SPLITK") ? atoi(glc_dev_env("GLC_GEMM_SPLITK")) : 1;        // developer A/B switch (0 = off)
    if (!mode || epi == EPI_QKV || !a.ws || nk < 8) return 1;
    const int ncu = glc_device_cus();
    const long long tiles = (long long)(a.N / BN) * (a.Mpad / BM);
    if (tiles * 2 > ncu) return 1;
    int parts = (int)(ncu / tiles);                       // aim at about one workgroup per CU
    if (parts > 8) parts = 8;
    if (parts > nk / 4) parts = nk / 4;                   // keep at least 4 K stages per part
    while (parts > 1 && (size_t)parts * a.Mpad * a.N * sizeof(float) > a.ws_bytes) --parts;
    return parts < 2 ? 1 : parts;
}

template <typename T, bool SPLIT = false> void launch_t(hipStream_t st, int epi, const GemmArgs& a) {
    const int nskip = (epi == EPI_QKV && a.qkv_skip_q) ? a.H / BN : 0;   // pruned last layer: K and V^T columns only
    dim3 grid(a.N / BN - nskip, a.Mpad / BM), block(256);
    const int parts = splitk_parts(epi, a, a.K / (ROWB / (int)sizeof(T)));
    if (parts > 1) {
        grid.z = parts;
        const unsigned rblocks = (unsigned)(((size_t)a.Mpad * (a.N / 4) + 255) / 256);
        switch (epi) {
            case EPI_BIAS: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_BIAS, true, SPLIT>), grid, block, 0, st, a, parts);
                           hipLaunchKernelGGL((splitk_reduce_kernel<T, EPI_BIAS>), dim3(rblocks), block, 0, st, a, parts); break;
            case EPI_GELU: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_GELU, true, SPLIT>), grid, block, 0, st, a, parts);
                           hipLaunchKernelGGL((splitk_reduce_kernel<T, EPI_GELU>), dim3(rblocks), block, 0, st, a, parts); break;
            default:       hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_RESID, true, SPLIT>), grid, block, 0, st, a, parts);
                           hipLaunchKernelGGL((splitk_reduce_kernel<T, EPI_RESID>), dim3(rblocks), block, 0, st, a, parts); break;
        }
        return;
    }
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_BIAS, false, SPLIT>), grid, block, 0, st, a, 1); break;
        case EPI_GELU: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_GELU, false, SPLIT>), grid, block, 0, st, a, 1); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_RESID, false, SPLIT>), grid, block, 0, st, a, 1); break;
        case EPI_QKV: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_QKV, false, SPLIT>), grid, block, 0, st, a, 1); break;
    }
}

}  // namespace

// Host-side shape contract (checked here so a bad shape can never reach the kernel):
//   Mpad % 128 == 0 and every [Mpad, *] buffer is allocated with Mpad rows; N % 128 == 0;
//   K % (128 / sizeof(T)) == 0; for EPI_QKV additionally H % 128 == 0, Sp % 64 == 0 (fragment-major
//   Q/K/V^T outputs, glc_layout.h).
const char* glc_launch_gemm(hipStream_t st, int dtype, int epi, const GemmArgs& a) {
    if (a.a_stats || a.r_stats || a.ln_part) return "gemm: the LayerNorm-fold arguments exist for the staggered 256-tile kernel only";
    const int esz = dtype == GLC_DT_F32 ? 4 : 2;
    if (a.Mpad <= 0 || a.Mpad % BM) return "gemm: Mpad must be a positive multiple of 128";
    if (a.N <= 0 || a.N % BN) return "gemm: N must be a multiple of 128";
    if (a.K <= 0 || a.K % (ROWB / esz)) return "gemm: K must be a multiple of 128 bytes";
    if (!a.A || !a.W) return "gemm: null operand";
    if (epi == EPI_QKV) {
        if (a.H % 128 || a.N != 3 * a.H || a.Sp % 64 || a.Sp < 64 || !a.Qh || !a.Kh || !a.Vt || a.nh * 64 != a.H) return "gemm: bad QKV epilogue shape";
    } else if (!a.C) return "gemm: null output";
    if (epi == EPI_RESID && !a.resid) return "gemm: null residual";
    if (a.W2 && (a.m_split % BM || a.m_split <= 0 || a.m_split >= a.Mpad)) return "gemm: m_split must be a tile-aligned row inside the matrix";
    switch (dtype) {
        case GLC_DT_F32: {
            // default: split-f16 (3-MFMA) products; GLICLASS_F32_GEMM=native keeps the fp32 16x16x4 MFMA kernel (A/B and wide-range data)
            static const bool split = !(getenv("GLICLASS_F32_GEMM") && !strcmp(getenv("GLICLASS_F32_GEMM"), "native"));
            if (split) launch_t<float, true>(st, epi, a); else launch_t<float>(st, epi, a);
            break;
        }
        case GLC_DT_BF16: launch_t<bf16_t>(st, epi, a); break;
        case GLC_DT_F16: launch_t<f16_t>(st, epi, a); break;
        default: return "gemm: bad dtype";
    }
    return nullptr;
}
