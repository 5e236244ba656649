// Row-wise (HBM-bound) kernels: embedding gather + LayerNorm + mask, LayerNorm, the per-row input
// scan, head gather / dot-product scorer and dtype conversion.  One wave per row, 16-byte vector
// accesses, fp32 statistics with wavefront shuffles (no LDS).
#include "glc_common.h"
#include "glc_kernels.h"

namespace {

constexpr int MAXC = 4;  // chunks of 16 B per lane: H <= 64*MAXC*VEC (H <= 2048 for 16-bit, 1024 for f32)

template <typename T> struct RowVec { static constexpr int VEC = 16 / (int)sizeof(T); };

// LayerNorm of one row held in registers by one wave. torch.nn.LayerNorm semantics (biased var).
template <typename T, bool MASKED>
__device__ __forceinline__ void ln_row_wave(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, float eps, int H, float mk, int lane) {
    constexpr int VEC = RowVec<T>::VEC;
    typedef __attribute__((ext_vector_type(VEC))) T vecT;
    const int nch = H / VEC;
    float v[MAXC][VEC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            vecT t = *reinterpret_cast<const vecT*>(x + (size_t)ch * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { v[c][e] = (float)t[e]; s += v[c][e]; }
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (lane + 64 * c < nch) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v[c][e] - mean; ss += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            vecT o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float r = (v[c][e] - mean) * rstd * gamma[ch * VEC + e] + beta[ch * VEC + e];
                if (MASKED) r *= mk;
                o[e] = (T)r;
            }
            *reinterpret_cast<vecT*>(y + (size_t)ch * VEC) = o;
        }
    }
}

// ---- group-split ("GS") activations of the fp32 mode ----
// A row of K fp32 values is stored in the same 4 K bytes as K/32 groups of [32 hi halves | 32 lo halves], x = hi + lo with
// hi = f16(x), lo = f16(x - hi) — the row image the split-f16 GEMMs consume (gemm.hip presplit weights use the same format), so the
// 256-tile LDS-DMA GEMM can fetch hi and lo parts of a 32-deep K step as plain 64-byte row pieces.  Element e of a row lives at
// halves (e >> 5) * 64 + (e & 31) (hi) and + 32 (lo).
// LayerNorm of one fp32 row by one wave, output in the GS format (8 elements per lane and chunk)
// (GX = true: the same 128-byte groups as GX rows — [32 hi | 32 lo8 | 32 hi8], glc_common.h — for the MX cross-term GEMM; activation exponent 0)
template <bool GX> __device__ __forceinline__ void row_store8(f16_t* y, int e0, const float (&v)[8], unsigned* sat = nullptr, int act_sc = 0) {
    if constexpr (GX) gx_store8(reinterpret_cast<unsigned char*>(y), e0, v, gx_act_khi(act_sc), gx_act_klo(act_sc), sat);      // (sat: fp8 range guard; act_sc: activation exponent, glc_common.h)
    else gs_store8(y, e0, v);
}
template <bool GX> __device__ __forceinline__ void row_load8(const f16_t* y, int e0, float (&v)[8], int act_sc = 0) {
    if constexpr (GX) gx_load8(reinterpret_cast<const unsigned char*>(y), e0, v, gx_pow2_inv(gx_act_klo(act_sc)));
    else gs_load8(y, e0, v);
}
template <bool MASKED, bool GX = false>
__device__ __forceinline__ void ln_row_wave_gs(const float* __restrict__ x, f16_t* __restrict__ y, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float eps, int H, float mk, int lane, unsigned* sat = nullptr, int act_sc = 0) {
    const int nch = H / 8;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)ch * 8), b = *reinterpret_cast<const f32x4*>(x + (size_t)ch * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[c][e] = a[e]; v[c][4 + e] = b[e]; s += a[e] + b[e]; }
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (lane + 64 * c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; ss += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float r = (v[c][e] - mean) * rstd * gamma[ch * 8 + e] + beta[ch * 8 + e];
                if (MASKED) r *= mk;
                o[e] = r;
            }
            row_store8<GX>(y, ch * 8, o, sat, act_sc);
        }
    }
}
template <bool GX>
__global__ __launch_bounds__(256) void layernorm_gs_kernel(const float* __restrict__ X, f16_t* __restrict__ Y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int M, int H, unsigned* sat, int act_sc) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    ln_row_wave_gs<false, GX>(X + (size_t)row * H, Y + (size_t)row * 2 * H, gamma, beta, eps, H, 1.f, threadIdx.x & 63, sat, act_sc);
}
// decoder backbone, RMSNorm folded into the GEMMs: the embedding rows (plain fp32) enter the pipeline as raw group-split rows + (0, rstd)
template <bool GX>
__global__ __launch_bounds__(256) void rows_to_gs_rms_kernel(const float* __restrict__ X, f16_t* __restrict__ Y, float2* __restrict__ stats, float eps, int M, int H, unsigned* sat, int act_sc) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63, nch = H / 8;
    const float* x = X + (size_t)row * H;
    f16_t* y = Y + (size_t)row * 2 * H;
    float ss = 0.f;
    for (int ch = lane; ch < nch; ch += 64) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)ch * 8), b = *reinterpret_cast<const f32x4*>(x + (size_t)ch * 8 + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
        row_store8<GX>(y, ch * 8, v, sat, act_sc);
    }
    ss = wave_sum(ss);
    if (lane == 0) stats[row] = make_float2(0.f, rsqrtf(ss / (float)H + eps));
}
// LayerNorm statistics from the producer GEMM's partials (GemmArgs::ln_part: per 64-column block the sum and the squared deviations from the block mean)
__global__ __launch_bounds__(256) void ln_stats_kernel(const float2* __restrict__ part, int nparts, float2* __restrict__ stats, int M, double invH, float eps, int rms) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    // partials: (sum, M2 about the block's own mean) of each 64-column block; Chan's merge: M2 = sum M2_i + sum 64 (mean_i - mean)^2
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += (double)part[(size_t)row * nparts + i].x;
    const double mean = s * invH;
    double m2 = 0.0;
    for (int i = 0; i < nparts; ++i) {
        const float2 v = part[(size_t)row * nparts + i];
        const double d = (double)v.x * (1.0 / 64.0) - mean;
        m2 += (double)v.y + 64.0 * d * d;
    }
    // rms: RMSNorm statistics (0, 1 / sqrt(E[x^2] + eps)), E[x^2] = var + mean^2
    if (rms) stats[row] = make_float2(0.f, (float)(1.0 / sqrt(m2 * invH + mean * mean + (double)eps)));
    else stats[row] = make_float2((float)mean, (float)(1.0 / sqrt(m2 * invH + (double)eps)));
}
template <bool GX>
__global__ __launch_bounds__(256) void embed_gs_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask,
                                                       const float* __restrict__ table, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, f16_t* __restrict__ X,
                                                       float* __restrict__ kbias, int B, int S, int Sp, int H, int vocab, int pad_id, unsigned* sat, int act_sc) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);   // row in the padded [B, Sp] grid
    if (row >= B * Sp) return;
    const int lane = threadIdx.x & 63;
    const int b = row / Sp, s = row - b * Sp;
    long long id = pad_id;
    float mk = 0.f;
    if (s < S) {
        id = ids[(size_t)b * S + s];
        mk = mask[(size_t)b * S + s] != 0 ? 1.f : 0.f;
        if (id < 0 || id >= vocab) id = pad_id;
    }
    if (lane == 0) kbias[row] = mk != 0.f ? 0.f : GLC_NEG_BIG;
    ln_row_wave_gs<true, GX>(table + (size_t)id * H, X + (size_t)row * 2 * H, gamma, beta, eps, H, mk, lane, sat, act_sc);
}
// pruned last layer: rows the head reads, GS hidden states -> plain fp32 compact rows (that layer runs on the fp32-format kernels)
template <bool GX>
__global__ __launch_bounds__(256) void gather_rows_gs_kernel(const f16_t* __restrict__ X, const int* __restrict__ cls_pos, int c_cap,
                                                             float* __restrict__ Xs, int* __restrict__ sel_b, int* __restrict__ sel_q,
                                                             unsigned char* __restrict__ tile_flag, int B, int Sp, int H, int C, int act_sc) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * (1 + C)) return;
    const int lane = threadIdx.x & 63;
    int b = r, pos = 0;
    if (r >= B) {
        const int rr = r - B, j = rr % C;
        b = rr / C;
        pos = j < c_cap ? cls_pos[(size_t)b * c_cap + j] : -1;
        if (pos < 0) pos = 0;
    }
    if (lane == 0) { sel_b[r] = b; sel_q[r] = pos; if (tile_flag) tile_flag[(size_t)b * (Sp >> 5) + (pos >> 5)] = 1; }
    const f16_t* src = X + ((size_t)b * Sp + pos) * 2 * H;
    for (int i = lane; i < H / 8; i += 64) {
        float v[8];
        row_load8<GX>(src, i * 8, v, act_sc);
        *reinterpret_cast<f32x4*>(Xs + (size_t)r * H + i * 8) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(Xs + (size_t)r * H + i * 8 + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ X, T* __restrict__ Y, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int M, int H) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    ln_row_wave<T, false>(X + (size_t)row * H, Y + (size_t)row * H, gamma, beta, eps, H, 1.f, threadIdx.x & 63);
}

template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask,
                                                    const T* __restrict__ table, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, T* __restrict__ X,
                                                    float* __restrict__ kbias, int B, int S, int Sp, int H, int vocab, int pad_id) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);   // row in the padded [B, Sp] grid
    if (row >= B * Sp) return;
    const int lane = threadIdx.x & 63;
    const int b = row / Sp, s = row - b * Sp;
    long long id = pad_id;
    float mk = 0.f;
    if (s < S) {
        id = ids[(size_t)b * S + s];
        mk = mask[(size_t)b * S + s] != 0 ? 1.f : 0.f;
        if (id < 0 || id >= vocab) id = pad_id;
    }
    if (lane == 0) kbias[row] = mk != 0.f ? 0.f : GLC_NEG_BIG;
    ln_row_wave<T, true>(table + (size_t)id * H, X + (size_t)row * H, gamma, beta, eps, H, mk, lane);
}

// one block per batch row; ordered compaction of class-token positions by a block-wide scan
__global__ __launch_bounds__(256) void scan_rows_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int S,
                                                        int class_token, int embed_class_token, int* __restrict__ klen,
                                                        int* __restrict__ kfirst, int* __restrict__ cls_pos, int* __restrict__ cls_cnt, int c_cap) {
    __shared__ int cnt[256];
    __shared__ int last[256];
    __shared__ int first[256];
    __shared__ int tot;
    const int b = blockIdx.x, t = threadIdx.x;
    const int per = (S + 255) / 256;
    const int lo = t * per, hi = min(S, lo + per);
    int c = 0, lv = 0, fz = S;
    for (int s = lo; s < hi; ++s) {
        c += ids[(size_t)b * S + s] == class_token;
        if (mask[(size_t)b * S + s] != 0) lv = s + 1;
        else if (fz == S) fz = s;
    }
    cnt[t] = c;
    last[t] = lv;
    first[t] = fz;
    __syncthreads();
    if (t == 0) {
        int run = 0, mx = 0, mn = S;
        for (int i = 0; i < 256; ++i) { const int v = cnt[i]; cnt[i] = run; run += v; mx = max(mx, last[i]); mn = min(mn, first[i]); }
        cls_cnt[b] = run;
        klen[b] = mx;
        kfirst[b] = mn;
        tot = run;
    }
    __syncthreads();
    int j = cnt[t];
    for (int s = lo; s < hi; ++s)
        if (ids[(size_t)b * S + s] == class_token) {
            if (j < c_cap) cls_pos[(size_t)b * c_cap + j] = embed_class_token ? s : min(s + 1, S - 1);
            ++j;
        }
    // slots beyond this row's count
    for (int k = tot + t; k < c_cap; k += 256) cls_pos[(size_t)b * c_cap + k] = -1;
}

// Gt rows [0,B): pooled = hidden[b, 0, :] ("first" pooling); Gc rows b*C + j: class token j (zeros if absent)
template <typename T>
__global__ __launch_bounds__(256) void head_gather_kernel(const T* __restrict__ X, const int* __restrict__ cls_pos, int c_cap,
                                                          float* __restrict__ Gt, float* __restrict__ Gc, int B, int Sp, int H, int C,
                                                          const int* __restrict__ klen) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * (1 + C)) return;
    const int lane = threadIdx.x & 63;
    long long src = -1;
    float* dst;
    if (row < B) {                          // pooled row: position 0 ('first'), or the last attended token when klen is given
        const int last = klen ? (klen[row] > 0 ? klen[row] - 1 : 0) : 0;
        src = (long long)row * Sp + last; dst = Gt + (size_t)row * H;
    }
    else {
        const int r = row - B, b = r / C, j = r - b * C;
        const int pos = j < c_cap ? cls_pos[(size_t)b * c_cap + j] : -1;
        if (pos >= 0) src = (long long)b * Sp + pos;
        dst = Gc + (size_t)r * H;
    }
    for (int i = lane; i < H; i += 64) dst[i] = src >= 0 ? (float)X[(size_t)src * H + i] : 0.f;
}

// pooling = 'avg': Gt[b, :] = mean of X[b, s, :] over the attended positions (kbias == 0); block = (64 columns, batch row)
template <typename T>
__global__ __launch_bounds__(256) void pool_avg_kernel(const T* __restrict__ X, const float* __restrict__ kbias, float* __restrict__ Gt,
                                                       int Sp, int H) {
    __shared__ float red[4][64];
    __shared__ int cnt[4];
    const int b = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float acc = 0.f;
    int n = 0;
    for (int s = ph; s < Sp; s += 4)
        if (kbias[(size_t)b * Sp + s] == 0.f) { acc += (float)X[((size_t)b * Sp + s) * H + col]; ++n; }
    red[ph][threadIdx.x & 63] = acc;
    if ((threadIdx.x & 63) == 0) cnt[ph] = n;
    __syncthreads();
    if (ph == 0) {
        const int tot = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        Gt[(size_t)b * H + col] = tot > 0 ? v / (float)tot : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ X, const int* __restrict__ cls_pos, int c_cap,
                                                          T* __restrict__ Xs, int* __restrict__ sel_b, int* __restrict__ sel_q,
                                                          unsigned char* __restrict__ tile_flag, int B, int Sp, int H, int C) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * (1 + C)) return;
    const int lane = threadIdx.x & 63;
    int b = r, pos = 0;
    if (r >= B) {
        const int rr = r - B, j = rr % C;
        b = rr / C;
        pos = j < c_cap ? cls_pos[(size_t)b * c_cap + j] : -1;
        if (pos < 0) pos = 0;
    }
    if (lane == 0) { sel_b[r] = b; sel_q[r] = pos; if (tile_flag) tile_flag[(size_t)b * (Sp >> 5) + (pos >> 5)] = 1; }
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef __attribute__((ext_vector_type(VEC))) T vecT;
    const T* src = X + ((size_t)b * Sp + pos) * H;
    for (int i = lane; i < H / VEC; i += 64) reinterpret_cast<vecT*>(Xs + (size_t)r * H)[i] = reinterpret_cast<const vecT*>(src)[i];
}

template <typename T>
__global__ __launch_bounds__(256) void gather_sel_kernel(const T* __restrict__ src, const int* __restrict__ sel_b, const int* __restrict__ sel_q,
                                                         T* __restrict__ dst, int R, int Sp, int H) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    constexpr int VEC = 16 / (int)sizeof(T);
    typedef __attribute__((ext_vector_type(VEC))) T vecT;
    const T* s = src + ((size_t)sel_b[r] * Sp + sel_q[r]) * H;
    for (int i = lane; i < H / VEC; i += 64) reinterpret_cast<vecT*>(dst + (size_t)r * H)[i] = reinterpret_cast<const vecT*>(s)[i];
}

template <typename T>
__global__ __launch_bounds__(256) void head_gather_sel_kernel(const T* __restrict__ Xs, const int* __restrict__ cls_pos, int c_cap,
                                                              float* __restrict__ Gt, float* __restrict__ Gc, int B, int H, int C) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * (1 + C)) return;
    const int lane = threadIdx.x & 63;
    bool valid = true;
    float* dst;
    if (r < B) dst = Gt + (size_t)r * H;
    else {
        const int rr = r - B, b = rr / C, j = rr - b * C;
        valid = j < c_cap && cls_pos[(size_t)b * c_cap + j] >= 0;
        dst = Gc + (size_t)rr * H;
    }
    for (int i = lane; i < H; i += 64) dst[i] = valid ? (float)Xs[(size_t)r * H + i] : 0.f;
}

// logits[b*C + j] = <Tt[b], Cc[b*C+j]>   (gliclass scorer 'simple': einsum('BD,BCD->BC'))
__global__ __launch_bounds__(256) void head_score_kernel(const float* __restrict__ Tt, const float* __restrict__ Cc,
                                                         float* __restrict__ logits, int B, int C, int H, int normalize,
                                                         float logit_scale) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= B * C) return;
    const int lane = threadIdx.x & 63;
    const int b = idx / C;
    const float* t = Tt + (size_t)b * H;
    const float* c = Cc + (size_t)idx * H;
    float dot = 0.f, nt = 0.f, nc = 0.f;
    for (int i = lane; i < H; i += 64) { const float a = t[i], d = c[i]; dot += a * d; nt += a * a; nc += d * d; }
    dot = wave_sum(dot);
    if (normalize) {
        nt = wave_sum(nt); nc = wave_sum(nc);
        dot = dot / ((sqrtf(nt) + 1e-8f) * (sqrtf(nc) + 1e-8f)) * logit_scale;
    }
    if (lane == 0) logits[idx] = dot;
}

// ---- scorers other than 'simple' (include/gliclass_hip.h): row shuffles and the last Linear(., 1); their GEMMs are the head's ----
// weighted-dot: cat[b*C + j] = [t1(b), c1(b,j), t2(b) * c2(b,j)]  with (t1|t2) = St[b] and (c1|c2) = Sc[b*C + j], rows of 2H
__global__ __launch_bounds__(256) void scorer_wd_cat_kernel(const float* __restrict__ St, const float* __restrict__ Sc, float* __restrict__ cat,
                                                            int B, int C, int H) {
    const int r = blockIdx.x;
    const float* t = St + (size_t)(r / C) * 2 * H;
    const float* c = Sc + (size_t)r * 2 * H;
    float* o = cat + (size_t)r * 3 * H;
    for (int i = threadIdx.x; i < H; i += 256) { o[i] = t[i]; o[H + i] = c[i]; o[2 * H + i] = t[H + i] * c[H + i]; }
}
// mlp: pair[b*C + j] = [text(b), class(b,j)]
__global__ __launch_bounds__(256) void scorer_pair_kernel(const float* __restrict__ Tt, const float* __restrict__ Cc, float* __restrict__ out,
                                                          int B, int C, int H) {
    const int r = blockIdx.x;
    const float* t = Tt + (size_t)(r / C) * H;
    const float* c = Cc + (size_t)r * H;
    float* o = out + (size_t)r * 2 * H;
    for (int i = threadIdx.x; i < H; i += 256) { o[i] = t[i]; o[H + i] = c[i]; }
}
__global__ __launch_bounds__(256) void relu_kernel(float* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = fmaxf(x[i], 0.f);
}
// X[r][:] /= (|X[r]| + 1e-8)   (normalize_features ahead of the 'weighted-dot' / 'mlp' scorers; the dot scorer normalises inside its kernel); one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ X, int rows, int H) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float* x = X + (size_t)r * H;
    float a = 0.f;
    for (int i = lane; i < H; i += 64) a += x[i] * x[i];
    a = wave_sum(a);
    const float inv = 1.0f / (sqrtf(a) + 1e-8f);
    for (int i = lane; i < H; i += 64) x[i] *= inv;
}
// logits[r] = (sum_k relu(X[r][k]) * w[k] + bias[0]) * scale   (ReLU -> Linear(K, 1)); one wave per row
__global__ __launch_bounds__(256) void relu_dot_kernel(const float* __restrict__ X, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ logits, int rows, int K, float scale) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* x = X + (size_t)r * K;
    float a = 0.f;
    for (int i = lane; i < K; i += 64) a += fmaxf(x[i], 0.f) * w[i];
    a = wave_sum(a);
    if (lane == 0) logits[r] = (a + bias[0]) * scale;
}

template <typename T>
__global__ __launch_bounds__(256) void convert_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (T)src[i];
}
template <typename T>
__global__ __launch_bounds__(256) void to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (float)src[i];
}

template <typename T> bool h_ok(int H) { return H > 0 && H % RowVec<T>::VEC == 0 && H / RowVec<T>::VEC <= 64 * MAXC; }

}  // namespace

#define DISPATCH_T(dtype, CALL)                                   \
    switch (dtype) {                                              \
        case GLC_DT_F32: { typedef float T; CALL; } break;        \
        case GLC_DT_BF16: { typedef bf16_t T; CALL; } break;      \
        case GLC_DT_F16: { typedef f16_t T; CALL; } break;        \
        default: return "bad dtype";                              \
    }

const char* glc_launch_layernorm(hipStream_t st, int dtype, const void* X, void* Y, const float* gamma, const float* beta,
                                 float eps, int M, int H) {
    if (M <= 0 || !X || !Y || !gamma || !beta) return "layernorm: bad args";
    DISPATCH_T(dtype, {
        if (!h_ok<T>(H)) return "layernorm: unsupported hidden size";
        hipLaunchKernelGGL(layernorm_kernel<T>, dim3((M + 3) / 4), dim3(256), 0, st, (const T*)X, (T*)Y, gamma, beta, eps, M, H);
    });
    return nullptr;
}

const char* glc_launch_embed(hipStream_t st, int dtype, const int64_t* ids, const int64_t* mask, const void* table,
                             const float* gamma, const float* beta, float eps, void* X, float* kbias, int B, int S, int Sp,
                             int H, int vocab, int pad_id) {
    if (B <= 0 || S <= 0 || Sp < S || !ids || !mask || !table || !X || !kbias) return "embed: bad args";
    if (pad_id < 0 || pad_id >= vocab) return "embed: pad id outside vocab";
    DISPATCH_T(dtype, {
        if (!h_ok<T>(H)) return "embed: unsupported hidden size";
        hipLaunchKernelGGL(embed_kernel<T>, dim3((B * Sp + 3) / 4), dim3(256), 0, st, ids, mask, (const T*)table, gamma, beta, eps,
                           (T*)X, kbias, B, S, Sp, H, vocab, pad_id);
    });
    return nullptr;
}

const char* glc_launch_layernorm_gs(hipStream_t st, const float* X, void* Y, const float* gamma, const float* beta, float eps, int M, int H, int gx) {
    if (M <= 0 || !X || !Y || !gamma || !beta) return "layernorm_gs: bad args";
    if (H <= 0 || H % 32 || H / 8 > 64 * MAXC) return "layernorm_gs: unsupported hidden size";
    if (gx) hipLaunchKernelGGL(layernorm_gs_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, X, (f16_t*)Y, gamma, beta, eps, M, H, glc_gx_sat_ptr(), glc_gx_act_sc());
    else hipLaunchKernelGGL(layernorm_gs_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, X, (f16_t*)Y, gamma, beta, eps, M, H, (unsigned*)nullptr, 0);
    return nullptr;
}

const char* glc_launch_ln_stats(hipStream_t st, const float2* part, int nparts, float2* stats, int M, int H, float eps, int rms) {
    if (!part || !stats || M <= 0 || H <= 0 || nparts <= 0 || nparts > 1024 || nparts * 64 != H) return "ln_stats: bad args";
    hipLaunchKernelGGL(ln_stats_kernel, dim3((M + 255) / 256), dim3(256), 0, st, part, nparts, stats, M, 1.0 / (double)H, eps, rms);
    return nullptr;
}

const char* glc_launch_rows_to_gs_rms(hipStream_t st, const float* X, void* Y, float2* stats, float eps, int M, int H, int gx) {
    if (!X || !Y || !stats || M <= 0 || H <= 0 || H % 32) return "rows_to_gs_rms: bad args";
    if (gx) hipLaunchKernelGGL(rows_to_gs_rms_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, X, (f16_t*)Y, stats, eps, M, H, glc_gx_sat_ptr(), glc_gx_act_sc());
    else hipLaunchKernelGGL(rows_to_gs_rms_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, X, (f16_t*)Y, stats, eps, M, H, (unsigned*)nullptr, 0);
    return nullptr;
}

const char* glc_launch_embed_gs(hipStream_t st, const int64_t* ids, const int64_t* mask, const float* table, const float* gamma,
                                const float* beta, float eps, void* X, float* kbias, int B, int S, int Sp, int H, int vocab, int pad_id, int gx) {
    if (B <= 0 || S <= 0 || Sp < S || !ids || !mask || !table || !X || !kbias) return "embed_gs: bad args";
    if (pad_id < 0 || pad_id >= vocab) return "embed_gs: pad id outside vocab";
    if (H <= 0 || H % 32 || H / 8 > 64 * MAXC) return "embed_gs: unsupported hidden size";
    if (gx) hipLaunchKernelGGL(embed_gs_kernel<true>, dim3((B * Sp + 3) / 4), dim3(256), 0, st, ids, mask, table, gamma, beta, eps, (f16_t*)X, kbias, B, S, Sp, H, vocab, pad_id, glc_gx_sat_ptr(), glc_gx_act_sc());
    else hipLaunchKernelGGL(embed_gs_kernel<false>, dim3((B * Sp + 3) / 4), dim3(256), 0, st, ids, mask, table, gamma, beta, eps, (f16_t*)X, kbias, B, S, Sp, H, vocab, pad_id, (unsigned*)nullptr, 0);
    return nullptr;
}

const char* glc_launch_gather_rows_gs(hipStream_t st, const void* X, const int* cls_pos, int c_cap, float* Xs, int* sel_b, int* sel_q,
                                      unsigned char* tile_flag, int B, int Sp, int H, int C, int gx) {
    if (B <= 0 || C < 0 || !X || !cls_pos || !Xs || !sel_b || !sel_q || H % 32) return "gather_rows_gs: bad args";
    const int rows = B * (1 + C);
    if (gx) hipLaunchKernelGGL(gather_rows_gs_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, st, (const f16_t*)X, cls_pos, c_cap, Xs, sel_b, sel_q, tile_flag, B, Sp, H, C, glc_gx_act_sc());
    else hipLaunchKernelGGL(gather_rows_gs_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, st, (const f16_t*)X, cls_pos, c_cap, Xs, sel_b, sel_q, tile_flag, B, Sp, H, C, 0);
    return nullptr;
}

const char* glc_launch_scan_rows(hipStream_t st, const int64_t* ids, const int64_t* mask, int B, int S, int class_token,
                                 int embed_class_token, int* klen, int* kfirst, int* cls_pos, int* cls_cnt, int c_cap) {
    if (B <= 0 || S <= 0 || c_cap <= 0 || !ids || !mask || !klen || !kfirst || !cls_pos || !cls_cnt) return "scan_rows: bad args";
    hipLaunchKernelGGL(scan_rows_kernel, dim3(B), dim3(256), 0, st, ids, mask, S, class_token, embed_class_token, klen, kfirst, cls_pos,
                       cls_cnt, c_cap);
    return nullptr;
}

const char* glc_launch_head_gather(hipStream_t st, int dtype, const void* X, const int* cls_pos, int c_cap, float* Gt, float* Gc,
                                   int B, int Sp, int H, int C, const int* klen) {
    if (B <= 0 || C < 0 || !X || !cls_pos || !Gt || !Gc) return "head_gather: bad args";
    const int rows = B * (1 + C);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL(head_gather_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, st, (const T*)X, cls_pos, c_cap, Gt, Gc, B, Sp, H, C, klen);
    });
    return nullptr;
}

const char* glc_launch_pool_avg(hipStream_t st, int dtype, const void* X, const float* kbias, float* Gt, int B, int Sp, int H) {
    if (B <= 0 || Sp <= 0 || H % 64 || !X || !kbias || !Gt) return "pool_avg: bad args";
    DISPATCH_T(dtype, { hipLaunchKernelGGL(pool_avg_kernel<T>, dim3(H / 64, B), dim3(256), 0, st, (const T*)X, kbias, Gt, Sp, H); });
    return nullptr;
}

const char* glc_launch_gather_sel(hipStream_t st, int dtype, const void* src, const int* sel_b, const int* sel_q, void* dst, int R, int Sp, int H) {
    if (R <= 0 || !src || !sel_b || !sel_q || !dst || H % 8) return "gather_sel: bad args";
    DISPATCH_T(dtype, { hipLaunchKernelGGL(gather_sel_kernel<T>, dim3((R + 3) / 4), dim3(256), 0, st, (const T*)src, sel_b, sel_q, (T*)dst, R, Sp, H); });
    return nullptr;
}

const char* glc_launch_gather_rows(hipStream_t st, int dtype, const void* X, const int* cls_pos, int c_cap, void* Xs, int* sel_b,
                                   int* sel_q, unsigned char* tile_flag, int B, int Sp, int H, int C) {
    if (B <= 0 || C < 0 || !X || !cls_pos || !Xs || !sel_b || !sel_q || H % 8) return "gather_rows: bad args";
    const int rows = B * (1 + C);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL(gather_rows_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, st, (const T*)X, cls_pos, c_cap, (T*)Xs, sel_b, sel_q, tile_flag, B, Sp, H, C);
    });
    return nullptr;
}

const char* glc_launch_head_gather_sel(hipStream_t st, int dtype, const void* Xs, const int* cls_pos, int c_cap, float* Gt, float* Gc,
                                       int B, int H, int C) {
    if (B <= 0 || C < 0 || !Xs || !cls_pos || !Gt || !Gc) return "head_gather_sel: bad args";
    const int rows = B * (1 + C);
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL(head_gather_sel_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, st, (const T*)Xs, cls_pos, c_cap, Gt, Gc, B, H, C);
    });
    return nullptr;
}

const char* glc_launch_head_score(hipStream_t st, const float* Tt, const float* Cc, float* logits, int B, int C, int H,
                                  int normalize, float logit_scale) {
    if (B <= 0 || C <= 0 || !Tt || !Cc || !logits) return "head_score: bad args";
    hipLaunchKernelGGL(head_score_kernel, dim3((B * C + 3) / 4), dim3(256), 0, st, Tt, Cc, logits, B, C, H, normalize, logit_scale);
    return nullptr;
}

const char* glc_launch_scorer_wd_cat(hipStream_t st, const float* St, const float* Sc, float* cat, int B, int C, int H) {
    if (B <= 0 || C <= 0 || H <= 0 || !St || !Sc || !cat) return "scorer_wd_cat: bad args";
    hipLaunchKernelGGL(scorer_wd_cat_kernel, dim3(B * C), dim3(256), 0, st, St, Sc, cat, B, C, H);
    return nullptr;
}
const char* glc_launch_scorer_pair(hipStream_t st, const float* Tt, const float* Cc, float* out, int B, int C, int H) {
    if (B <= 0 || C <= 0 || H <= 0 || !Tt || !Cc || !out) return "scorer_pair: bad args";
    hipLaunchKernelGGL(scorer_pair_kernel, dim3(B * C), dim3(256), 0, st, Tt, Cc, out, B, C, H);
    return nullptr;
}
const char* glc_launch_relu(hipStream_t st, float* x, size_t n) {
    if (!x) return "relu: null";
    if (n) hipLaunchKernelGGL(relu_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, x, n);
    return nullptr;
}
const char* glc_launch_relu_dot(hipStream_t st, const float* X, const float* w, const float* bias, float* logits, int rows, int K, float scale) {
    if (rows <= 0 || K <= 0 || !X || !w || !bias || !logits) return "relu_dot: bad args";
    hipLaunchKernelGGL(relu_dot_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, X, w, bias, logits, rows, K, scale);
    return nullptr;
}
const char* glc_launch_l2norm_rows(hipStream_t st, float* X, int rows, int H) {
    if (rows <= 0 || H <= 0 || !X) return "l2norm_rows: bad args";
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, X, rows, H);
    return nullptr;
}

namespace {
__global__ __launch_bounds__(256) void presplit_kernel(float* __restrict__ w, size_t ngroups) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    f32x4 v[8];
    float* base = w + gi * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(base + 4 * i);      // the whole group is read before it is overwritten
    h8 hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[i][e];
            const _Float16 h = (_Float16)x;
            hi[i >> 1][4 * (i & 1) + e] = h;
            lo[i >> 1][4 * (i & 1) + e] = (_Float16)(x - (float)h);
        }
    h8* out = reinterpret_cast<h8*>(base);
#pragma unroll
    for (int i = 0; i < 4; ++i) { out[i] = hi[i]; out[4 + i] = lo[i]; }
}
}  // namespace

const char* glc_launch_presplit(hipStream_t st, void* w, size_t n) {
    if (!w || n % 32) return "presplit: element count must be a multiple of 32";
    const size_t groups = n / 32;
    if (groups) hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, (float*)w, groups);
    return nullptr;
}

const char* glc_launch_convert(hipStream_t st, int dtype, const float* src, void* dst, size_t n) {
    if (!src || !dst) return "convert: null";
    if (n == 0) return nullptr;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    DISPATCH_T(dtype, { hipLaunchKernelGGL(convert_kernel<T>, dim3(grid), dim3(256), 0, st, src, (T*)dst, n); });
    return nullptr;
}

const char* glc_launch_to_f32(hipStream_t st, int dtype, const void* src, float* dst, size_t n) {
    if (!src || !dst) return "to_f32: null";
    if (n == 0) return nullptr;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    DISPATCH_T(dtype, { hipLaunchKernelGGL(to_f32_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)src, dst, n); });
    return nullptr;
}
