// Kernels of the decoder-style backbone (BASELINE.json configs[4]): Qwen2 arithmetic after
// transformers' models/qwen2/modeling_qwen2.py (cited as Q2:<line>).  The dense projections reuse the MFMA GEMMs
// (gemm256s.hip / gemm.hip); this file holds what the DeBERTa path does not have:
//   embed_plain   Q2:384      token gather, no norm / mask multiply; also writes the additive key bias
//   rmsnorm       Q2:247-252  w * x * rsqrt(mean(x^2) + eps), fp32 statistics, one wave per row
//   rope_qk       Q2:86-100, 105-109, 133-134  rotate-half RoPE applied in place to the Q and K columns of the fused
//                 QKV projection; the softmax scale log2(e)/sqrt(d) (Q2:186, :163) is folded into Q here
//   swiglu        Q2:47       silu(gate) * up on the fused [gate | up] projection
//   attn_gqa      Q2:160-170  grouped-query attention with causal and key-padding mask (create_causal_mask), exp2 softmax
// Row kernels are HBM streams: one wave per row, 16-byte vectors.
#include "glc_common.h"
#include "glc_kernels.h"

namespace {

template <typename T> struct Vec16 {
    static constexpr int N = 16 / (int)sizeof(T);
    typedef __attribute__((ext_vector_type(N))) T type;
};

template <typename T>
__global__ __launch_bounds__(256) void embed_plain_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask,
                                                          const T* __restrict__ table, T* __restrict__ X, float* __restrict__ kbias,
                                                          int B, int S, int Sp, int H, int vocab, int pad_id) {
    typedef typename Vec16<T>::type vecT;
    constexpr int VEC = Vec16<T>::N;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * Sp) return;
    const int b = row / Sp, s = row - b * Sp;
    long long id = pad_id;
    bool valid = false;
    if (s < S) {
        id = ids[(size_t)b * S + s];
        if (id < 0 || id >= vocab) id = pad_id;
        valid = mask[(size_t)b * S + s] != 0;
    }
    if (lane == 0) kbias[row] = valid ? 0.f : GLC_NEG_BIG;
    const T* src = table + (size_t)id * H;
    T* dst = X + (size_t)row * H;
    for (int ch = lane; ch < H / VEC; ch += 64) *reinterpret_cast<vecT*>(dst + (size_t)ch * VEC) = *reinterpret_cast<const vecT*>(src + (size_t)ch * VEC);
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ X, T* __restrict__ Y, const float* __restrict__ w, float eps,
                                                      int M, int H) {
    typedef typename Vec16<T>::type vecT;
    constexpr int VEC = Vec16<T>::N;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const T* x = X + (size_t)row * H;
    T* y = Y + (size_t)row * H;
    const int nch = H / VEC;
    float ss = 0.f;
    for (int ch = lane; ch < nch; ch += 64) {
        const vecT t = *reinterpret_cast<const vecT*>(x + (size_t)ch * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float v = (float)t[e]; ss += v * v; }
    }
    const float r = rsqrtf(wave_sum(ss) / (float)H + eps);
    for (int ch = lane; ch < nch; ch += 64) {          // second pass re-reads the row from L2 (H up to 8192 does not fit registers)
        const vecT t = *reinterpret_cast<const vecT*>(x + (size_t)ch * VEC);
        vecT o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (T)(w[ch * VEC + e] * ((float)t[e] * r));
        *reinterpret_cast<vecT*>(y + (size_t)ch * VEC) = o;
    }
}

// fp32 rows in, group-split rows out (the A operand of the 256-tile GS GEMM)
__global__ __launch_bounds__(256) void rmsnorm_gs_kernel(const float* __restrict__ X, f16_t* __restrict__ Y, const float* __restrict__ w, float eps,
                                                         int M, int H) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* x = X + (size_t)row * H;
    f16_t* y = Y + (size_t)row * 2 * H;
    const int nch = H / 8;
    float ss = 0.f;
    for (int ch = lane; ch < nch; ch += 64) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)ch * 8), b = *reinterpret_cast<const f32x4*>(x + (size_t)ch * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += a[e] * a[e] + b[e] * b[e];
    }
    const float r = rsqrtf(wave_sum(ss) / (float)H + eps);
    for (int ch = lane; ch < nch; ch += 64) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)ch * 8), b = *reinterpret_cast<const f32x4*>(x + (size_t)ch * 8 + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = w[ch * 8 + e] * (a[e] * r); o[4 + e] = w[ch * 8 + 4 + e] * (b[e] * r); }
        gs_store8(y, ch * 8, o);
    }
}
// F[m, i] = silu(GU[m, i]) * GU[m, I + i]: plain fp32 [gate | up] rows in, group-split rows out
__global__ __launch_bounds__(256) void swiglu_gs_kernel(const float* __restrict__ GU, f16_t* __restrict__ F, size_t M, int I) {
    const size_t nch = (size_t)I / 8, total = M * nch;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t m = idx / nch, ch = idx - m * nch;
        const float* g = GU + m * 2 * I + ch * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(g), g1 = *reinterpret_cast<const f32x4*>(g + 4);
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(g + I), u1 = *reinterpret_cast<const f32x4*>(g + I + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = g0[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-g0[e])) * u0[e];
            o[4 + e] = g1[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-g1[e])) * u1[e];
        }
        gs_store8(F + m * 2 * I, (int)(ch * 8), o);
    }
}

// In place on QKV [M, (nq + 2 nkv) d]: heads 0..nq-1 are Q (also scaled by qscale), nq..nq+nkv-1 are K.  cs = [Sp][d/2][2].
template <typename T>
__global__ __launch_bounds__(256) void rope_qk_kernel(T* __restrict__ QKV, const float* __restrict__ cs, int M, int Sp, int nq, int nkv,
                                                      int d, float qscale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int s = row % Sp, hd2 = d >> 1, ld = (nq + 2 * nkv) * d;
    T* base = QKV + (size_t)row * ld;
    const float* c = cs + (size_t)s * hd2 * 2;
    const int npair = (nq + nkv) * hd2;
    for (int p = lane; p < npair; p += 64) {
        const int h = p / hd2, i = p - h * hd2;
        const float co = c[2 * i], sn = c[2 * i + 1];
        const float x1 = (float)base[h * d + i], x2 = (float)base[h * d + i + hd2];
        const float sc = h < nq ? qscale : 1.f;
        base[h * d + i] = (T)((x1 * co - x2 * sn) * sc);
        base[h * d + i + hd2] = (T)((x2 * co + x1 * sn) * sc);
    }
}

// F[m, i] = silu(GU[m, i]) * GU[m, I + i]
// inter = 1: the [gate | up] columns interleave 16 gate / 16 up features (the fused weight layout of the SwiGLU GEMM epilogue,
// engine.hip): feature i's gate is column 32 (i / 16) + i % 16, its up column 16 further on
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const T* __restrict__ GU, T* __restrict__ F, size_t M, int I, int inter) {
    typedef typename Vec16<T>::type vecT;
    constexpr int VEC = Vec16<T>::N;
    const size_t nch = (size_t)I / VEC, total = M * nch;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t m = idx / nch, ch = idx - m * nch;
        const size_t f0 = ch * VEC;
        const size_t gcol = inter ? 32 * (f0 / 16) + f0 % 16 : f0, ucol = inter ? gcol + 16 : (size_t)I + f0;
        const vecT g = *reinterpret_cast<const vecT*>(GU + m * 2 * I + gcol);
        const vecT u = *reinterpret_cast<const vecT*>(GU + m * 2 * I + ucol);
        vecT o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float gv = (float)g[e];
            o[e] = (T)(gv * __builtin_amdgcn_rcpf(1.0f + __expf(-gv)) * (float)u[e]);
        }
        *reinterpret_cast<vecT*>(F + m * I + ch * VEC) = o;
    }
}

// Straightforward grouped-query attention (any T, no MFMA): one block per (query row, head).  Q already carries
// log2(e)/sqrt(d); scores are in log2 units.  Keys j > q are excluded when causal; padded keys carry the -1e30 bias.
template <typename T>
__global__ __launch_bounds__(256) void attn_gqa_simple_kernel(const T* __restrict__ QKV, const float* __restrict__ kbias,
                                                              const int* __restrict__ klen, T* __restrict__ CTX, int Sp, int nq, int nkv,
                                                              int d, int causal) {
    typedef typename Vec16<T>::type vecT;
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qv = sm;                 // [d]
    float* red = sm + d;            // [256]
    float* sc = sm + d + 256;       // [Sp]
    const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z, t = threadIdx.x;
    const int ld = (nq + 2 * nkv) * d, kvh = h / (nq / nkv);
    const T* Qr = QKV + ((size_t)b * Sp + q) * ld + h * d;
    const T* Kb = QKV + (size_t)b * Sp * ld + (size_t)(nq + kvh) * d;
    const T* Vb = QKV + (size_t)b * Sp * ld + (size_t)(nq + nkv + kvh) * d;
    const float* kb = kbias + (size_t)b * Sp;
    for (int i = t; i < d; i += 256) qv[i] = (float)Qr[i];
    __syncthreads();
    int kend = klen[b];
    if (causal && q + 1 < kend) kend = q + 1;
    if (kend < 1) kend = 1;
    if (kend > Sp) kend = Sp;
    float mx = -3.0e38f;
    for (int k = t; k < kend; k += 256) {
        const T* kr = Kb + (size_t)k * ld;
        float s = 0.f;
        for (int e0 = 0; e0 < d; e0 += VEC) {
            const vecT kv = *reinterpret_cast<const vecT*>(kr + e0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += qv[e0 + e] * (float)kv[e];
        }
        s += kb[k];
        sc[k] = s;
        mx = fmaxf(mx, s);
    }
    red[t] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmaxf(red[t], red[t + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int k = t; k < kend; k += 256) { const float p = __builtin_amdgcn_exp2f(sc[k] - mx); sc[k] = p; sum += p; }
    red[t] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const float inv = 1.0f / red[0];
    __syncthreads();
    // O[dd] = sum_k p_k V[k][dd]: 256 threads = (256/d) key phases x d columns (d = 64 or 128)
    const int dd = t % d, part = t / d, nparts = 256 / d;
    float acc = 0.f;
    for (int k = part; k < kend; k += nparts) acc += sc[k] * (float)Vb[(size_t)k * ld + dd];
    red[t] = acc;
    __syncthreads();
    if (t < d) {
        float v = 0.f;
        for (int p = 0; p < nparts; ++p) v += red[t + p * d];
        CTX[((size_t)b * Sp + q) * ((size_t)nq * d) + h * d + t] = (T)(v * inv);
    }
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                   \
    switch (dtype) {                                              \
        case GLC_DT_F32: { typedef float T; CALL; } break;        \
        case GLC_DT_BF16: { typedef bf16_t T; CALL; } break;      \
        case GLC_DT_F16: { typedef f16_t T; CALL; } break;        \
        default: return "bad dtype";                              \
    }

const char* glc_launch_embed_plain(hipStream_t st, int dtype, const int64_t* ids, const int64_t* mask, const void* table, void* X,
                                   float* kbias, int B, int S, int Sp, int H, int vocab, int pad_id) {
    if (B <= 0 || S <= 0 || Sp < S || !ids || !mask || !table || !X || !kbias) return "embed_plain: bad args";
    if (pad_id < 0 || pad_id >= vocab || H % 8) return "embed_plain: bad pad id or hidden size";
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL(embed_plain_kernel<T>, dim3((B * Sp + 3) / 4), dim3(256), 0, st, ids, mask, (const T*)table, (T*)X, kbias, B, S,
                           Sp, H, vocab, pad_id);
    });
    return nullptr;
}

const char* glc_launch_rmsnorm(hipStream_t st, int dtype, const void* X, void* Y, const float* w, float eps, int M, int H) {
    if (M <= 0 || !X || !Y || !w || H % 8) return "rmsnorm: bad args";
    DISPATCH_T(dtype, { hipLaunchKernelGGL(rmsnorm_kernel<T>, dim3((M + 3) / 4), dim3(256), 0, st, (const T*)X, (T*)Y, w, eps, M, H); });
    return nullptr;
}

const char* glc_launch_rmsnorm_gs(hipStream_t st, const float* X, void* Y, const float* w, float eps, int M, int H) {
    if (!X || !Y || !w || M <= 0 || H <= 0 || H % 32) return "rmsnorm_gs: bad args";
    hipLaunchKernelGGL(rmsnorm_gs_kernel, dim3((M + 3) / 4), dim3(256), 0, st, X, (f16_t*)Y, w, eps, M, H);
    return nullptr;
}
const char* glc_launch_swiglu_gs(hipStream_t st, const float* GU, void* F, size_t M, int I) {
    if (!GU || !F || M == 0 || I <= 0 || I % 32) return "swiglu_gs: bad args";
    const size_t total = M * ((size_t)I / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(swiglu_gs_kernel, dim3(grid), dim3(256), 0, st, GU, (f16_t*)F, M, I);
    return nullptr;
}

const char* glc_launch_rope_qk(hipStream_t st, int dtype, void* QKV, const float* cs, int M, int Sp, int nq, int nkv, int d, float qscale) {
    if (M <= 0 || Sp <= 0 || !QKV || !cs || nq <= 0 || nkv <= 0 || d <= 0 || d % 2) return "rope: bad args";
    DISPATCH_T(dtype, { hipLaunchKernelGGL(rope_qk_kernel<T>, dim3((M + 3) / 4), dim3(256), 0, st, (T*)QKV, cs, M, Sp, nq, nkv, d, qscale); });
    return nullptr;
}

const char* glc_launch_swiglu(hipStream_t st, int dtype, const void* GU, void* F, size_t M, int I, int inter) {
    if (M == 0 || I <= 0 || I % 8 || !GU || !F) return "swiglu: bad args";
    DISPATCH_T(dtype, {
        const size_t total = M * ((size_t)I / Vec16<T>::N);
        size_t blocks = (total + 255) / 256;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL(swiglu_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)GU, (T*)F, M, I, inter);
    });
    return nullptr;
}

// impl: 1 = straightforward kernel (any T)
const char* glc_launch_attention_gqa(hipStream_t st, int dtype, int impl, const void* QKV, const float* kbias, const int* klen, void* CTX,
                                     int B, int Sp, int nq, int nkv, int d, int causal) {
    if (!QKV || !kbias || !klen || !CTX || B <= 0 || Sp <= 0 || nq <= 0 || nkv <= 0 || nq % nkv) return "attention_gqa: bad args";
    if (d != 64 && d != 128) return "attention_gqa: head_dim must be 64 or 128";
    if (impl != 1) return "attention_gqa: unknown implementation";
    const size_t shm = (size_t)(d + 256 + Sp) * sizeof(float);
    if (shm > 64 * 1024) return "attention_gqa(simple): sequence too long for the straightforward kernel";
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL(attn_gqa_simple_kernel<T>, dim3(Sp, nq, B), dim3(256), shm, st, (const T*)QKV, kbias, klen, (T*)CTX, Sp, nq, nkv, d,
                           causal);
    });
    return nullptr;
}

// =====================================================================================================================
// MFMA path of the decoder attention (16-bit operands).
//
// qkv_layout: one pass over the row-major fused projection that applies RoPE (+ the softmax scale on Q) and writes the
// operands in the fragment-major order the 32x32x16 MFMA consumes (the same idea as glc_layout.h, head_dim D in {64,128}):
//   Qf : [b*nq + h ][tile32][s < D/16][lane = 32 hh + r][8]   = Q [32 tile + r    ][16 s + 8 hh + j]
//   Kf : [b*nkv + g][tile32][s < D/16][lane = 32 hh + r][8]   = K [32 tile + pi(r)][16 s + 8 hh + j]      pi = swap(bit2, bit3)
//   Vt : [b*nkv + g][tile32][dt < D/32][t < 2][lane = 32 hh + r][8] = V^T[32 dt + r][32 tile + 16 t + 8 hh + j]
// so every wave-level 16-byte-per-lane load of the attention kernel is one contiguous 1 KiB.
//
// attn_gqa_mfma: flash-style, one wave = 32 queries of one query head, 4 waves (4 consecutive query tiles) per block, no
// LDS and no workgroup barrier:  S^T = K Q^T (D/16 MFMA 32x32x16), online softmax in log2 units with the deferred rescale of
// attention.hip, O^T += V^T P^T (D/16 MFMA, P^T taken straight from the S^T accumulators thanks to the pi row order of K).
// Causal: key tiles above the diagonal are never visited, the diagonal tile masks key > query per element; padded keys
// carry the additive -1e30 bias on tiles at / after the first masked key.  Grouped queries: head h reads K/V of group
// h / (nq/nkv); the grid keeps all query heads of one (batch, group) on one XCD so they share that group's K/V in L2.
// =====================================================================================================================
#include "glc_layout.h"

namespace {

// SPLIT (T = float, the fp32 mode): the outputs are split-f16 units — the 32 bytes a lane owns of an 8-element unit hold
// [8 hi halves | 8 lo halves] (glc_common.h f16x8s), at the addresses of the fp32 fragment-major layout.
template <typename T, bool SPLIT> __device__ __forceinline__ void store_unit8(T* dst, const float (&v)[8]) {
    if constexpr (SPLIT) {
        f16x8s u;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const f16_t hv = (f16_t)v[j]; u.hi[j] = hv; u.lo[j] = (f16_t)(v[j] - (float)hv); }
        *reinterpret_cast<f16x8s*>(dst) = u;
    } else {
        typedef __attribute__((ext_vector_type(8))) T vec8;
        vec8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)v[j];
        *reinterpret_cast<vec8*>(dst) = o;
    }
}

template <typename T, int D, bool SPLIT = false>
__global__ __launch_bounds__(256) void qkv_layout_kernel(const T* __restrict__ QKV, const float* __restrict__ cs, T* __restrict__ Qf,
                                                         T* __restrict__ Kf, T* __restrict__ Vt, int Sp, int nq, int nkv, float qscale) {
    static_assert(!SPLIT || sizeof(T) == 4, "split units live in the fp32 layouts");
    typedef __attribute__((ext_vector_type(8))) T vec8;
    constexpr int HD2 = D / 2, NS = D / 16;
    __shared__ T vs[32][D + 8];
    const int tile = blockIdx.x, head = blockIdx.y, t = threadIdx.x;           // tile over all B*Sp/32 row tiles
    const int ld = (nq + 2 * nkv) * D;
    const int m0 = tile * 32, b = m0 / Sp, st = (m0 - b * Sp) >> 5, nt = Sp >> 5;
    if (head < nq + nkv) {
        // ---- Q or K head: RoPE on the (i, i + D/2) pairs, 8 consecutive i per thread ----
        const bool isq = head < nq;
        for (int idx = t; idx < 32 * (HD2 / 8); idx += 256) {
            const int r = idx / (HD2 / 8), c8 = idx - r * (HD2 / 8);           // row in tile, chunk of 8 in the first half
            const int s = (m0 - b * Sp) + r;
            const T* src = QKV + (size_t)(m0 + r) * ld + (size_t)head * D + c8 * 8;
            const vec8 a = *reinterpret_cast<const vec8*>(src);
            const vec8 bq = *reinterpret_cast<const vec8*>(src + HD2);
            const float* c = cs + ((size_t)s * HD2 + c8 * 8) * 2;
            const float sc = isq ? qscale : 1.f;
            float o1[8], o2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float co = c[2 * j], sn = c[2 * j + 1], x1 = (float)a[j], x2 = (float)bq[j];
                o1[j] = (x1 * co - x2 * sn) * sc;
                o2[j] = (x2 * co + x1 * sn) * sc;
            }
            const int lr = isq ? r : glc_pi32(r);                              // K rows sit at lane pi(r)
            T* dst = isq ? Qf + (((size_t)(b * nq + head) * nt + st) * NS) * 512 : Kf + (((size_t)(b * nkv + (head - nq)) * nt + st) * NS) * 512;
            const int d1 = c8 * 8, d2 = d1 + HD2;
            store_unit8<T, SPLIT>(dst + ((size_t)(d1 >> 4) * 64 + 32 * ((d1 >> 3) & 1) + lr) * 8, o1);
            store_unit8<T, SPLIT>(dst + ((size_t)(d2 >> 4) * 64 + 32 * ((d2 >> 3) & 1) + lr) * 8, o2);
        }
    } else {
        // ---- V head: transpose the 32 x D tile through LDS ----
        const int g = head - nq - nkv;
        for (int idx = t; idx < 32 * (D / 8); idx += 256) {
            const int r = idx / (D / 8), c8 = idx - r * (D / 8);
            *reinterpret_cast<vec8*>(&vs[r][c8 * 8]) = *reinterpret_cast<const vec8*>(QKV + (size_t)(m0 + r) * ld + (size_t)(nq + nkv + g) * D + c8 * 8);
        }
        __syncthreads();
        T* dst = Vt + ((size_t)(b * nkv + g) * nt + st) * (size_t)(D / 32) * 2 * 512;
        for (int u = t; u < (D / 32) * 2 * 64; u += 256) {                     // unit = (dt, tt, lane)
            const int lane = u & 63, tt = (u >> 6) & 1, dt = u >> 7;
            const int dd = 32 * dt + (lane & 31), k0 = 16 * tt + 8 * (lane >> 5);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (float)vs[k0 + j][dd];
            store_unit8<T, SPLIT>(dst + (size_t)u * 8, o);
        }
    }
}

constexpr float DEC_RESCALE_THR = 8.0f;   // log2 units (attention.hip)

template <bool SPLIT, typename T> struct GqaFrag { typedef typename Frag<T>::type type; };
template <typename T> struct GqaFrag<true, T> { typedef f16x8s type; };
// SPLIT (T = float): operands are split-f16 units, every product is three f16 MFMAs (glc_common.h), the probabilities are split on
// the fly; one wave per SIMD (the doubled fragment sets need > 256 registers at D = 128).
template <typename T, int D, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? 1 : 2) void attn_gqa_mfma_kernel(const T* __restrict__ Qf, const T* __restrict__ Kf, const T* __restrict__ Vt,
                                                               const float* __restrict__ kbias, const int* __restrict__ klen,
                                                               const int* __restrict__ kfirst_, T* __restrict__ CTX, int B, int Sp, int nq,
                                                               int nkv, int causal, int ctx_gs, unsigned* gx_sat, int act_sc) {
    static_assert(!SPLIT || sizeof(T) == 4, "split units live in the fp32 layouts");
    typedef typename GqaFrag<SPLIT, T>::type frag_t;
    constexpr int NS = D / 16, ND = D / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int nt = Sp >> 5, nqb = (nt + 3) >> 2, grp = nq / nkv;
    // XCD-aware decode: blocks b and b + 8 share an XCD; give every block of one (batch, kv group) the same id % 8
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int per = grp * nqb;                               // blocks per (batch, group)
    const int bg = xcd + 8 * (jj / per), rem = jj % per;
    if (bg >= B * nkv) return;
    const int b = bg / nkv, g = bg - b * nkv;
    const int hq = g * grp + rem / nqb;
    // longest tiles first within a head (causal work grows with the tile index)
    const int qt = nt - 1 - ((rem % nqb) * 4 + wave);
    if (qt < 0) return;
    const int q0 = qt * 32;

    const T* __restrict__ Qp = Qf + (((size_t)(b * nq + hq) * nt + qt) * NS) * 512 + lane * 8;
    const T* __restrict__ Kp = Kf + ((size_t)(b * nkv + g) * nt * NS) * 512 + lane * 8;
    const T* __restrict__ Vp = Vt + ((size_t)(b * nkv + g) * nt * ND * 2) * 512 + lane * 8;
    const float* __restrict__ kb = kbias + (size_t)b * Sp;

    if (q0 >= klen[b] && q0 > 0) {           // padding-only query tile of a ragged batch: no attended row reads it; store zeros
        T* outz = CTX + ((size_t)b * Sp + q0 + c) * ((size_t)nq * D) + (size_t)hq * D;
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) store4<T>(outz + 32 * a + 8 * gq + 4 * h, 0.f, 0.f, 0.f, 0.f);
        return;
    }
    int nkt = (klen[b] + 31) >> 5;
    nkt = nkt < 1 ? 1 : (nkt > nt ? nt : nkt);
    if (causal && nkt > qt + 1) nkt = qt + 1;
    const int kfirst = kfirst_[b];
    const int foff = 8 * h;

    frag_t qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const frag_t*>(Qp + s * 512);
    f32x16 o[ND];
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[a][i] = 0.f;
    float m = -3.0e38f, l = 0.f;

    frag_t kf[NS];                   // ONE K set: the next tile is loaded into it right after its last MFMA of this tile has issued
#pragma unroll
    for (int s = 0; s < NS; ++s) kf[s] = *reinterpret_cast<const frag_t*>(Kp + s * 512);
    for (int kt = 0; kt < nkt; ++kt) {
        const int ktn = kt + 1 < nkt ? kt + 1 : kt;
        frag_t vt[ND][2];
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int t = 0; t < 2; ++t) vt[a][t] = *reinterpret_cast<const frag_t*>(Vp + ((size_t)kt * ND * 2 + a * 2 + t) * 512);
        // S^T = K Q^T ; reg i <-> key k0 + 16*(i>>3) + 8h + (i&7), column = query c
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) mma32(kf[s], qf[s], sacc);
#pragma unroll
        for (int s = 0; s < NS; ++s) kf[s] = *reinterpret_cast<const frag_t*>(Kp + (size_t)ktn * NS * 512 + s * 512);
        float sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = sacc[i];
        const int k0 = kt * 32;
        if (k0 + 32 > kfirst) {                                             // wave-uniform: tile holds masked keys
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(kb + k0 + foff);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(kb + k0 + foff + 4);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff);
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sv[i] += b0[i]; sv[4 + i] += b1[i]; sv[8 + i] += b2[i]; sv[12 + i] += b3[i]; }
        }
        if (causal && kt == qt) {                                           // diagonal tile: key offset > query offset is masked
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ko = 16 * (i >> 3) + foff + (i & 7);
                if (ko > c) sv[i] = GLC_NEG_BIG;
            }
        }
        float mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, sv[i]), sv[i + 1]);
        mx = fmaxf(mx, sv[15]);
        if (__builtin_amdgcn_ballot_w64(mx - m > DEC_RESCALE_THR) != 0ull) {   // deferred rescale (attention.hip)
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            l *= alpha;
#pragma unroll
            for (int a = 0; a < ND; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[a][i] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(sv[i] - m); psum += sv[i]; }
        l += psum;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            frag_t pfr;
            if constexpr (SPLIT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const f16_t ph = (f16_t)sv[8 * t + j]; pfr.hi[j] = ph; pfr.lo[j] = (f16_t)(sv[8 * t + j] - (float)ph); }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) pfr[j] = (T)sv[8 * t + j];
            }
#pragma unroll
            for (int a = 0; a < ND; ++a) mma32(vt[a][t], pfr, o[a]);        // O^T[dd = 32a + (i&3) + 8(i>>2) + 4h][query c]
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    T* out = CTX + ((size_t)b * Sp + q0 + c) * ((size_t)nq * D) + (size_t)hq * D;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int a = 0; a < ND; ++a) store_acc32_wide<T>(o[a], inv, out + 32 * a, h);      // 16-byte stores via v_permlane32_swap (glc_common.h)
    } else if (ctx_gs == 2) {
        // GX context rows (glc_common.h; the A operand of the MX cross-term GEMM): as attention_wg.hip, one cross-half exchange per register
        unsigned char* row = reinterpret_cast<unsigned char*>(CTX) + ((size_t)b * Sp + q0 + c) * 4 * ((size_t)nq * D);
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float own_a = o[a][8 * p + e] * inv, own_b = o[a][8 * p + 4 + e] * inv;
                    const float got = __shfl_xor(h ? own_a : own_b, 32, 64);
                    v[e] = h ? got : own_a;
                    v[4 + e] = h ? own_b : got;
                }
                gx_store8(row, hq * D + 32 * a + 16 * p + 8 * h, v, gx_act_khi(act_sc), gx_act_klo(act_sc), gx_sat);
            }
    } else if (ctx_gs) {
        // group-split context rows: block a of this head is group hq * D/32 + a of the row; hi = f16(v), lo = f16(v - hi)
        f16_t* row = reinterpret_cast<f16_t*>(CTX) + ((size_t)b * Sp + q0 + c) * 2 * ((size_t)nq * D) + (size_t)(hq * (D / 32)) * 64;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            f32x16 lo;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float v = o[a][i] * inv; lo[i] = v - (float)(f16_t)v; }
            store_acc32_wide<f16_t>(o[a], inv, row + 64 * a, h);
            store_acc32_wide<f16_t>(lo, 1.0f, row + 64 * a + 32, h);
        }
    } else {
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                store4<T>(out + 32 * a + 8 * gq + 4 * h, o[a][4 * gq] * inv, o[a][4 * gq + 1] * inv, o[a][4 * gq + 2] * inv, o[a][4 * gq + 3] * inv);
    }
}

template <typename T, bool SPLIT = false> const char* launch_layout_t(hipStream_t st, const void* QKV, const float* cs, void* Qf, void* Kf, void* Vt, int B, int Sp,
                                                  int nq, int nkv, int d, float qscale) {
    const dim3 grid(B * Sp / 32, nq + 2 * nkv), block(256);
    if (d == 128) hipLaunchKernelGGL((qkv_layout_kernel<T, 128, SPLIT>), grid, block, 0, st, (const T*)QKV, cs, (T*)Qf, (T*)Kf, (T*)Vt, Sp, nq, nkv, qscale);
    else hipLaunchKernelGGL((qkv_layout_kernel<T, 64, SPLIT>), grid, block, 0, st, (const T*)QKV, cs, (T*)Qf, (T*)Kf, (T*)Vt, Sp, nq, nkv, qscale);
    return nullptr;
}
template <typename T, bool SPLIT = false> const char* launch_gqa_t(hipStream_t st, const void* Qf, const void* Kf, const void* Vt, const float* kbias, const int* klen,
                                               const int* kfirst, void* CTX, int B, int Sp, int nq, int nkv, int d, int causal, int ctx_gs = 0) {
    const int nt = Sp / 32, nqb = (nt + 3) / 4, per = (nq / nkv) * nqb, bg8 = (B * nkv + 7) / 8 * 8;
    const dim3 grid(per * bg8), block(256);
    if (d == 128) hipLaunchKernelGGL((attn_gqa_mfma_kernel<T, 128, SPLIT>), grid, block, 0, st, (const T*)Qf, (const T*)Kf, (const T*)Vt, kbias, klen, kfirst, (T*)CTX, B, Sp, nq, nkv, causal, ctx_gs, ctx_gs == 2 ? glc_gx_sat_ptr() : nullptr, glc_gx_act_sc());
    else hipLaunchKernelGGL((attn_gqa_mfma_kernel<T, 64, SPLIT>), grid, block, 0, st, (const T*)Qf, (const T*)Kf, (const T*)Vt, kbias, klen, kfirst, (T*)CTX, B, Sp, nq, nkv, causal, ctx_gs, ctx_gs == 2 ? glc_gx_sat_ptr() : nullptr, glc_gx_act_sc());
    return nullptr;
}

}  // namespace

// RoPE + softmax scale + fragment-major layout of the fused projection (16-bit T).  Sp % 64 == 0, d in {64, 128}.
const char* glc_launch_qkv_layout(hipStream_t st, int dtype, const void* QKV, const float* cs, void* Qf, void* Kf, void* Vt, int B, int Sp,
                                  int nq, int nkv, int d, float qscale) {
    if (!QKV || !cs || !Qf || !Kf || !Vt || B <= 0 || Sp <= 0 || Sp % 64 || nq <= 0 || nkv <= 0 || (d != 64 && d != 128)) return "qkv_layout: bad args";
    if (dtype == GLC_DT_BF16) return launch_layout_t<bf16_t>(st, QKV, cs, Qf, Kf, Vt, B, Sp, nq, nkv, d, qscale);
    if (dtype == GLC_DT_F16) return launch_layout_t<f16_t>(st, QKV, cs, Qf, Kf, Vt, B, Sp, nq, nkv, d, qscale);
    if (dtype == GLC_DT_F32) return launch_layout_t<float, true>(st, QKV, cs, Qf, Kf, Vt, B, Sp, nq, nkv, d, qscale);    // fp32 data -> split-f16 units
    return "qkv_layout: bad dtype";
}

// MFMA grouped-query attention on the fragment-major operands written by glc_launch_qkv_layout.  CTX [B*Sp, nq*d] row-major.
const char* glc_launch_attention_gqa_mfma(hipStream_t st, int dtype, const void* Qf, const void* Kf, const void* Vt, const float* kbias,
                                          const int* klen, const int* kfirst, void* CTX, int B, int Sp, int nq, int nkv, int d, int causal, int ctx_gs) {
    if (!Qf || !Kf || !Vt || !kbias || !klen || !kfirst || !CTX || B <= 0 || Sp <= 0 || Sp % 64 || nq <= 0 || nkv <= 0 || nq % nkv ||
        (d != 64 && d != 128))
        return "attention_gqa_mfma: bad args";
    if (dtype == GLC_DT_BF16) return launch_gqa_t<bf16_t>(st, Qf, Kf, Vt, kbias, klen, kfirst, CTX, B, Sp, nq, nkv, d, causal);
    if (dtype == GLC_DT_F16) return launch_gqa_t<f16_t>(st, Qf, Kf, Vt, kbias, klen, kfirst, CTX, B, Sp, nq, nkv, d, causal);
    if (dtype == GLC_DT_F32) return launch_gqa_t<float, true>(st, Qf, Kf, Vt, kbias, klen, kfirst, CTX, B, Sp, nq, nkv, d, causal, ctx_gs);   // split-f16 units
    return "attention_gqa_mfma: bad dtype";
}
