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
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const T* __restrict__ GU, T* __restrict__ F, size_t M, int I) {
    typedef typename Vec16<T>::type vecT;
    constexpr int VEC = Vec16<T>::N;
    const size_t nch = (size_t)I / VEC, total = M * nch;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t m = idx / nch, ch = idx - m * nch;
        const vecT g = *reinterpret_cast<const vecT*>(GU + m * 2 * I + ch * VEC);
        const vecT u = *reinterpret_cast<const vecT*>(GU + m * 2 * I + I + ch * VEC);
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

const char* glc_launch_rope_qk(hipStream_t st, int dtype, void* QKV, const float* cs, int M, int Sp, int nq, int nkv, int d, float qscale) {
    if (M <= 0 || Sp <= 0 || !QKV || !cs || nq <= 0 || nkv <= 0 || d <= 0 || d % 2) return "rope: bad args";
    DISPATCH_T(dtype, { hipLaunchKernelGGL(rope_qk_kernel<T>, dim3((M + 3) / 4), dim3(256), 0, st, (T*)QKV, cs, M, Sp, nq, nkv, d, qscale); });
    return nullptr;
}

const char* glc_launch_swiglu(hipStream_t st, int dtype, const void* GU, void* F, size_t M, int I) {
    if (M == 0 || I <= 0 || I % 8 || !GU || !F) return "swiglu: bad args";
    DISPATCH_T(dtype, {
        const size_t total = M * ((size_t)I / Vec16<T>::N);
        size_t blocks = (total + 255) / 256;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL(swiglu_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)GU, (T*)F, M, I);
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
