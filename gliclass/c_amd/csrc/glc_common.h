// Shared device-side types and helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define GLC_WAVE 64
#define GLC_NEG_BIG (-1.0e30f)

// A 16-byte MFMA operand fragment, viewed per element type.
template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; static constexpr int N = 8; };
template <> struct Frag<f16_t> { typedef f16x8 type; static constexpr int N = 8; };

// 16x16 tile: D[i][j] += sum_k A[i][k] * B[k][j]; lane l supplies A[l&15][...] and B[...][l&15];
// result reg r of lane l is D[4*(l>>4)+r][l&15].  One 16-byte fragment per operand covers a K-span
// of 32 (16-bit types, one MFMA) or 16 (f32, four MFMAs with the k order permuted identically on
// both operands: MFMA t, lane group g <-> k = 4g+t).
__device__ __forceinline__ void mma16(const bf16x8& a, const bf16x8& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const f16x8& a, const f16x8& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma16(const f32x4& a, const f32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
}
// 32x32 tile, K-span 16: lane l supplies A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31];
// result reg r of lane l is D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
__device__ __forceinline__ void mma32(const bf16x8& a, const bf16x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma32(const f16x8& a, const f16x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// Attention operand fragment: 8 consecutive k-elements per lane for every T (one 32x32x16 MFMA for the 16-bit types; for fp32
// eight 32x32x2 MFMAs — MFMA j pairs k = 8*(l>>5) + j of the two lane halves, identically on both operands, so the fragment-major
// layouts of glc_layout.h serve fp32 unchanged and the 32x32 result layout is the same).
typedef __attribute__((ext_vector_type(8))) float f32x8;
template <typename T> struct AFrag;
template <> struct AFrag<bf16_t> { typedef bf16x8 type; };
template <> struct AFrag<f16_t> { typedef f16x8 type; };
template <> struct AFrag<float> { typedef f32x8 type; };
__device__ __forceinline__ void mma32(const f32x8& a, const f32x8& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// pack 4 floats -> 4 x T and store (8 B for 16-bit types, 16 B for f32)
template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) T v4;
    v4 v = {(T)a, (T)b, (T)c, (T)d};
    *reinterpret_cast<v4*>(p) = v;
}
template <typename T> __device__ __forceinline__ void load4(const T* p, float& a, float& b, float& c, float& d) {
    typedef __attribute__((ext_vector_type(4))) T v4;
    v4 v = *reinterpret_cast<const v4*>(p);
    a = (float)v[0]; b = (float)v[1]; c = (float)v[2]; d = (float)v[3];
}

// erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, i.e. at fp32 resolution for GELU's use)
__device__ __forceinline__ float glc_erf(float x) {
    float ax = fabsf(x);
    float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    float r = 1.0f - p * __expf(-ax * ax);
    return copysignf(r, x);
}
// HF ACT2FN["gelu"]: 0.5 x (1 + erf(x / sqrt 2))   (modeling_deberta_v2.py:393-396)
__device__ __forceinline__ float glc_gelu(float x) { return 0.5f * x * (1.0f + glc_erf(x * 0.70710678118654752f)); }

// Packed GELU on two values at once: the polynomial part runs on v_pk_*_f32 (two lanes' worth per issue),
// and GELU(x) = 0.5 (x + |x| erf(|x|/sqrt2)) needs no copysign.  Same Abramowitz-Stegun 7.1.26 erf as above.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 glc_gelu2(f32x2 x) {
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 z = ax * 0.70710678118654752f;
    const f32x2 d = z * 0.3275911f + 1.0f;
    f32x2 t;
    t[0] = __builtin_amdgcn_rcpf(d[0]); t[1] = __builtin_amdgcn_rcpf(d[1]);
    f32x2 pl = t * 1.061405429f + (-1.453152027f);
    pl = pl * t + 1.421413741f;
    pl = pl * t + (-0.284496736f);
    pl = pl * t + 0.254829592f;
    pl = pl * t;
    const f32x2 w = ax * 0.84932180028801904f;            // sqrt(log2(e) / 2): exp(-z^2) = exp2(-w^2)
    const f32x2 w2 = w * w;
    f32x2 ex;
    ex[0] = __builtin_amdgcn_exp2f(-w2[0]); ex[1] = __builtin_amdgcn_exp2f(-w2[1]);
    const f32x2 y = 1.0f - pl * ex;                        // erf(|x|/sqrt2)
    return (ax * y + x) * 0.5f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
