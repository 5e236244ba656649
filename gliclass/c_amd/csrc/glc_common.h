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

// Split-f16 attention operand (fp32 mode): the 32 bytes a lane holds of an fp32 fragment unit carry [8 hi halves | 8 lo halves] of
// the same 8 k-elements (x = hi + lo, hi = f16(x), lo = f16(x - hi)), so every address of the fp32 fragment-major layout stays
// valid; a product is three f16 MFMAs a_lo*b_hi + a_hi*b_lo + a_hi*b_hi with fp32 accumulation (lo*lo < 2^-22 of the product).
struct f16x8s { f16x8 hi, lo; };
__device__ __forceinline__ void mma32(const f16x8s& a, const f16x8s& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
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

// Epilogue store of one 32x32 MFMA accumulator o = O^T[dd][query] scaled by `inv`, 16-bit T: a query row's 32 outputs are split between
// lane c (h = 0: columns 8g..8g+3) and lane c + 32 (columns 8g+4..8g+7).  One v_permlane32_swap per dword and pair of groups (2p, 2p+1)
// hands lane c the whole 16 bytes of group 2p and lane c + 32 those of group 2p+1: two 16-byte stores per lane instead of four 8-byte
// ones (the store tail of an attention wave is bound by store INSTRUCTIONS, not bytes: -1 % on the band kernel).  Same values, same
// rounding, same addresses as store4 per group.  `base` = this query row's first of the 32 columns.
template <typename T> __device__ __forceinline__ void store_acc32_wide(const f32x16& o, float inv, T* base, int h) {
    static_assert(sizeof(T) == 2, "16-bit outputs");
    typedef __attribute__((ext_vector_type(2))) T T2;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const T2 a0 = {(T)(o[8 * p] * inv), (T)(o[8 * p + 1] * inv)}, a1 = {(T)(o[8 * p + 2] * inv), (T)(o[8 * p + 3] * inv)};          // group 2p
        const T2 b0 = {(T)(o[8 * p + 4] * inv), (T)(o[8 * p + 5] * inv)}, b1 = {(T)(o[8 * p + 6] * inv), (T)(o[8 * p + 7] * inv)};      // group 2p + 1
        const auto rx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, b0), false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1), false, false);
        const u32x4 v = {rx[0], ry[0], rx[1], ry[1]};      // h = 0: [own 2p | partner's 2p]; h = 1: [partner's 2p+1 | own 2p+1]
        *reinterpret_cast<u32x4*>(base + 16 * p + 8 * h) = v;
    }
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

// GELU for the 16-bit GEMM epilogue, two values at once.  gelu(x) = x * Phi(x) with Phi(x) = sigmoid(x * (c1 + c3 x^2 + c5 x^4)):
// an odd-polynomial logit fitted to the normal CDF (scripts-free derivation: minimax-weighted least squares on [-9, 9]); maximum
// absolute deviation from the erf form 2.6e-5 over all x, i.e. below one f16 ulp of the result for |gelu| >= 0.05 and 20x below
// the 16-bit modes' operand noise.  7 VALU + 2 transcendentals per element instead of 14 + 2 for Abramowitz-Stegun 7.1.26 —
// the epilogue's GELU was 30 % of the FFN1 kernel's time (SQ counters).  x^2 is clamped at 64 (Phi(8) = 1 - 6e-16) so the x^4
// term cannot turn the logit around for huge |x|; exp2 overflow gives rcp(inf) = 0, the correct limit.  The fp32 path
// (gemm.hip) keeps glc_gelu above.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 glc_gelu2(f32x2 x) {
    f32x2 x2 = x * x;
    x2[0] = fminf(x2[0], 64.0f); x2[1] = fminf(x2[1], 64.0f);
    f32x2 p = x2 * 0.0010142675173720906f + (-0.10677575394624186f);      // -log2(e) * (c5 x^4 + c3 x^2 + c1)
    p = p * x2 + (-2.3011212982378404f);
    const f32x2 u = x * p;
    f32x2 d;
    d[0] = 1.0f + __builtin_amdgcn_exp2f(u[0]); d[1] = 1.0f + __builtin_amdgcn_exp2f(u[1]);
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(d[0]); r[1] = __builtin_amdgcn_rcpf(d[1]);
    return x * r;
}

// GELU for the fp32 mode's 256-tile GEMM epilogues, two values at once, at fp32 resolution: the same form, Phi(x) = 1 / (1 + 2^(x P(x^2))), with
// P of degree 6 in x^2 (weighted minimax fit of log2((1 - Phi) / Phi) / x on x^2 <= 36, weight d gelu / d P): |gelu - x Phi(x)| <= 6.7e-8 in
// exact arithmetic and 5.9e-7 evaluated in fp32 over [-14, 14] (one ulp of the result near x = 5; glc_gelu's Abramowitz-Stegun form in fp32:
// 4.6e-7).  ~6 packed VALU + 2 transcendentals per element instead of ~16 VALU + 2: the epilogue is VALU-bound
// (profiles/r04/gemm_epilogue_ablation.txt).  x^2 clamped at 36 (P(36) = -4.98: 2^(x P) over- / underflows to the correct limits).
__device__ __forceinline__ f32x2 glc_gelu2_f32(f32x2 x) {
    f32x2 t = x * x;
    t[0] = fminf(t[0], 36.0f); t[1] = fminf(t[1], 36.0f);
    f32x2 p = t * (-5.21120420238525393e-09f) + 3.85102717598681759e-07f;
    p = p * t + (-1.14533653638075342e-05f);
    p = p * t + 1.59396300033220019e-04f;
    p = p * t + 9.55694841200346347e-05f;
    p = p * t + (-1.04838548282411514e-01f);
    p = p * t + (-2.30220721681703733e+00f);
    const f32x2 u = x * p;
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u[0])); r[1] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u[1]));
    return x * r;
}

// Group-split ("GS") rows of the fp32 mode (glc_kernels.h): element e of a row lives at halves (e >> 5) * 64 + (e & 31) (hi) and + 32 (lo).
typedef __attribute__((ext_vector_type(8))) _Float16 gs_h8;
__device__ __forceinline__ void gs_store8(f16_t* row, int e0, const float (&v)[8]) {      // e0 % 8 == 0
    gs_h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const f16_t h = (f16_t)v[e]; hi[e] = h; lo[e] = (f16_t)(v[e] - (float)h); }
    f16_t* p = row + (e0 >> 5) * 64 + (e0 & 31);
    *reinterpret_cast<gs_h8*>(p) = hi;
    *reinterpret_cast<gs_h8*>(p + 32) = lo;
}
__device__ __forceinline__ void gs_load8(const f16_t* row, int e0, float (&v)[8]) {
    const f16_t* p = row + (e0 >> 5) * 64 + (e0 & 31);
    const gs_h8 hi = *reinterpret_cast<const gs_h8*>(p), lo = *reinterpret_cast<const gs_h8*>(p + 32);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)hi[e] + (float)lo[e];
}

// "GX" rows (round 3; the operand image of the MX cross-term GEMM, gemm256x.hip): per 32 elements the same 128 bytes as a GS group, as
//   [32 x f16 hi | 4 x (8 x fp8 lo8, 8 x fp8 hi8)],  hi = f16(x),  lo8 = e4m3((x - hi) * 2^(GLC_GX_SHIFT + sc)),  hi8 = e4m3(x * 2^sc)   (weights: saturating; activations: see gx_split8)
// i.e. the fp8 parts of elements 8 j .. 8 j + 7 sit together in the 16 bytes at 64 + 16 j — activations ("A order") as [lo8 | hi8],
// weights ("W order") as [hi8 | lo8] — so a producer that holds 8 consecutive values writes two 16-byte pieces, like a GS row.
// A split product a*w = a_hi*w_hi + (a_hi*w_lo + a_lo*w_hi) keeps its f16 MFMA for the first term and runs BOTH cross terms as ONE
// block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4): lane (row, h) hands over the 32 fp8 bytes of elements 16 h .. 16 h + 15, so
// the 64 k-slots of the instruction pair a_lo8 with w_hi8 and a_hi8 with w_lo8 slot by slot, and every slot's product carries the same
// factor 2^-(SHIFT + sc_a + sc_w): ONE e8m0 scale per operand (2^-(SHIFT + sc_a) on A, 2^-sc_w on W) for every block.  The cross terms
// are ~2^-11 of the product, so their 4-bit operands leave a relative error of ~2^-15 — sixteen times below single f16 operands — at
// 2 instead of 3 f16-MFMA times per product.  x = hi + lo8 * 2^-(SHIFT + sc) also reads a row back to ~15 bits (residual adds, row
// gathers).  SHIFT = 11: |lo| <= 2^-11 |hi|, so the scaled residual never exceeds |hi| and shares hi8's range; what leaves that range
// (|x| 2^sc > 448) is counted by the fp8 range guard below and that forward is repeated on the split-f16 kernels.
#ifndef GLC_GX_SHIFT
#define GLC_GX_SHIFT 11
#endif
__device__ __forceinline__ uint32_t glc_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
// fp8 range guard.  An activation beyond the e4m3 range (|x| 2^sc > 448) has no e4m3 image (the unclamped conversion of gx_split8 gives
// 448 up to 464 and NaN beyond; a clamped one would silently leave that element at single-f16 accuracy).  Synthetic weights never get there; the outlier channels of trained checkpoints (10^2 .. 10^4 in the raw residual
// stream of a pre-norm decoder) do.  Every producer of an activation operand image (GX rows, MX tiles) therefore counts such elements
// in a per-engine device counter (sat != nullptr); the engine reads it with the logits and repeats a forward that counted any on the
// split-f16 kernels, whose operands hold up to 65504 (engine.hip forward_one).
__device__ __forceinline__ void gx_range_note(const float (&v)[8], float k_hi, unsigned* sat) {
    if (!sat) return;
    float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
    m = fmaxf(fmaxf(m, fabsf(v[3])), fabsf(v[4]));
    m = fmaxf(fmaxf(m, fabsf(v[5])), fabsf(v[6]));
    m = fmaxf(m, fabsf(v[7]));
    if (m * k_hi > 448.0f) atomicAdd(sat, 1u);
}
// 1 / k for k = 2^e (bit pattern (e + 127) << 23 -> (127 - e) << 23): the scales of the activation images are powers of two held in scalar registers
__device__ __forceinline__ float gx_pow2_inv(float k) { return __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(uint32_t, k)); }
// Activation images with exponent sc (engine.hip act_sc; 0 unless a forward left the e4m3 range): k_hi = 2^sc, k_lo = 2^(sc + SHIFT)
__device__ __forceinline__ float gx_act_khi(int sc) { return __builtin_bit_cast(float, (uint32_t)(127 + sc) << 23); }
__device__ __forceinline__ float gx_act_klo(int sc) { return __builtin_bit_cast(float, (uint32_t)(127 + sc + GLC_GX_SHIFT) << 23); }
// The activation split of 8 values: hi = f16(x) (RNE), lo8 = e4m3((x - hi) / inv_lo), hi8 = e4m3(x / inv_hi), inv_* powers of two (the
// conversion takes the divisor as a scale operand: no multiply).  Runs once per element in every GEMM epilogue, where it IS the epilogue's
// time (profiles/r04/gemm_epilogue_ablation.txt), so it does NOT clamp: a value beyond e4m3's range (|x / inv_hi| > 448, NaN from 464 on)
// has been counted by the fp8 range guard at the same call site (gx_range_note) and the result of that forward is never used (engine.hip
// forward_one / glc_engine_sync).  Weights and tables (converted once, no guard) keep the saturating glc_fp8x4.
__device__ __forceinline__ void gx_split8(const float (&v)[8], float inv_hi, float inv_lo, gs_h8& hi, u32x2& l8, u32x2& h8) {
    typedef short v2i16 __attribute__((ext_vector_type(2)));
    float l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const f16_t h = (f16_t)v[e]; hi[e] = h; l[e] = v[e] - (float)h; }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        v2i16 wl = {0, 0}, wh = {0, 0};
        wl = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl, l[4 * q], l[4 * q + 1], inv_lo, false);
        wl = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl, l[4 * q + 2], l[4 * q + 3], inv_lo, true);
        wh = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wh, v[4 * q], v[4 * q + 1], inv_hi, false);
        wh = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wh, v[4 * q + 2], v[4 * q + 3], inv_hi, true);
        l8[q] = __builtin_bit_cast(uint32_t, wl);
        h8[q] = __builtin_bit_cast(uint32_t, wh);
    }
}
// eight consecutive elements e0 .. e0 + 7 (e0 % 8 == 0) of a GX row; `row` = the row's first byte; k_hi = 2^sc, k_lo = 2^(sc + SHIFT)
// NT: non-temporal stores — for an output that streams to HBM and is too large for the caches to keep until its reader runs (FFN1's 805 MB
// intermediate at c3): it then does not evict the operand panels the same launch is still re-reading (FFN1 + GELU 750 -> 717 us).
template <bool WORDER = false, bool NT = false>
__device__ __forceinline__ void gx_store8(unsigned char* row, int e0, const float (&v)[8], float k_hi, float k_lo, unsigned* sat = nullptr) {
    gx_range_note(v, k_hi, sat);
    gs_h8 hi;
    unsigned char* p = row + (e0 >> 5) * 128;
    u32x4 x8;
    if constexpr (WORDER) {          // weights, once at load: saturating
        float l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const f16_t h = (f16_t)v[e]; hi[e] = h; l[e] = (v[e] - (float)h) * k_lo; }
        const uint32_t l0 = glc_fp8x4(l[0], l[1], l[2], l[3]), l1 = glc_fp8x4(l[4], l[5], l[6], l[7]);
        const uint32_t h0 = glc_fp8x4(v[0] * k_hi, v[1] * k_hi, v[2] * k_hi, v[3] * k_hi), h1 = glc_fp8x4(v[4] * k_hi, v[5] * k_hi, v[6] * k_hi, v[7] * k_hi);
        x8 = (u32x4){h0, h1, l0, l1};
    } else {                         // activations, every forward: gx_split8 (k_hi, k_lo powers of two: their reciprocals are exact)
        u32x2 l8, h8;
        gx_split8(v, gx_pow2_inv(k_hi), gx_pow2_inv(k_lo), hi, l8, h8);
        x8 = (u32x4){l8[0], l8[1], h8[0], h8[1]};
    }
    if constexpr (NT) {
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, hi), reinterpret_cast<u32x4*>(p + (e0 & 31) * 2));
        __builtin_nontemporal_store(x8, reinterpret_cast<u32x4*>(p + 64 + (e0 & 31) * 2));
    } else {
        *reinterpret_cast<gs_h8*>(p + (e0 & 31) * 2) = hi;
        *reinterpret_cast<u32x4*>(p + 64 + (e0 & 31) * 2) = x8;
    }
}
// 8 consecutive values of one row of an MX tile (decoder_mx.hip; the decoder QKV epilogue of gemm256x.hip): the 16-byte f16 piece and the two
// 8-byte fp8 pieces (hl: (hi8 | lo8), else (lo8 | hi8)); exponent 0
__device__ __forceinline__ void store_mx8(unsigned char* f16_dst, unsigned char* mx_dst, const float (&v)[8], bool hl, unsigned* sat) {
    gx_range_note(v, 1.0f, sat ? sat + 1 : nullptr);      // (MX tiles count in the guard's SECOND word: their exponent is fixed, lowering the rows' exponent cannot help them)
    gs_h8 hi;
    u32x2 l8, h8;
    gx_split8(v, 1.0f, 1.0f / (float)(1 << GLC_GX_SHIFT), hi, l8, h8);
    *reinterpret_cast<gs_h8*>(f16_dst) = hi;
    *reinterpret_cast<u32x2*>(mx_dst) = hl ? h8 : l8;
    *reinterpret_cast<u32x2*>(mx_dst + 16) = hl ? l8 : h8;
}
// x = hi + lo8 * inv_lo, inv_lo = 2^-(sc + SHIFT)
__device__ __forceinline__ void gx_decode8(const gs_h8& hi, const u32x2& lo8, float inv_lo, float (&v)[8]) {
    v[0] = (float)hi[0] + __builtin_amdgcn_cvt_f32_fp8((int)lo8[0], 0) * inv_lo; v[1] = (float)hi[1] + __builtin_amdgcn_cvt_f32_fp8((int)lo8[0], 1) * inv_lo;
    v[2] = (float)hi[2] + __builtin_amdgcn_cvt_f32_fp8((int)lo8[0], 2) * inv_lo; v[3] = (float)hi[3] + __builtin_amdgcn_cvt_f32_fp8((int)lo8[0], 3) * inv_lo;
    v[4] = (float)hi[4] + __builtin_amdgcn_cvt_f32_fp8((int)lo8[1], 0) * inv_lo; v[5] = (float)hi[5] + __builtin_amdgcn_cvt_f32_fp8((int)lo8[1], 1) * inv_lo;
    v[6] = (float)hi[6] + __builtin_amdgcn_cvt_f32_fp8((int)lo8[1], 2) * inv_lo; v[7] = (float)hi[7] + __builtin_amdgcn_cvt_f32_fp8((int)lo8[1], 3) * inv_lo;
}
// (activation rows: A order)
__device__ __forceinline__ void gx_load8(const unsigned char* row, int e0, float (&v)[8], float inv_lo) {
    const unsigned char* p = row + (e0 >> 5) * 128;
    gx_decode8(*reinterpret_cast<const gs_h8*>(p + (e0 & 31) * 2), *reinterpret_cast<const u32x2*>(p + 64 + (e0 & 31) * 2), inv_lo, v);
}

// ---- "GY" operand images (round 4; the MX cross-term GEMM with e2m3 parts and per-block scales — "MX proper") ----
// Same split product as GX rows (a w = a_hi w_hi in f16 + both cross terms in ONE block-scaled MFMA), but the 8-bit e4m3 parts with a fixed
// exponent become 6-bit e2m3 parts (the same 4 significant bits) with an e8m0 scale per block of 16 elements: the scaled MFMA runs at
// twice the fp8 rate (32 instead of 64 cycles per 32x32x64), a row group shrinks from 128 to 112 bytes on both data paths of the GEMM loop, the
// conversion is ONE v_cvt_scalef32_2xpk16_fp6_f32 per 16 elements (16 v_cvt_scalef32_pk_fp8_f32 before), and activations need no range
// guard.  Semantics pinned by scripts/probes/mx6_probe.hip (profiles/r04/mx6_probe.txt); timing-only case profiles/r04/gemm_fp6_shaped.txt.
// A matrix of `rows` rows x K elements (K % 64 == 0) is stored GROUP-MAJOR — [K / 32 groups][rows][112 bytes], then the scales
// [K / 64 group pairs][rows][4 bytes] — so that what the GEMM stages per K step (one group of 256 consecutive rows) is 28 KiB of consecutive
// whole cache lines (a 112-byte group inside a row-major row would straddle two lines, and every line would be fetched twice).
// gy_bytes(rows, K) = rows (3.5 K + K / 16): inside the 4 K bytes per row the GX / GS images take.  Group (g, row):
//   [ 64 B: 32 x f16 hi | 16 B: block 0 bytes 0..15 | 16 B: block 1 bytes 0..15 | 8 B: block 0 bytes 16..23 | 8 B: block 1 bytes 16..23 ]
// block h = elements 16 h .. 16 h + 15 of the group as 32 packed e2m3 values, value 2 i = P0(element i), value 2 i + 1 = P1(element i) — the
// order the conversion instruction interleaves its two sources in — with (P0, P1) = (lo 2^11, hi) for activations ("A order") and
// (hi, lo 2^11) for weights ("W order"), lo = x - f16(x): an MFMA lane (row, h) reads its block as one 16-byte and one 8-byte piece at the
// same offsets for both h, and the 64 k-slots pair a_lo with w_hi and a_hi with w_lo slot by slot.  Scale byte of (g, h, row): byte
// 2 (g & 1) + h of the row's dword in pair g >> 1.  With E = the exponent of the block's largest |x| (largest part lands in [4, 8); e2m3
// tops out at 7.5 and saturates), W order stores 127 + E - 2, A order 127 + E - 2 - 11 — the byte the MFMA takes as it is (the 2^-11 of
// the cross terms rides on the A side), and the factor that turns an A image's lo parts back into lo (x = hi + e2m3 2^(b - 127)).
__host__ __device__ inline size_t gy_bytes(size_t rows, int K) { return rows * ((size_t)(K >> 5) * 112 + (size_t)(K >> 4)); }
__host__ __device__ inline size_t gy_group_off(size_t rows, size_t row, int g) { return ((size_t)g * rows + row) * 112; }
__host__ __device__ inline size_t gy_scale_off(size_t rows, int K, size_t row, int g) { return (size_t)(K >> 5) * rows * 112 + ((size_t)(g >> 1) * rows + row) * 4 + 2 * (g & 1); }
typedef __attribute__((ext_vector_type(16))) float gy_f16v;
typedef __attribute__((ext_vector_type(32))) float gy_f32v;
typedef __attribute__((ext_vector_type(6))) int gy_i6;
// The two fp6 conversions through inline assembly with an EARLY-CLOBBER destination: left to the register allocator, the 6-register result of
// v_cvt_scalef32_2xpk16_fp6_f32 lands on top of one of its 16-register sources (it did in every instance of this library), and the
// instruction then converts values it has already overwritten — measured: W-order rows whose hi parts came out saturated.
__device__ __forceinline__ gy_i6 gy_cvt_2x16(const gy_f16v& s0, const gy_f16v& s1, float scale) {
    gy_i6 r;
    asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(s0), "v"(s1), "v"(scale));
    return r;
}
__device__ __forceinline__ gy_f32v gy_cvt_pk32_f32(const gy_i6& r, float scale) {
    gy_f32v d;
    asm("v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2" : "=&v"(d) : "v"(r), "v"(scale));
    return d;
}
// 16 consecutive elements e0 .. e0 + 15 (e0 % 16 == 0) of row `row` = one block.  Returns the block's scale byte; SCALE = false leaves storing
// it to the caller (the GEMM epilogues pack the four bytes of a row's 64 columns into one dword store).
template <bool WORDER = false, bool NT = false, bool SCALE = true>
__device__ __forceinline__ unsigned gy_store16(unsigned char* base, size_t rows, int K, size_t row, int e0, const float (&v)[16]) {
    float m = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
    for (int e = 2; e < 16; e += 2) m = fmaxf(fmaxf(m, fabsf(v[e])), fabsf(v[e + 1]));
    uint32_t eb = __builtin_bit_cast(uint32_t, m) >> 23;          // biased exponent of the largest magnitude (m >= 0)
    eb = eb < 16u ? 16u : (eb > 254u ? 254u : eb);                // (an all-zero / denormal block: any small scale does)
    const float scale = __builtin_bit_cast(float, (eb - 2u) << 23);
    gs_h8 h0, h1;
    gy_f16v lo, hi;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const f16_t hv = (f16_t)v[e];
        if (e < 8) h0[e] = hv; else h1[e - 8] = hv;
        hi[e] = (float)hv;
        lo[e] = (v[e] - (float)hv) * (float)(1 << GLC_GX_SHIFT);
    }
    const gy_i6 r = WORDER ? gy_cvt_2x16(hi, lo, scale) : gy_cvt_2x16(lo, hi, scale);
    const int g = e0 >> 5, h = (e0 >> 4) & 1;
    unsigned char* p = base + gy_group_off(rows, row, g);
    const u32x4 r4 = {(uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]};
    const u32x2 r2 = {(uint32_t)r[4], (uint32_t)r[5]};
    if constexpr (NT) {
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, h0), reinterpret_cast<u32x4*>(p + 32 * h));
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, h1), reinterpret_cast<u32x4*>(p + 32 * h + 16));
        __builtin_nontemporal_store(r4, reinterpret_cast<u32x4*>(p + 64 + 16 * h));
        __builtin_nontemporal_store(r2, reinterpret_cast<u32x2*>(p + 96 + 8 * h));
    } else {
        *reinterpret_cast<gs_h8*>(p + 32 * h) = h0;
        *reinterpret_cast<gs_h8*>(p + 32 * h + 16) = h1;
        *reinterpret_cast<u32x4*>(p + 64 + 16 * h) = r4;
        *reinterpret_cast<u32x2*>(p + 96 + 8 * h) = r2;
    }
    const unsigned b = WORDER ? eb - 2u : eb - 2u - GLC_GX_SHIFT;
    if constexpr (SCALE) base[gy_scale_off(rows, K, row, g) + h] = (unsigned char)b;
    return b;
}
// 8 consecutive elements e0 .. e0 + 7 (e0 % 8 == 0) per lane, the lane `partner` (lane ^ partner_xor) holding the other half of the block
// (both lanes must be active): the producers that have 8 values per lane (row kernels, attention context epilogues); A order
__device__ __forceinline__ void gy_store8_pair(unsigned char* base, size_t rows, int K, size_t row, int e0, const float (&v)[8], int partner_xor) {
    float m = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
    for (int e = 2; e < 8; e += 2) m = fmaxf(fmaxf(m, fabsf(v[e])), fabsf(v[e + 1]));
    m = fmaxf(m, __shfl_xor(m, partner_xor, 64));
    uint32_t eb = __builtin_bit_cast(uint32_t, m) >> 23;
    eb = eb < 16u ? 16u : (eb > 254u ? 254u : eb);
    const float scale = __builtin_bit_cast(float, (eb - 2u) << 23);
    gs_h8 hh;
    gy_f16v s0, s1;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const f16_t hv = (f16_t)v[e]; hh[e] = hv; s1[e] = (float)hv; s0[e] = (v[e] - (float)hv) * (float)(1 << GLC_GX_SHIFT); }
#pragma unroll
    for (int e = 8; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    const gy_i6 r = gy_cvt_2x16(s0, s1, scale);       // values 0 .. 15 (bytes 0 .. 11) are this lane's
    const int g = e0 >> 5, h = (e0 >> 4) & 1, j = (e0 >> 3) & 1;
    unsigned char* p = base + gy_group_off(rows, row, g);
    *reinterpret_cast<gs_h8*>(p + (e0 & 31) * 2) = hh;
    if (j == 0) {                                            // block bytes 0 .. 11
        typedef __attribute__((ext_vector_type(3))) uint32_t u32x3;
        *reinterpret_cast<u32x3*>(p + 64 + 16 * h) = (u32x3){(uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2]};
        base[gy_scale_off(rows, K, row, g) + h] = (unsigned char)(eb - 2u - GLC_GX_SHIFT);
    } else {                                                 // block bytes 12 .. 15 and 16 .. 23
        *reinterpret_cast<uint32_t*>(p + 64 + 16 * h + 12) = (uint32_t)r[0];
        *reinterpret_cast<u32x2*>(p + 96 + 8 * h) = (u32x2){(uint32_t)r[1], (uint32_t)r[2]};
    }
}
// one block of an A-order image back to fp32: x = hi + lo, lo = e2m3 2^(b - 127)
__device__ __forceinline__ void gy_decode16(const gs_h8& h0, const gs_h8& h1, const u32x4& r4, const u32x2& r2, unsigned b, float (&v)[16]) {
    const gy_i6 r = {(int)r4[0], (int)r4[1], (int)r4[2], (int)r4[3], (int)r2[0], (int)r2[1]};
    const gy_f32v d = gy_cvt_pk32_f32(r, __builtin_bit_cast(float, b << 23));
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = (float)h0[e] + d[2 * e]; v[8 + e] = (float)h1[e] + d[16 + 2 * e]; }
}
__device__ __forceinline__ void gy_load16(const unsigned char* base, size_t rows, int K, size_t row, int e0, float (&v)[16]) {
    const int g = e0 >> 5, h = (e0 >> 4) & 1;
    const unsigned char* p = base + gy_group_off(rows, row, g);
    gy_decode16(*reinterpret_cast<const gs_h8*>(p + 32 * h), *reinterpret_cast<const gs_h8*>(p + 32 * h + 16), *reinterpret_cast<const u32x4*>(p + 64 + 16 * h),
                *reinterpret_cast<const u32x2*>(p + 96 + 8 * h), base[gy_scale_off(rows, K, row, g) + h], v);
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
