// Probe (developer, standalone; round 4): the cross terms of a split product as ONE block-scaled MFMA on e2m3 (fp6) parts with per-block
// e8m0 scales — the format docs/LOG_r01-r05.md §9 names as the next step — next to today's e4m3 parts with a fixed exponent.  Pins, on gfx950:
//   * v_cvt_scalef32_2xpk16_fp6_f32: order of the 32 packed values (MEASURED: interleaved — value 2 i = src0[i], value 2 i + 1 = src1[i]) and
//     result = e2m3(src / scale);
//   * v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 2: lane (row, h) supplies 32 fp6 values in six registers = k slots 32 h .. 32 h + 31,
//     scaled by the lane's own e8m0 byte (op_sel 0 = byte 0 of the scale register);
//   * the accuracy of a_lo w_hi + a_hi w_lo from such parts (block = the 16 elements a lane holds, scale 2^(E - 2), E = exponent of the block's
//     largest |x|) against the exact cross terms, for ordinary operands and for rows with an outlier of 3000.
// NOTE: the conversion builtin lets the compiler place the 6-register result on top of a source tuple, and the instruction then reads values it has
// already overwritten (found in gemm256x.hip's GY path; glc_common.h gy_cvt_2x16 uses inline assembly with an early-clobber result) — this probe's
// allocation happened to be harmless.
// One wave: A 32 rows x 32 elements, W 32 columns x 32 elements (one MFMA's worth).   hipcc --offload-arch=gfx950 -O2 mx6_probe.hip -o mx6_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef short v2s __attribute__((ext_vector_type(2)));

constexpr int SHIFT = 11;

// in: A[32][32], W[32][32] fp32.  out6 / out8: D[32][32] of the cross terms from fp6 block-scaled parts / e4m3 parts with exponent 0 (W: exponent ws)
__global__ void probe(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out6, float* __restrict__ out8, int* __restrict__ dump, int ws) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    float a[16], w[16];
    for (int e = 0; e < 16; ++e) { a[e] = A[r * 32 + 16 * h + e]; w[e] = W[r * 32 + 16 * h + e]; }
    float ahi[16], alo[16], whi[16], wlo[16];
    float amax = 0.f, wmax = 0.f;
    for (int e = 0; e < 16; ++e) {
        const _Float16 ah = (_Float16)a[e], wh = (_Float16)w[e];
        ahi[e] = (float)ah; alo[e] = (a[e] - (float)ah) * (float)(1 << SHIFT);
        whi[e] = (float)wh; wlo[e] = (w[e] - (float)wh) * (float)(1 << SHIFT);
        amax = fmaxf(amax, fabsf(a[e])); wmax = fmaxf(wmax, fabsf(w[e]));
    }
    // ---- fp6: block scale 2^(E - 2) (largest element lands in [4, 8): 7.5 is e2m3's largest value, beyond it the conversion saturates)
    int ea, ew;
    (void)frexpf(amax, &ea); (void)frexpf(wmax, &ew);      // amax = m 2^ea, m in [0.5, 1)  ->  E = ea - 1
    const int sa_e = amax > 0.f ? ea - 1 - 2 : 0, sw_e = wmax > 0.f ? ew - 1 - 2 : 0;
    const float sa = ldexpf(1.f, sa_e), sw = ldexpf(1.f, sw_e);
    v16f s0, s1;
    for (int e = 0; e < 16; ++e) { s0[e] = alo[e]; s1[e] = ahi[e]; }
    const v6i a6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(s0, s1, sa);      // slots 2 i: a_lo[i], 2 i + 1: a_hi[i]
    for (int e = 0; e < 16; ++e) { s0[e] = whi[e]; s1[e] = wlo[e]; }
    const v6i w6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(s0, s1, sw);      // slots 2 i: w_hi[i], 2 i + 1: w_lo[i] — pairs a_lo w_hi and a_hi w_lo slot by slot
    if (lane == 0) for (int i = 0; i < 6; ++i) dump[i] = a6[i];
    v8i a8r = {a6[0], a6[1], a6[2], a6[3], a6[4], a6[5], 0, 0}, w8r = {w6[0], w6[1], w6[2], w6[3], w6[4], w6[5], 0, 0};
    const int sca = 127 + sa_e - SHIFT, scw = 127 + sw_e;       // per-lane e8m0 bytes (byte 0)
    typedef float v16acc __attribute__((ext_vector_type(16)));
    v16acc c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8r, a8r, c, 2, 2, 0, scw, 0, sca);       // D[n][m] as gemm256x.hip
    for (int i = 0; i < 16; ++i) out6[(8 * (i >> 2) + 4 * h + (i & 3)) * 32 + r] = c[i];          // row n = 8 (i >> 2) + 4 h + (i & 3), column m = r
    // ---- e4m3, fixed exponents (today's format): lane holds [lo8 x16 | hi8 x16] / [hi8 x16 | lo8 x16]
    v8i a8, w8;
    const float kw = ldexpf(1.f, ws);
    for (int q = 0; q < 4; ++q) {
        v2s t = {0, 0};
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, alo[4 * q], alo[4 * q + 1], 1.f, false);
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, alo[4 * q + 2], alo[4 * q + 3], 1.f, true);
        a8[q] = __builtin_bit_cast(int, t);
        t = (v2s){0, 0};
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, ahi[4 * q], ahi[4 * q + 1], 1.f, false);
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, ahi[4 * q + 2], ahi[4 * q + 3], 1.f, true);
        a8[4 + q] = __builtin_bit_cast(int, t);
        t = (v2s){0, 0};
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, whi[4 * q] * kw, whi[4 * q + 1] * kw, 1.f, false);
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, whi[4 * q + 2] * kw, whi[4 * q + 3] * kw, 1.f, true);
        w8[q] = __builtin_bit_cast(int, t);
        t = (v2s){0, 0};
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, wlo[4 * q] * kw, wlo[4 * q + 1] * kw, 1.f, false);
        t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, wlo[4 * q + 2] * kw, wlo[4 * q + 3] * kw, 1.f, true);
        w8[4 + q] = __builtin_bit_cast(int, t);
    }
    v16acc d = {};
    d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, a8, d, 0, 0, 0, 127 - ws, 0, 127 - SHIFT);
    for (int i = 0; i < 16; ++i) out8[(8 * (i >> 2) + 4 * h + (i & 3)) * 32 + r] = d[i];
}

static float e2m3_decode(int v) {       // 6 bits: sign, 2 exponent (bias 1), 3 mantissa
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    const float x = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
    return s ? -x : x;
}

int main() {
    const int N = 32 * 32;
    std::vector<float> A(N), W(N), o6(N), o8(N);
    float *dA, *dW, *d6, *d8; int* dd;
    hipMalloc(&dA, N * 4); hipMalloc(&dW, N * 4); hipMalloc(&d6, N * 4); hipMalloc(&d8, N * 4); hipMalloc(&dd, 64);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.f - 1.f; };
    auto gauss = [&]() { float t = 0; for (int i = 0; i < 6; ++i) t += rnd(); return t * 0.7071f; };
    for (int scen = 0; scen < 3; ++scen) {
        for (int i = 0; i < N; ++i) { A[i] = gauss() * (scen == 2 ? 20.f : 1.f); W[i] = gauss() * 0.03f; }
        if (scen >= 1) for (int r = 0; r < 32; r += 2) A[r * 32 + 5] = scen == 1 ? 3000.f : 400.f;     // an outlier channel in every other row
        float wmax = 0; for (int i = 0; i < N; ++i) wmax = fmaxf(wmax, fabsf(W[i]));
        int ws = 0; while (ldexpf(wmax, ws + 1) <= 240.f) ++ws;          // glc_gx_weight_exponent
        hipMemcpy(dA, A.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), N * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dW, d6, d8, dd, ws);
        hipMemcpy(o6.data(), d6, N * 4, hipMemcpyDeviceToHost); hipMemcpy(o8.data(), d8, N * 4, hipMemcpyDeviceToHost);
        int dump[6]; hipMemcpy(dump, dd, 24, hipMemcpyDeviceToHost);
        if (scen == 0) {      // lane 0's packed values against a host encoding of the same inputs: the order of the 32 values
            unsigned long long bits[3] = {0, 0, 0};
            for (int i = 0; i < 6; ++i) bits[i / 2] |= (unsigned long long)(unsigned)dump[i] << (32 * (i & 1));
            float amax = 0; for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(A[e]));
            int ea; (void)frexpf(amax, &ea); const float sa = ldexpf(1.f, ea - 3);
            float got[32];
            for (int k = 0; k < 32; ++k) {
                const int bit = 6 * k, w0 = bit / 64, off = bit % 64;
                unsigned long long v = bits[w0] >> off;
                if (off > 58) v |= bits[w0 + 1] << (64 - off);
                got[k] = e2m3_decode((int)(v & 63));
            }
            float want0[16], want1[16];
            for (int e = 0; e < 16; ++e) { const float hi = (float)(_Float16)A[e]; want0[e] = (A[e] - hi) * 2048.f / sa; want1[e] = hi / sa; }
            auto near = [](float g, float w) { const float aw = fabsf(w) > 7.5f ? 7.5f : fabsf(w); const float step = aw < 1.f ? 0.125f : aw < 2.f ? 0.125f : aw < 4.f ? 0.25f : 0.5f;
                                               return fabsf(fabsf(g) - aw) <= 0.5f * step + 1e-6f && (g == 0.f || w == 0.f || (g < 0) == (w < 0)); };
            int seq = 0, inter = 0;
            for (int e = 0; e < 16; ++e) {
                seq += near(got[e], want0[e]) + near(got[16 + e], want1[e]);
                inter += near(got[2 * e], want0[e]) + near(got[2 * e + 1], want1[e]);
            }
            printf("packing of v_cvt_scalef32_2xpk16_fp6_f32 (lane 0, 32 values): [src0 x16 | src1 x16] matches %d of 32, interleaved (src0[i], src1[i]) matches %d of 32\n", seq, inter);
            printf("  first values decoded: %.3f %.3f %.3f %.3f   src0/scale: %.3f %.3f   src1/scale: %.3f %.3f\n", got[0], got[1], got[2], got[3], want0[0], want0[1], want1[0], want1[1]);
        }
        double e6 = 0, e8 = 0, ref2 = 0, m6 = 0, m8 = 0, mag = 0;
        for (int n = 0; n < 32; ++n) for (int m = 0; m < 32; ++m) {
            double ex = 0, ab = 0;
            for (int k = 0; k < 32; ++k) {
                const double a = A[m * 32 + k], w = W[n * 32 + k];
                const double ah = (double)(float)(_Float16)(float)a, wh = (double)(float)(_Float16)(float)w;
                ex += (a - ah) * wh + ah * (w - wh);
                ab += fabs(a * w);
            }
            const double d6v = o6[n * 32 + m] - ex, d8v = o8[n * 32 + m] - ex;
            e6 += d6v * d6v; e8 += d8v * d8v; ref2 += ab * ab; mag = fmax(mag, ab);
            m6 = fmax(m6, fabs(d6v) / ab); m8 = fmax(m8, fabs(d8v) / ab);
        }
        printf("scenario %d (%s): cross-term error relative to sum |a||w|:  fp6 block-scaled rms %.2e max %.2e   e4m3 fixed exponent rms %.2e max %.2e\n", scen,
               scen == 0 ? "a ~ N(0,1), w ~ N(0,0.03)" : scen == 1 ? "+ outlier 3000 in every other row (beyond e4m3's 448)" : "a ~ N(0,20) + outlier 400",
               sqrt(e6 / ref2), m6, sqrt(e8 / ref2), m8);
    }
    return 0;
}
