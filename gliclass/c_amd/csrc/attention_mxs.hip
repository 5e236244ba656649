// DeBERTa-v2/v3 disentangled self-attention on "MX tiles", role-split workgroup (round 5) — the attention of the MX pipeline.
//
// Same algebra, operands (glc_layout.h "MX tiles"), arithmetic (one f16 MFMA + one block-scaled fp8 MFMA of cross terms per product, fp32
// accumulators, deferred-rescale online softmax) and outputs (GX context rows) as attention_mx.hip — the results are bit-identical to that
// kernel's — but the work of a (32-query tile, 32-key tile) pair is no longer one wave's serial chain.  Rounds 2-4 established that the
// band kernels are a chain of latencies whose matrix-pipe time is fully exposed (every component of the tile additive, two symmetric
// waves per SIMD overlapping a quarter of it: DESIGN.md §3f / §9).  Here a SIMD's two waves have different JOBS:
//
//   matrix wave  M_w (waves 0-3, one per SIMD)   owns query tile w of the workgroup's four: every MFMA of the tile — S^T = K Q^T on top of the
//                                                gathered c2p band, the p2c block K PQ^T for the shared image, the c2p block PK Q^T of the next
//                                                key tile, O^T += V^T P — plus the LDS-DMA of the K / V^T ring and the position-row requests.
//                                                It never evaluates an exponential and never waits for one inside a half step.
//   softmax wave X_w (waves 4-7, M_w's partner   reads S^T of tile w from LDS, adds the p2c band (image gather) and the key bias, keeps the running
//                on the same SIMD)               maximum / sum, exponentials, splits P into f16 + (hi8 | lo8) and hands it back through LDS.
//
// A key tile t is one STEP of two halves separated by workgroup barriers A(t), B(t):
//   first half   M: DMA K(t+2), V^T(t+1) | K(t) fragments | c2p(t+1) = PK Q^T -> own ring | S^T(t) = ring gather + K Q^T -> S buffer |
//                   p2c(t) = PQ K^T (+ the block nobody owns, one wave in four) -> shared image
//                X: second part of softmax(t-1): row sums, f16 / fp8 split -> P buffer (+ rescale factors, flag)
//   A(t)         S(t), image(t), P(t-1) are published
//   second half  M: O^T += V^T(t-1) P(t-1) (after the deferred rescale, if X flagged one) | position rows of step t+1 requested
//                X: S(t) + image gather + key bias, maximum, rescale decision, exponentials
//   B(t)         S buffer, image and P buffer are free again
// so P.V runs one step behind the scores and every buffer is single: between a write and the reads of it lies one barrier, between
// those reads and the next write the other.  Ring of three K and three V^T slots (DMA two / one tile ahead).  LDS: c2p rings 34 KB,
// p2c image 20.5 KB, K / V^T rings 48 KB, S and P buffers 32 KB, factors 1 KB = 135.6 KB, one workgroup (8 waves) per CU.
//
// Saturated key tiles (delta constant: attention_wg.hip) take the same pipeline with S^T = cq + K Q^T + K PQ[d*]^T and no image.
#include <stdio.h>
#include <stdlib.h>
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

constexpr float RESCALE_THR = 8.0f;   // log2 units (as attention_mx.hip)
constexpr int NQ = 4;                 // query tiles = matrix waves per workgroup
constexpr int LROW = 68;              // floats per c2p ring row (2 blocks of 32 + 4 pad)
constexpr int LROWP = 32 * (NQ + 1) + 4;            // floats per p2c image row
constexpr int TILEB = GLC_MXT_BYTES;
constexpr int OFF_RING = 0;
constexpr int OFF_IMG = OFF_RING + NQ * 32 * LROW * 4;
constexpr int OFF_K = OFF_IMG + 32 * LROWP * 4;
constexpr int OFF_V = OFF_K + 3 * TILEB;
constexpr int OFF_S = OFF_V + 3 * TILEB;
constexpr int OFF_P = OFF_S + NQ * 4096;
constexpr int OFF_F = OFF_P + NQ * 4096;            // per tile: 64 floats (rescale factor / final 1 / l per lane) ...
constexpr int OFF_FLAG = OFF_F + NQ * 256;          // ... and one flag word per tile
constexpr int MXS_LDS = OFF_FLAG + 64;
static_assert(MXS_LDS <= 160 * 1024, "LDS budget");

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
struct MxFrag { f16x8 f[4]; i32x8 x[2]; };      // a 32-row x 64-column operand tile in registers (32 VGPRs)

__device__ __forceinline__ void glds16_sv(const unsigned char* ubase, unsigned lane_off, void* l) {      // attention_wg.hip
    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)l;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la), "v"(lane_off), "s"(ubase) : "memory");
}
__device__ __forceinline__ i32x8 cat8(const i32x4& a, const i32x4& b) {
    i32x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
template <int N> struct IC { static constexpr int value = N; };

extern __shared__ __attribute__((aligned(16))) unsigned char smem_mxs[];

__global__ __launch_bounds__(512, 2) void attn_mxs_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = wave & 3;
    const bool is_m = wave < NQ;
    const int c = lane & 31, h = lane >> 5;
    const int Sp = a.Sp;

    const int nqb = (Sp + 32 * NQ - 1) / (32 * NQ);
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int bh = xcd + 8 * (jj / nqb);
    const int Q0 = (jj % nqb) * 32 * NQ;
    const int QX = Q0 + 32 * NQ;                        // the query tile whose LOW block is the workgroup's unowned (last high) block
    if (bh >= a.B * a.nh) return;
    const int b = bh / a.nh, hh = bh - b * a.nh;
    const int q0 = Q0 + 32 * w;
    const bool active = q0 < Sp;
    const int q0m = active ? q0 : Sp - 32;
    const int klen = a.klen[b];
    if (Q0 >= klen && Q0 > 0) {
        // every query of this block lies past the row's last attended token: never read by an attended row; store zeros and leave
        if (active && is_m) {
            unsigned char* row = reinterpret_cast<unsigned char*>(a.CTX) + ((size_t)b * Sp + q0 + c) * 4 * a.H + (size_t)(2 * hh) * 128 + h * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(row + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
        }
        return;
    }
    const int nt = Sp >> 5;
    int nkt = (klen + 31) >> 5;
    nkt = nkt < 1 ? 1 : (nkt > nt ? nt : nkt);
    // key-tile ranges, workgroup-uniform (attention_wg.hip): [0, kt_a) delta == P - 1, [kt_a, kt_b) the band, [kt_b, nkt) delta == 0
    int kt_a = Q0 - 31 - a.rsat_pos >= 0 ? (Q0 - 31 - a.rsat_pos) / 32 + 1 : 0;
    kt_a = kt_a > nkt ? nkt : kt_a;
    int kt_b = (Q0 + 32 * (NQ - 1) + 31 - a.rsat_neg + 31) / 32;
    kt_b = kt_b < kt_a ? kt_a : (kt_b > nkt ? nkt : kt_b);

    float* s_buf = reinterpret_cast<float*>(smem_mxs + OFF_S + w * 4096);
    unsigned char* p_buf = smem_mxs + OFF_P + w * 4096;
    float* f_buf = reinterpret_cast<float*>(smem_mxs + OFF_F + w * 256);
    int* flag = reinterpret_cast<int*>(smem_mxs + OFF_FLAG + w * 4);
    float* p2c_img = reinterpret_cast<float*>(smem_mxs + OFF_IMG);
    const int rr_base = c - 8 * h + 31;

    if (!is_m) {
        // =============================== softmax wave ===============================
        const float* __restrict__ kb = a.kbias + (size_t)b * Sp;
        const int kfirst = a.kfirst[b];
        const int foff = 8 * h;
        float m = -3.0e38f, l = 0.f;
        float one_f = 1.0f;
        asm volatile("" : "+s"(one_f));      // opaque to the optimiser: fma(p, 1, -half) stays a v_fma_mix_f32
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // INIT
        for (int kt = 0; kt < nkt; ++kt) {
            const bool band = kt >= kt_a && kt < kt_b;
            const int k0 = kt * 32;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // A(kt): S(kt) and image(kt) are complete
            float sv[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(s_buf + g * 256 + lane * 4);
                sv[4 * g] = v[0]; sv[4 * g + 1] = v[1]; sv[4 * g + 2] = v[2]; sv[4 * g + 3] = v[3];
            }
            if (band) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const int kc = 16 * (i >> 3) + (i & 7);
                    const int prow = 16 * (i >> 3) + 8 * ((i >> 2) & 1) + (i & 3);
                    const f32x2 g = (f32x2){sv[i], sv[i + 1]} +
                                    (f32x2){p2c_img[(prow + 4 * h) * LROWP + 32 * w + rr_base - kc], p2c_img[(prow + 1 + 4 * h) * LROWP + 32 * w + rr_base - kc - 1]};
                    sv[i] = g[0]; sv[i + 1] = g[1];
                }
            }
            if (k0 + 32 > kfirst) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(kb + k0 + foff);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(kb + k0 + foff + 4);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff);
                const f32x4 b3 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { sv[i] += b0[i]; sv[4 + i] += b1[i]; sv[8 + i] += b2[i]; sv[12 + i] += b3[i]; }
            }
            float mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, sv[i]), sv[i + 1]);
            mx = fmaxf(mx, sv[15]);
            float alpha = 1.0f;
            int resc = 0;
            if (__builtin_amdgcn_ballot_w64(mx - m > RESCALE_THR) != 0ull) {     // deferred rescale (attention.hip)
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(m, mx);
                alpha = __builtin_amdgcn_exp2f(m - mnew);
                m = mnew;
                l *= alpha;
                resc = 1;
            }
            const f32x2 m2 = {m, m};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 d = (f32x2){sv[i], sv[i + 1]} - m2;
                sv[i] = __builtin_amdgcn_exp2f(d[0]); sv[i + 1] = __builtin_amdgcn_exp2f(d[1]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // B(kt): S buffer, image and P buffer are free
#pragma unroll
            for (int i = 0; i < 16; i += 2) ps2 += (f32x2){sv[i], sv[i + 1]};
            l += ps2[0] + ps2[1];
            // P travels as (hi8 | lo8): f16(p) for the f16 MFMAs (k-step t = keys 16 t + 8 h + j), fp8 parts of the 16 keys for the scaled one
            f16x8 pf[2];
            i32x8 px;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[t][j] = (f16_t)sv[8 * t + j];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * q], sv[4 * q + 1], 0, false);
                wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * q + 2], sv[4 * q + 3], wh, true);
                px[q] = wh;
                // lo8 = e4m3((p - f16(p)) 2^SHIFT): attention_mx.hip
                float r[4];
                const i32x4 pfw = __builtin_bit_cast(i32x4, pf[q >> 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pw = pfw[2 * (q & 1) + (e >> 1)];
                    if (e & 1) asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r[e]) : "v"(sv[4 * q + e]), "s"(one_f), "v"(pw));
                    else asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r[e]) : "v"(sv[4 * q + e]), "s"(one_f), "v"(pw));
                }
                typedef short v2i16 __attribute__((ext_vector_type(2)));
                v2i16 wl2 = {0, 0};
                wl2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl2, r[0], r[1], 1.0f / (float)(1 << GLC_GX_SHIFT), false);
                wl2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl2, r[2], r[3], 1.0f / (float)(1 << GLC_GX_SHIFT), true);
                px[4 + q] = __builtin_bit_cast(int, wl2);
            }
            *reinterpret_cast<f16x8*>(p_buf + lane * 16) = pf[0];
            *reinterpret_cast<f16x8*>(p_buf + 1024 + lane * 16) = pf[1];
            *reinterpret_cast<i32x4*>(p_buf + 2048 + lane * 16) = (i32x4){px[0], px[1], px[2], px[3]};
            *reinterpret_cast<i32x4*>(p_buf + 3072 + lane * 16) = (i32x4){px[4], px[5], px[6], px[7]};
            if (resc) f_buf[lane] = alpha;
            if (lane == 0) *flag = resc;
        }
        // final 1 / l: through the factor slot, once M has consumed the last tile's rescale factors (barrier B(nkt))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // A(nkt): P(nkt - 1) published
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // B(nkt): M has read the last factors
        l += __shfl_xor(l, 32, 64);
        f_buf[lane] = 1.0f / l;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // FIN
        return;
    }

    // =============================== matrix wave ===============================
    // e8m0 scales of the block-scaled MFMA (attention_mx.hip)
    const int SC = h ? (127 | ((127 - GLC_GX_SHIFT) << 8)) : ((127 - GLC_GX_SHIFT) | (127 << 8));
    auto mm_lh_hl = [&](const MxFrag& lh, const MxFrag& hl, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh.f[s], hl.f[s], acc, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(lh.x[m], hl.x[m], acc, 0, 0, 0, SC, 1, SC);
    };
    auto mm_hl_lh = [&](const MxFrag& hl, const MxFrag& lh, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl.f[s], lh.f[s], acc, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hl.x[m], lh.x[m], acc, 0, 0, 1, SC, 0, SC);
    };

    float* c2p_l = reinterpret_cast<float*>(smem_mxs + OFF_RING) + (size_t)w * 32 * LROW;             // this wave's ring [32 q][64 + 4]
    unsigned char* k_ring = smem_mxs + OFF_K;
    unsigned char* v_ring = smem_mxs + OFF_V;

    const unsigned char* __restrict__ Qg = reinterpret_cast<const unsigned char*>(a.Qh) + ((size_t)bh * nt + (q0m >> 5)) * TILEB;
    const unsigned char* __restrict__ Kg = reinterpret_cast<const unsigned char*>(a.Kh) + (size_t)bh * nt * TILEB;
    const unsigned char* __restrict__ Vg = reinterpret_cast<const unsigned char*>(a.Vt) + (size_t)bh * nt * TILEB;
    const unsigned char* __restrict__ PKg = reinterpret_cast<const unsigned char*>(a.PK) + (size_t)hh * (a.P >> 5) * TILEB;
    const unsigned char* __restrict__ PQg = reinterpret_cast<const unsigned char*>(a.PQ) + (size_t)hh * (a.P >> 5) * TILEB;

    // Position rows (attention_mx.hip): otab entry (q - k) + Sp - 1 + 64 = byte offsets of row delta(q - k) in the SPLIT-unit PQ (x) / PK (y) layouts
    const int otab_max = 2 * Sp - 2 + 128;
    auto block_x = [&](int qb, int t) -> int {
        int idx = qb - 32 * t - 31 + c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return reinterpret_cast<const int*>(a.otab)[2 * idx];
    };
    auto block_xy = [&](int qb, int t) -> int2 {
        int idx = qb - 32 * t - 31 + c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return reinterpret_cast<const int2*>(a.otab)[idx];
    };
    auto load_rows = [&](const unsigned char* base, int off, MxFrag& f) __attribute__((always_inline)) {       // gathered table rows: off = split-form offset
        const unsigned vf = (unsigned)((off & ~8191) + ((off & 8191) >> 1) + h * 512);
        const unsigned vx = (unsigned)(off + 4096 + h * 1024);
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(base + (size_t)vf + s * 1024);
#pragma unroll
        for (int m = 0; m < 2; ++m)
            f.x[m] = cat8(*reinterpret_cast<const i32x4*>(base + (size_t)vx + m * 2048), *reinterpret_cast<const i32x4*>(base + (size_t)vx + (m * 2048 + 16)));
    };
    auto k_tile = [&](const unsigned char* tile, MxFrag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(tile + s * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) f.x[m] = cat8(*reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + lane * 16), *reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + 1024 + lane * 16));
    };
    auto band_store = [&](float* dst, const f32x16& v) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(dst + 8 * g + 4 * h) = (f32x4){v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    };
    // LDS-DMA (attention_mx.hip, NW = 4): wave w moves the 1-KiB pieces 2 w, 2 w + 1 of a K tile and of a V^T tile
    const unsigned off16 = lane * 16, off32 = lane * 32;
    const int piece_src = w * 2048;
    auto uniform_ptr = [](const unsigned char* q) -> const unsigned char* {
        const unsigned long long v = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
    };
    auto dma_pair = [&](const unsigned char* src, unsigned char* dst, const bool mx_piece) __attribute__((always_inline)) {
        if (mx_piece) { glds16_sv(uniform_ptr(src), off32, dst); glds16_sv(uniform_ptr(src + 16), off32, dst + 1024); }
        else { glds16_sv(uniform_ptr(src), off16, dst); glds16_sv(uniform_ptr(src + 1024), off16, dst + 1024); }
    };
    auto dma_k = [&](int t, int slot) { dma_pair(Kg + (size_t)t * TILEB + piece_src, k_ring + slot * TILEB + piece_src, w >= 2); };
    auto dma_v = [&](int t, int slot) { dma_pair(Vg + (size_t)t * TILEB + piece_src, v_ring + slot * TILEB + piece_src, (w & 1) != 0); };

    MxFrag qf;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf.f[s] = *reinterpret_cast<const f16x8*>(Qg + s * 1024 + lane * 16);
#pragma unroll
    for (int m = 0; m < 2; ++m) qf.x[m] = cat8(*reinterpret_cast<const i32x4*>(Qg + 4096 + m * 2048 + lane * 32), *reinterpret_cast<const i32x4*>(Qg + 4096 + m * 2048 + lane * 32 + 16));
    dma_k(0, 0);
    dma_v(0, 0);
    dma_k(nkt > 1 ? 1 : 0, 1);

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }

    MxFrag pq, pqx, pk, kf;
    float cq = 0.f;
    int2 ea = {0, 0}, eb = {0, 0};
    int ex = 0;
    // saturated tiles: pq holds the broadcast fragment (every row = table row d*), cq = Q_q . PK[d*]
    auto sat_prep = [&](int dstar) {
        MxFrag pkb;
        load_rows(PQg, (dstar >> 5) * 8192 + (dstar & 31) * 32, pq);
        load_rows(PKg, (dstar >> 5) * 8192 + glc_pi32(dstar & 31) * 32, pkb);
        f32x16 t;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = 0.f;
        mm_lh_hl(pkb, qf, t);                // every row = PK[d*] . Q_c
        cq = t[0];
    };
    // band prologue: this wave's c2p blocks L(kt_a - 1), L(kt_a); rows of step kt_a; offsets of the next two
    auto band_prep = [&]() {
        f32x16 bacc;
        load_rows(PKg, block_xy(q0, kt_a - 1).y, pk);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
        mm_lh_hl(pk, qf, bacc);
        band_store(c2p_l + c * LROW + 32, bacc);            // ring half 1
        load_rows(PKg, block_xy(q0, kt_a).y, pk);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
        mm_lh_hl(pk, qf, bacc);
        band_store(c2p_l + c * LROW, bacc);                 // ring half 0
        load_rows(PQg, block_x(q0, kt_a), pq);
        if ((kt_a % NQ) == w) load_rows(PQg, block_x(QX, kt_a), pqx);
        ea = block_xy(q0, kt_a + 1);
        eb = block_xy(q0, kt_a + 2);
        ex = block_x(QX, kt_a + 1);
        load_rows(PKg, ea.y, pk);                           // rows of L(kt_a + 1): the c2p block computed during step kt_a
    };

    // O^T += V^T(t) P(t), one step behind the scores
    auto pv = [&](int vslot) __attribute__((always_inline)) {
        const unsigned char* vtile = v_ring + vslot * TILEB;
        const int fl = __builtin_amdgcn_readfirstlane(*flag);
        f16x8 pf[2];
        i32x8 px;
        pf[0] = *reinterpret_cast<const f16x8*>(p_buf + lane * 16);
        pf[1] = *reinterpret_cast<const f16x8*>(p_buf + 1024 + lane * 16);
        px = cat8(*reinterpret_cast<const i32x4*>(p_buf + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(p_buf + 3072 + lane * 16));
        f16x8 vf[2];
        i32x8 vx;
        auto load_v = [&](int d) __attribute__((always_inline)) {
            vf[0] = *reinterpret_cast<const f16x8*>(vtile + d * 4096 + lane * 16);
            vf[1] = *reinterpret_cast<const f16x8*>(vtile + d * 4096 + 1024 + lane * 16);
            vx = cat8(*reinterpret_cast<const i32x4*>(vtile + d * 4096 + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(vtile + d * 4096 + 3072 + lane * 16));
        };
        load_v(0);
        if (fl) {
            const float alpha = f_buf[lane];
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t], pf[t], o0, 0, 0, 0);      // O^T[dd][query c]
        o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o0, 0, 0, 0, SC, 1, SC);
        __builtin_amdgcn_sched_barrier(0);
        load_v(1);
#pragma unroll
        for (int t = 0; t < 2; ++t) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t], pf[t], o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o1, 0, 0, 0, SC, 1, SC);
    };

    int ks = 0;                                            // kt % 3
    // one key tile.  MODE 0: saturated; 1 / 2: band, c2p ring parity 0 / 32
    auto step = [&](const int kt, auto mode_) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_)::value;
        constexpr bool BAND = MODE != 0;
        constexpr int XR = MODE == 2 ? 32 : 0;
        const int ks1 = ks == 2 ? 0 : ks + 1, ks2 = ks == 0 ? 2 : ks - 1;      // (kt + 1) % 3, (kt + 2) % 3
        // ---------------- first half ----------------
        dma_k(kt + 2 < nkt ? kt + 2 : nkt - 1, ks2);
        dma_v(kt + 1 < nkt ? kt + 1 : nkt - 1, ks1);
        k_tile(k_ring + ks * TILEB, kf);
        f32x16 sacc;
        if constexpr (BAND) {
            const bool extra = (kt % NQ) == w;                // wave-uniform: this wave also computes the block nobody owns
            int rbo = rr_base;
            asm volatile("" : "+v"(rbo));                     // (recomputed gather addresses: attention_mx.hip RECOMP)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kc = 16 * (i >> 3) + (i & 7);
                sacc[i] = XR ? c2p_l[c * LROW + ((rbo - kc) ^ 32)] : c2p_l[c * LROW + rr_base - kc];
            }
            f32x16 cacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) cacc[i] = 0.f;
            mm_lh_hl(pk, qf, cacc);                           // c2p of L(kt + 1)  [rr][query c]
            mm_lh_hl(kf, qf, sacc);                           // S^T = K Q^T + c2p
            f32x16 bacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
            mm_hl_lh(pq, kf, bacc);                           // p2c: low block of this wave
            band_store(c2p_l + c * LROW + (XR ^ 32), cacc);
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(s_buf + g * 256 + lane * 4) = (f32x4){sacc[4 * g], sacc[4 * g + 1], sacc[4 * g + 2], sacc[4 * g + 3]};
            if (extra) {
                f32x16 bacc2;
#pragma unroll
                for (int i = 0; i < 16; ++i) bacc2[i] = 0.f;
                mm_hl_lh(pqx, kf, bacc2);                     // ... and, one wave per tile, the high block of the last wave
                band_store(p2c_img + c * LROWP + 32 * NQ, bacc2);
            }
            band_store(p2c_img + c * LROWP + 32 * w, bacc);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = cq;
            mm_lh_hl(kf, qf, sacc);
            mm_lh_hl(kf, pq, sacc);                           // + K_k . PQ[d*] (same for every query column)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(s_buf + g * 256 + lane * 4) = (f32x4){sacc[4 * g], sacc[4 * g + 1], sacc[4 * g + 2], sacc[4 * g + 3]};
        }
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // everything but this half's four DMA pieces has landed
        __builtin_amdgcn_s_barrier();                                    // A(kt)
        // ---------------- second half ----------------
        if (kt > 0) pv(ks2);                                             // V^T(kt - 1) sits in slot (kt - 1) % 3 == (kt + 2) % 3
        if constexpr (BAND) {
            if (kt + 1 < kt_b) {
                if (((kt + 1) % NQ) == w) load_rows(PQg, ex, pqx);
                load_rows(PQg, ea.x, pq);                                // rows of L(kt + 1): p2c of step kt + 1
                load_rows(PKg, eb.y, pk);                                // rows of L(kt + 2): the c2p block computed during step kt + 1
                ea = eb;
                eb = block_xy(q0, kt + 3);
                ex = block_x(QX, kt + 2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // B(kt)
        ks = ks1;
    };

    if (kt_a > 0) sat_prep(a.P - 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // INIT: K(0), K(1), V^T(0) are in the ring
    int kt = 0;
    for (; kt < kt_a; ++kt) step(kt, IC<0>());
    if (kt_a < kt_b) {
        band_prep();
        for (;;) {
            step(kt, IC<1>());
            if (++kt >= kt_b) break;
            step(kt, IC<2>());
            if (++kt >= kt_b) break;
        }
    }
    if (kt_b < nkt) {
        sat_prep(0);
        for (; kt < nkt; ++kt) step(kt, IC<0>());
    }
    // drain: P(nkt - 1) . V^T(nkt - 1), then the row sums
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // A(nkt)
    pv(ks == 0 ? 2 : ks - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // B(nkt)
    __builtin_amdgcn_s_barrier();                                        // FIN: 1 / l is in the factor slot
    if (!active) return;
    const float inv = f_buf[lane];
    // GX context rows (attention_mx.hip)
    unsigned char* row = reinterpret_cast<unsigned char*>(a.CTX) + ((size_t)b * Sp + q0 + c) * 4 * a.H;
    auto store_gx = [&](const f32x16& o, int col0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float own_a = o[8 * p + e] * inv, own_b = o[8 * p + 4 + e] * inv;
                const float got = __shfl_xor(h ? own_a : own_b, 32, 64);
                v[e] = h ? got : own_a;
                v[4 + e] = h ? own_b : got;
            }
            gx_store8(row, col0 + 16 * p + 8 * h, v, gx_act_khi(a.act_sc), gx_act_klo(a.act_sc), a.gx_sat);
        }
    };
    store_gx(o0, 64 * hh);
    store_gx(o1, 64 * hh + 32);
}

}  // namespace

// Same contract as glc_launch_attention_mx.
const char* glc_launch_attention_mxs(hipStream_t st, const AttnArgs& a_in) {
    AttnArgs a = a_in;
    if (!a.gx_sat) a.gx_sat = glc_gx_sat_ptr();              // fp8 range guard of the GX context rows
    if (!a.act_sc) a.act_sc = glc_gx_act_sc();               // ... and the exponent of the activation rows (engine.hip act_sc)
    if (!a.Qh || !a.Kh || !a.Vt || !a.PK || !a.PQ || !a.kbias || !a.klen || !a.kfirst || !a.CTX || !a.otab) return "attention(mxs): null pointer";
    if (a.B <= 0 || a.nh <= 0 || a.Sp <= 0 || a.Sp % 64 || a.H != a.nh * 64 || a.P <= 0 || a.P % 32) return "attention(mxs): bad shape";
    if (a.sel_b || a.tile_flag) return "attention(mxs): no row selection in this kernel";
    const int nqb = (a.Sp + 32 * NQ - 1) / (32 * NQ), bh8 = (a.B * a.nh + 7) / 8 * 8;
    static std::atomic<unsigned> r0{0};
    if (!glc_raise_lds_limit(attn_mxs_kernel, MXS_LDS, r0)) return "attention(mxs): cannot raise the dynamic LDS limit";
    hipLaunchKernelGGL(attn_mxs_kernel, dim3(nqb * bh8), dim3(512), MXS_LDS, st, a);
    return nullptr;
}
