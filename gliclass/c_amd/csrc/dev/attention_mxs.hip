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
//   first half   M: DMA V^T(t) | K(t) fragments | S^T(t) = ring gather + K Q^T -> S buffer | p2c(t) = PQ K^T (+ the leaving block, one wave in
//                   four) -> shared image | c2p(t+1) = PK Q_i^T for the tile i this wave's PK block belongs to at t+1 (stored to i's ring after A)
//                X: second part of softmax(t-1): row sums, f16 / fp8 split -> P buffer (+ rescale factors, flag)
//   A(t)         S(t), image(t), P(t-1) are published
//   second half  M: DMA K(t+2) | c2p block -> ring of tile i | O^T += V^T(t-1) P(t-1) (after the deferred rescale, if X flagged one) | every
//                   fourth step: the wave's next PQ and PK block requested
//                X: S(t) + image gather + key bias, maximum, rescale decision, exponentials
//   B(t)         S buffer, image and P buffer are free again
// so P.V runs one step behind the scores and every buffer is single: between a write and the reads of it lies one barrier, between
// those reads and the next write the other.  K and V^T rings of two slots (V^T(t) requested at the start of the first half of step t, K(t+2) at the
// start of the second.  LDS: c2p rings 34 KB, p2c image 20.5 KB, K / V^T rings 32 KB, the four Q tiles 32 KB,
// S and P buffers 32 KB, factors 1 KB = 151.6 KB, one workgroup (8 waves) per CU.
//
// Saturated key tiles (delta constant: attention_wg.hip) take the same pipeline with S^T = cq + K Q^T + K PQ[d*]^T and no image.
#include <stdio.h>
#include <stdlib.h>
#include "../glc_common.h"
#include "../glc_kernels.h"
#include "../glc_layout.h"
#include "../glc_pfrag.h"

namespace {

constexpr float RESCALE_THR = 8.0f;   // log2 units (as attention_mx.hip)
constexpr int NQ = 4;                 // query tiles = matrix waves per workgroup
constexpr int LROW = 68;              // floats per c2p ring row (2 blocks of 32 + 4 pad)
constexpr int LROWP = 32 * (NQ + 1) + 4;            // floats per p2c image row
constexpr int TILEB = GLC_MXT_BYTES;
constexpr int OFF_RING = 0;
constexpr int OFF_IMG = OFF_RING + NQ * 32 * LROW * 4;
constexpr int OFF_K = OFF_IMG + 32 * LROWP * 4;
constexpr int OFF_V = OFF_K + 2 * TILEB;
constexpr int OFF_Q = OFF_V + 2 * TILEB;            // the workgroup's four Q tiles, ring-image layout (read by the wave that computes a tile's c2p block)
constexpr int OFF_PX = OFF_Q + NQ * TILEB;          // the PQ block that leaves the band this step (image slot NQ), ring-image layout: every matrix wave computes a 16 x 16 quarter of its p2c block
constexpr int OFF_S = OFF_PX + TILEB;
constexpr int OFF_P = OFF_S + NQ * 4096;
constexpr int OFF_F = OFF_P + NQ * 4096;            // per tile: 64 floats (rescale factor / final 1 / l per lane) ...
constexpr int OFF_FLAG = OFF_F + NQ * 256;          // ... and one flag word per tile
constexpr int MXS_LDS = OFF_FLAG + 16;
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

// DIAG: s_memtime stamps at the phase boundaries of a step, summed per wave (glc_debug_attn_bench prints them; developer builds only).
template <bool DIAG, int XPRIO = 0>
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
    unsigned seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, nsteps = 0;      // (32-bit sums: a launch is < 2^32 ticks)
    const unsigned long long clk0 = DIAG ? __builtin_amdgcn_s_memtime() : 0, rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {      // time since the previous stamp goes to segment k (k < 0: start)
        if constexpr (DIAG) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
            if (k >= 0) seg[k] += t - tlast;
            tlast = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stamps_out = [&]() {
        if constexpr (DIAG) {
            if (a.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {     // 64 workgroups of XCD 0
                unsigned long long* o = a.stamps + ((size_t)(blockIdx.x >> 3) * 8 + wave) * 10;
                for (int k = 0; k < 8; ++k) o[k] = seg[k];
                const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk0, dr = __builtin_amdgcn_s_memrealtime() - rt0;
                o[8] = dr ? dc * 1000 / dr : 0; o[9] = nsteps;
            }
        }
    };

    if (!is_m) {
        // =============================== softmax wave ===============================
        const float* __restrict__ kb = a.kbias + (size_t)b * Sp;
        const int kfirst = a.kfirst[b];
        if (XPRIO == 1) __builtin_amdgcn_s_setprio(1);
        const int foff = 8 * h;
        float m = -3.0e38f, l = 0.f;
        float one_f = 1.0f;
        asm volatile("" : "+s"(one_f));      // opaque to the optimiser: fma(p, 1, -half) stays a v_fma_mix_f32
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // INIT
        for (int kt = 0; kt < nkt; ++kt) {
            const bool band = kt >= kt_a && kt < kt_b;
            const int k0 = kt * 32;
            if (kt == kt_a && band) __builtin_amdgcn_s_barrier();       // band entry: the matrix waves publish the first leaving PQ block
            stamp(-1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // A(kt): S(kt) and image(kt) are complete
            stamp(0);                                                   // X seg 0: P stores landed + barrier A
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
            stamp(1);                                                   // X seg 1: S + image gather + bias, maximum, exponentials
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // B(kt): S buffer, image and P buffer are free
            stamp(2);                                                   // X seg 2: barrier B
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(sv[i]));    // (the split and the sums stay behind the barrier: this part runs beside M's long half)
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
                int wh;
                asm volatile("" : "=v"(wh));                             // (both halves are written below: no zero to start from)
                wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * q], sv[4 * q + 1], wh, false);
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
                v2i16 wl2;
                asm volatile("" : "=v"(wl2));
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
            stamp(3);                                                   // X seg 3: row sums, P split, P stores issued
            if constexpr (DIAG) ++nsteps;
        }
        stamps_out();
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
    if (XPRIO == 2) __builtin_amdgcn_s_setprio(1);          // (developer A/B: XPRIO 1 = the softmax wave at priority 1; 2 / 3 = the matrix wave at priority 1 / 3)
    if (XPRIO == 3) __builtin_amdgcn_s_setprio(3);
    // e8m0 scales of the block-scaled MFMA (attention_mx.hip)
    const int SC = h ? (127 | ((127 - GLC_GX_SHIFT) << 8)) : ((127 - GLC_GX_SHIFT) | (127 << 8));
    auto mm_lh_hl = [&](const MxFrag& lh, const MxFrag& hl, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh.f[s], hl.f[s], acc, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(lh.x[m], hl.x[m], acc, 0, 0, 0, SC, 1, SC);
    };

    // the same products with the position block (PFrag) as first (A) operand: c2p = PK (lo8 | hi8) x Q, p2c = PQ (hi8 | lo8) x K
    auto mm_p_lh_hl = [&](const PFrag& lh, const MxFrag& hl, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh.f[s], hl.f[s], acc, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(lh.xa[m], lh.xb[m]), hl.x[m], acc, 0, 0, 0, SC, 1, SC);
    };
    auto mm_p_hl_lh_f16 = [&](const PFrag& hl, const MxFrag& lh, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl.f[s], lh.f[s], acc, 0, 0, 0);
    };
    auto mm_p_hl_lh_x = [&](const PFrag& hl, const MxFrag& lh, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(hl.xa[m], hl.xb[m]), lh.x[m], acc, 0, 0, 1, SC, 0, SC);
    };
    auto mm_lh_p_hl = [&](const MxFrag& lh, const PFrag& hl, f32x16& acc) __attribute__((always_inline)) {      // saturated tiles: K x PQ[d*]
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh.f[s], hl.f[s], acc, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(lh.x[m], cat8(hl.xa[m], hl.xb[m]), acc, 0, 0, 0, SC, 1, SC);
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
    auto rows_vf = [&](int off) -> unsigned { return (unsigned)((off & ~8191) + ((off & 8191) >> 1) + h * 512); };
    auto rows_vx = [&](int off) -> unsigned { return (unsigned)(off + 4096 + h * 1024); };
    auto load_rows_p = [&](const unsigned char* base, int off, PFrag& f) __attribute__((always_inline)) {
        const unsigned vf = rows_vf(off), vx = rows_vx(off);
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(base + (size_t)vf + s * 1024);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f.xa[m] = *reinterpret_cast<const i32x4*>(base + (size_t)vx + m * 2048);
            f.xb[m] = *reinterpret_cast<const i32x4*>(base + (size_t)vx + (m * 2048 + 16));
        }
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
    // a 32 x 32-layout operand tile from registers into LDS in the ring-image layout (k_tile reads it back)
    auto frag_to_lds = [&](unsigned char* dst, const MxFrag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<f16x8*>(dst + s * 1024 + lane * 16) = f.f[s];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            *reinterpret_cast<i32x4*>(dst + 4096 + m * 2048 + lane * 16) = (i32x4){f.x[m][0], f.x[m][1], f.x[m][2], f.x[m][3]};
            *reinterpret_cast<i32x4*>(dst + 4096 + m * 2048 + 1024 + lane * 16) = (i32x4){f.x[m][4], f.x[m][5], f.x[m][6], f.x[m][7]};
        }
    };
    auto pfrag_to_lds = [&](unsigned char* dst, const PFrag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<f16x8*>(dst + s * 1024 + lane * 16) = f.f[s];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            *reinterpret_cast<i32x4*>(dst + 4096 + m * 2048 + lane * 16) = f.xa[m];
            *reinterpret_cast<i32x4*>(dst + 4096 + m * 2048 + 1024 + lane * 16) = f.xb[m];
        }
    };
    // the tile's Q also goes to the static area: the wave that holds a tile's next PK block reads its Q from there
    unsigned char* q_static = smem_mxs + OFF_Q;
    unsigned char* px_area = smem_mxs + OFF_PX;
    frag_to_lds(q_static + w * TILEB, qf);
    dma_k(0, 0);
    dma_v(0, 0);
    if (nkt > 1) dma_k(1, 1);

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }

    // Position blocks live in registers for their whole life (rel-block g = j - t of the workgroup, j the image slot):
    //   PQ block g: wave g mod 4 holds it while j runs 0 .. 3 and computes its p2c block against every new K tile; after its last step it copies
    //               the rows to the px area (LDS) — the block's fifth step (j == 4, "the block nobody owns") is computed by ALL four waves, a
    //               16 x 16 quarter each (v_mfma_*_16x16x32 / 16x16x128), 64 cycles of matrix pipe per wave instead of 256 on one — and takes
    //               the block that enters (j == 0) into the same registers.
    //   PK block g: wave g mod 4 holds it for the four steps in which it is the NEW c2p block of query tile 0, 1, 2, 3 in turn (Q from the static
    //               area); rows in REVERSED order, so that the ring holds each block reversed and the gather's register pairs ascend.
    // One PQ and one PK block (one otab entry: x / y) are requested per wave every fourth step: 16 KB per step and workgroup instead of 72.
    PFrag pq, pk;
    MxFrag kf;
    float cq = 0.f;
    int2 e_pre = {0, 0};
    auto block_xy_rev = [&](int qb, int t) -> int2 {      // block_xy with the block's rows in reversed lane order
        int idx = qb - 32 * t - c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return reinterpret_cast<const int2*>(a.otab)[idx];
    };
    // saturated tiles: pq holds the broadcast fragment (every row = table row d*), cq = Q_q . PK[d*]
    auto sat_prep = [&](int dstar) {
        MxFrag pkb;
        load_rows_p(PQg, (dstar >> 5) * 8192 + (dstar & 31) * 32, pq);
        load_rows(PKg, (dstar >> 5) * 8192 + glc_pi32(dstar & 31) * 32, pkb);
        f32x16 t;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = 0.f;
        mm_lh_hl(pkb, qf, t);                // every row = PK[d*] . Q_c
        cq = t[0];
    };
    // band prologue: this wave's own c2p blocks L(kt_a - 1), L(kt_a) (reversed; half 0 / half 1); the position blocks it holds at step kt_a;
    // wave 0 publishes the block that leaves at step kt_a.  Ends with a workgroup barrier (the softmax waves join it).
    auto band_prep = [&]() {
        f32x16 bacc;
        MxFrag tmp;
        load_rows(PKg, block_xy_rev(q0, kt_a - 1).y, tmp);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
        mm_lh_hl(tmp, qf, bacc);
        band_store(c2p_l + c * LROW, bacc);                 // ring half 0
        load_rows(PKg, block_xy_rev(q0, kt_a).y, tmp);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
        mm_lh_hl(tmp, qf, bacc);
        band_store(c2p_l + c * LROW + 32, bacc);            // ring half 1
        if (w == 0) {
            load_rows(PQg, block_x(QX, kt_a), tmp);
            frag_to_lds(px_area, tmp);
        }
        const int jm0 = (w + kt_a) & 3;
        load_rows_p(PQg, block_x(Q0 + 32 * jm0, kt_a), pq);
        load_rows_p(PKg, block_xy_rev(Q0 + 32 * ((w + kt_a + 1) & 3), kt_a + 1).y, pk);
        const int2 ef = block_xy(Q0, jm0 == 3 ? kt_a + 1 : kt_a + 2), er = block_xy_rev(Q0, jm0 == 3 ? kt_a + 1 : kt_a + 2);
        e_pre.x = ef.x; e_pre.y = er.y;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // O^T += V^T(t) P(t), one step behind the scores
    auto pv = [&](int vslot) __attribute__((always_inline)) {
        const unsigned char* vtile = v_ring + vslot * TILEB;
        const int flv = *flag;
        const float alpha = f_buf[lane];
        f16x8 pf[2];
        i32x8 px;
        pf[0] = *reinterpret_cast<const f16x8*>(p_buf + lane * 16);
        pf[1] = *reinterpret_cast<const f16x8*>(p_buf + 1024 + lane * 16);
        px = cat8(*reinterpret_cast<const i32x4*>(p_buf + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(p_buf + 3072 + lane * 16));
        f16x8 vf[2], vg[2];
        i32x8 vx, vy;
        vf[0] = *reinterpret_cast<const f16x8*>(vtile + lane * 16);
        vf[1] = *reinterpret_cast<const f16x8*>(vtile + 1024 + lane * 16);
        vx = cat8(*reinterpret_cast<const i32x4*>(vtile + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(vtile + 3072 + lane * 16));
        vg[0] = *reinterpret_cast<const f16x8*>(vtile + 4096 + lane * 16);
        vg[1] = *reinterpret_cast<const f16x8*>(vtile + 4096 + 1024 + lane * 16);
        vy = cat8(*reinterpret_cast<const i32x4*>(vtile + 4096 + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(vtile + 4096 + 3072 + lane * 16));
        if (__builtin_amdgcn_readfirstlane(flv)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t], pf[t], o0, 0, 0, 0);      // O^T[dd][query c]
        o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o0, 0, 0, 0, SC, 1, SC);
#pragma unroll
        for (int t = 0; t < 2; ++t) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vg[t], pf[t], o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vy, px, o1, 0, 0, 0, SC, 1, SC);
    };
    auto store_s = [&](const f32x16& sacc) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(s_buf + g * 256 + lane * 4) = (f32x4){sacc[4 * g], sacc[4 * g + 1], sacc[4 * g + 2], sacc[4 * g + 3]};
    };
    // the leaving block's quarter of this wave: rels 16 qa .. + 15 x key slots 16 qb .. + 15
    const int qa = w >> 1, qb = w & 1, l15 = lane & 15, lg = lane >> 4;
    // byte offsets inside a ring-image tile: f16 unit 2 s' + (lg >> 1) at + s' * 2048; MX step (lg >> 1): first part, second at + 1024
    const int qrow_a = (32 * (lg & 1) + 16 * qa + l15) * 16 + (lg >> 1) * 1024, qrow_b = (32 * (lg & 1) + 16 * qb + l15) * 16 + (lg >> 1) * 1024;
    const int qx_a = (32 * (lg & 1) + 16 * qa + l15) * 16 + (lg >> 1) * 2048, qx_b = (32 * (lg & 1) + 16 * qb + l15) * 16 + (lg >> 1) * 2048;
    typedef float f32x4q __attribute__((ext_vector_type(4)));

    // one key tile.  MODE 0: saturated; 1 / 2: band, c2p ring parity 0 / 32.  On entry kf holds the fragments of K(kt).
    auto step = [&](const int kt, auto mode_) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_)::value;
        constexpr bool BAND = MODE != 0;
        constexpr int XR = MODE == 2 ? 32 : 0;
        // ---------------- first half ----------------
        stamp(-1);
        pfrag_wait_all(pq, pk);                                          // K(kt + 1), V^T(kt - 1) and the position rows requested during the last half step: all at least half a step old
        if (kt > 0) dma_v(kt, kt & 1);                                   // (its slot held V^T(kt - 2): read before B(kt - 1)); needed after A(kt + 1): waited for at the top of step kt + 1
        stamp(0);                                                        // M seg 0: request wait + DMA issue
        {
            const unsigned char* kc_ = k_ring + (kt & 1) * TILEB;
#pragma unroll
            for (int m = 0; m < 2; ++m) kf.x[m] = cat8(*reinterpret_cast<const i32x4*>(kc_ + 4096 + m * 2048 + lane * 16), *reinterpret_cast<const i32x4*>(kc_ + 4096 + m * 2048 + 1024 + lane * 16));
        }
        f32x16 sacc;
        if constexpr (BAND) {
            const unsigned char* ktile = k_ring + (kt & 1) * TILEB;
            int rbo = 63 - rr_base;
            asm volatile("" : "+v"(rbo));                     // (recomputed gather addresses: attention_mx.hip RECOMP)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kc = 16 * (i >> 3) + (i & 7);
                sacc[i] = XR ? c2p_l[c * LROW + ((rbo + kc) ^ 32)] : c2p_l[c * LROW + 63 - rr_base + kc];
            }
            f32x16 bacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
            mm_p_hl_lh_f16(pq, kf, bacc);                     // p2c block of image slot jm, f16 part: every operand is in registers — covers the LDS latency of the gather and of K's MX steps
            asm volatile("" : "+v"(bacc), "+v"(sacc));        // (order: the gathered values are waited for BEHIND those four MFMAs)
            mm_lh_hl(kf, qf, sacc);                           // S^T = K Q^T + c2p
            // the leaving block's quarter: A = PQ rows from the px area, B = K(kt) key slots, both in the 16 x 16 operand layouts
            f16x8 af[2], bf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                af[s2] = *reinterpret_cast<const f16x8*>(px_area + s2 * 2048 + qrow_a);
                bf[s2] = *reinterpret_cast<const f16x8*>(ktile + s2 * 2048 + qrow_b);
            }
            mm_p_hl_lh_x(pq, kf, bacc);                       // ... its cross terms
            __builtin_amdgcn_sched_barrier(0);
            f32x4q qacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) qacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s2], bf[s2], qacc, 0, 0, 0);
            store_s(sacc);
            __builtin_amdgcn_sched_barrier(0);
            {
                const i32x8 ax = cat8(*reinterpret_cast<const i32x4*>(px_area + 4096 + qx_a), *reinterpret_cast<const i32x4*>(px_area + 4096 + 1024 + qx_a));
                const i32x8 bx = cat8(*reinterpret_cast<const i32x4*>(ktile + 4096 + qx_b), *reinterpret_cast<const i32x4*>(ktile + 4096 + 1024 + qx_b));
                qacc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ax, bx, qacc, 0, 0, 0, 127 - GLC_GX_SHIFT, 0, 127);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int jm = (w + kt) & 3;                      // image slot of this wave's block
            band_store(p2c_img + c * LROWP + 32 * jm, bacc);
            *reinterpret_cast<f32x4q*>(p2c_img + (16 * qb + l15) * LROWP + 32 * NQ + 16 * qa + 4 * lg) = qacc;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = cq;
            mm_lh_hl(kf, qf, sacc);
            mm_lh_p_hl(kf, pq, sacc);                         // + K_k . PQ[d*] (same for every query column)
            store_s(sacc);
        }
        MxFrag qi;                                                       // the Q tile this wave's PK block belongs to this step: static data, read BEFORE the barrier —
        const int iq = (w + kt + 1) & 3;                                 // the c2p MFMAs behind A(kt) start on operands that are in registers
        if constexpr (BAND) k_tile(q_static + iq * TILEB, qi);
        stamp(1);                                                        // M seg 1: ring gather, operand reads, MFMA issue, stores issued
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // stores landed (this half's V^T pieces stay in flight)
        __builtin_amdgcn_s_barrier();                                    // A(kt)
        stamp(2);                                                        // M seg 2: stores landed + barrier A
        // ---------------- second half ----------------
        if (kt + 2 < nkt) dma_k(kt + 2, kt & 1);                         // (K(kt)'s slot: every wave holds its fragments since A(kt))
        // P(kt - 1), V^T(kt - 1) and the rescale flag / factors are requested first; the c2p block of step kt + 1 — operands: this wave's PK
        // block (registers) and the Q tile it belongs to this step (static area) — is computed under their latency
        const unsigned char* vtile = v_ring + ((kt - 1) & 1) * TILEB;
        const int flv = *flag;
        const float alpha = f_buf[lane];
        f16x8 pf[2], vf[2], vg[2];
        i32x8 px, vx, vy;
        pf[0] = *reinterpret_cast<const f16x8*>(p_buf + lane * 16);
        pf[1] = *reinterpret_cast<const f16x8*>(p_buf + 1024 + lane * 16);
        px = cat8(*reinterpret_cast<const i32x4*>(p_buf + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(p_buf + 3072 + lane * 16));
        vf[0] = *reinterpret_cast<const f16x8*>(vtile + lane * 16);
        vf[1] = *reinterpret_cast<const f16x8*>(vtile + 1024 + lane * 16);
        vx = cat8(*reinterpret_cast<const i32x4*>(vtile + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(vtile + 3072 + lane * 16));
        vg[0] = *reinterpret_cast<const f16x8*>(vtile + 4096 + lane * 16);
        vg[1] = *reinterpret_cast<const f16x8*>(vtile + 4096 + 1024 + lane * 16);
        vy = cat8(*reinterpret_cast<const i32x4*>(vtile + 4096 + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(vtile + 4096 + 3072 + lane * 16));
        if constexpr (BAND) {
            f32x16 cacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) cacc[i] = 0.f;
            mm_p_lh_hl(pk, qi, cacc);                         // [rr reversed][query]
            __builtin_amdgcn_sched_barrier(0);
            if (kt > 0) {
                if (__builtin_amdgcn_readfirstlane(flv)) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t], pf[t], o0, 0, 0, 0);      // O^T[dd][query c]
                o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o0, 0, 0, 0, SC, 1, SC);
            }
            __builtin_amdgcn_sched_barrier(0);
            band_store(reinterpret_cast<float*>(smem_mxs + OFF_RING) + (size_t)iq * 32 * LROW + c * LROW + XR, cacc);      // (tile iq gathered this half's victim before A(kt))
        } else if (kt > 0) {
            if (__builtin_amdgcn_readfirstlane(flv)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t], pf[t], o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o0, 0, 0, 0, SC, 1, SC);
        }
        if (kt > 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vg[t], pf[t], o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vy, px, o1, 0, 0, 0, SC, 1, SC);
        }
        __builtin_amdgcn_sched_barrier(0);
        {                                                                // K(kt + 1)'s f16 units: published by A(kt) (after the last tile: a stale slot, unused)
            const unsigned char* kn = k_ring + ((kt + 1) & 1) * TILEB;
#pragma unroll
            for (int s = 0; s < 4; ++s) kf.f[s] = *reinterpret_cast<const f16x8*>(kn + s * 1024 + lane * 16);
        }
        if constexpr (BAND) {
            const int ph = (kt + 1 < kt_b) ? ((w + kt) & 3) : 0;
            // ph == 3: the block's last step as slot 3 — its rows go to the px area, the block that enters takes the registers;
            // ph == 2: the PK block that serves tile 0 at step kt + 1; ph == 1: the table entry of both (block kt + 3 of tile 0)
            if (ph == 3) pfrag_to_lds(px_area, pq);
            pfrag_load_if(ph == 3, PQg, rows_vf(e_pre.x), rows_vx(e_pre.x), pq);
            pfrag_load_if(ph == 2, PKg, rows_vf(e_pre.y), rows_vx(e_pre.y), pk);
            if (ph == 1) { e_pre.x = block_xy(Q0, kt + 3).x; e_pre.y = block_xy_rev(Q0, kt + 3).y; }
        }
        stamp(3);                                                        // M seg 3: second half: K DMA, c2p block stored, P.V, K fragments of the next step, position-row requests
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // B(kt)
        stamp(4);                                                        // M seg 4: barrier B
        if constexpr (DIAG) ++nsteps;
    };

    if (kt_a > 0) sat_prep(a.P - 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // INIT: K(0), K(1), V^T(0) are in the ring, the Q tiles in the static area
    k_tile(k_ring, kf);
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
    stamps_out();
    // drain: P(nkt - 1) . V^T(nkt - 1), then the row sums
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // A(nkt)
    pv((nkt - 1) & 1);
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
#ifdef GLC_DEVELOPER
    static std::atomic<unsigned> r1{0};
    if (a.stamps) {
        if (!glc_raise_lds_limit(attn_mxs_kernel<true>, MXS_LDS, r1)) return "attention(mxs): cannot raise the dynamic LDS limit";
        hipLaunchKernelGGL(attn_mxs_kernel<true>, dim3(nqb * bh8), dim3(512), MXS_LDS, st, a);
        return nullptr;
    }
#endif
    if (a.stamps) return "attention(mxs): the stamped build exists in developer builds only (make DEV=1)";
#ifdef GLC_DEVELOPER
    {
        static std::atomic<unsigned> rp1{0}, rp2{0}, rp3{0};
        const int xp = a.variant & 3;
        if (xp == 1) { if (!glc_raise_lds_limit(attn_mxs_kernel<false, 1>, MXS_LDS, rp1)) return "lds"; hipLaunchKernelGGL((attn_mxs_kernel<false, 1>), dim3(nqb * bh8), dim3(512), MXS_LDS, st, a); return nullptr; }
        if (xp == 2) { if (!glc_raise_lds_limit(attn_mxs_kernel<false, 2>, MXS_LDS, rp2)) return "lds"; hipLaunchKernelGGL((attn_mxs_kernel<false, 2>), dim3(nqb * bh8), dim3(512), MXS_LDS, st, a); return nullptr; }
        if (xp == 3) { if (!glc_raise_lds_limit(attn_mxs_kernel<false, 3>, MXS_LDS, rp3)) return "lds"; hipLaunchKernelGGL((attn_mxs_kernel<false, 3>), dim3(nqb * bh8), dim3(512), MXS_LDS, st, a); return nullptr; }
    }
#endif
    if (!glc_raise_lds_limit(attn_mxs_kernel<false>, MXS_LDS, r0)) return "attention(mxs): cannot raise the dynamic LDS limit";
    hipLaunchKernelGGL(attn_mxs_kernel<false>, dim3(nqb * bh8), dim3(512), MXS_LDS, st, a);
    return nullptr;
}
