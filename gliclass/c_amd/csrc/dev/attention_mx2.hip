// DeBERTa-v2/v3 disentangled self-attention on MX tiles, round 4: position terms in BUCKET space, private to each wave.
//
// score[q][k] = Q_q.K_k + Q_q.PK[delta(q-k)] + K_k.PQ[delta(q-k)]   (attention.hip header; modeling_deberta_v2.py:276-345), products as in
// attention_mx.hip (a_hi*b_hi in f16 MFMAs + both cross terms in one block-scaled fp8 MFMA).  What changes is how the two position terms
// are produced.  attention_mx.hip keeps them as Toeplitz bands over the relative DISTANCE: one new 32-row position block per key tile
// and per term, the p2c band shared by the workgroup through one LDS image (two more workgroup barriers per tile).  But delta() is a
// non-decreasing staircase: beyond |q-k| = 128 consecutive distances share log buckets, so the 63 distances of a (query tile, key tile)
// pair touch 63 table rows near the diagonal and 6-30 further out.  Here both terms live in delta space and belong to ONE wave:
//   c2p  T[q][delta & 63]   a ring of two ALIGNED 32-row blocks of the table (block B = rows 32B..32B+31 = one PK tile, no gather): a new
//        block only when the window's lowest block changes — every key tile near the diagonal, every 2-4 tiles further out, never on
//        saturated tiles: 17 instead of 26 blocks per 32 key tiles at S = 1024.
//   p2c  I[key][delta & 31] per key tile: K_t . PQ[window]^T.  69 % of the (query tile, key tile) pairs at S = 1024 need 32 table rows
//        (from delta_s = delta_min & ~3 on): one block, one store, one gather.  The rest (near the diagonal) need the two aligned
//        blocks of their window: two passes through the same 4-KiB image, the element taking the pass its delta's block parity names.
//        No shared image, no image barriers.
//   both are read through ONE packed index per score element, 4 * (delta(q - k) & 63), from a host-built table (16 bytes per lane and
//   tile: one global_load_dwordx4); c2p enters S^T = K Q^T as its initial accumulator, p2c is added after it.
// LDS rows are unpadded and XOR-swizzled in 16-byte granules by (row & 7) (all 32 lanes of a store instruction write the same columns
// of different rows); a gather address is (lane constant) ^ (index byte) — one VALU op per element.
// K and V^T tiles go through LDS-DMA rings of two slots each, shared by the workgroup, with ONE workgroup barrier per key tile: after
// barrier(t) every wave has its K(t) fragments and has left V^T(t-1), so V^T(t+1) and K(t+2) are requested there; each wave waits for
// its own pieces at the top of tile t+1 and barrier(t+1) publishes them.  Workgroup = 4 waves = 4 consecutive query tiles of one
// (batch, head), 80 KiB of LDS (T 4 x 8, I 4 x 4, K 16, V^T 16), two workgroups per CU.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../glc_common.h"
#include "../glc_kernels.h"
#include "../glc_layout.h"

namespace {

constexpr float RESCALE_THR = 8.0f;   // log2 units (as attention.hip)
constexpr int NW = 4;
constexpr int TILEB = GLC_MXT_BYTES;
constexpr int T_OFF = 0, I_OFF = NW * 8192, K_OFF = I_OFF + NW * 4096, V_OFF = K_OFF + 2 * TILEB;
constexpr size_t LDS_BYTES = V_OFF + 2 * TILEB;          // 80 KiB: two workgroups per CU
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct MxFrag { f16x8 f[4]; i32x8 x[2]; };      // a 32-row x 64-column operand tile in registers (32 VGPRs)

__device__ __forceinline__ void glds16_sv(const unsigned char* ubase, unsigned lane_off, void* l) {      // attention_wg.hip
    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)l;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la), "v"(lane_off), "s"(ubase) : "memory");
}
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wg_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ i32x8 cat8(const i32x4& a, const i32x4& b) {
    i32x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

extern __shared__ __attribute__((aligned(256))) unsigned char smem_mx2[];

// DIAG: s_memtime stamps at the phase boundaries of a generic tile, summed per wave (glc_debug_attn_bench prints them).
// ABL: timing-only builds (wrong results; AttnArgs::variant bits 8-9 through glc_debug_attn_bench): 1 = no window-row / PK-block requests after the
// entry, 2 = no LDS gathers (c2p, p2c) and image stores, 4 = no softmax arithmetic, 8 = no block-scaled MFMAs, 16 = no ring barrier / DMA (bit mask).
template <bool DIAG = false, int ABL = 0>
__global__ __launch_bounds__(64 * NW, 2) void attn_mx2_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int Sp = a.Sp;
    const int SC = h ? (127 | ((127 - GLC_GX_SHIFT) << 8)) : ((127 - GLC_GX_SHIFT) | (127 << 8));      // attention_mx.hip
    auto mm_lh_hl = [&](const MxFrag& lh, const MxFrag& hl, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh.f[s], hl.f[s], acc, 0, 0, 0);
        if constexpr (!(ABL & 8)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(lh.x[m], hl.x[m], acc, 0, 0, 0, SC, 1, SC);
        }
    };
    auto mm_hl_lh = [&](const MxFrag& hl, const MxFrag& lh, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl.f[s], lh.f[s], acc, 0, 0, 0);
        if constexpr (!(ABL & 8)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hl.x[m], lh.x[m], acc, 0, 0, 1, SC, 0, SC);
        }
    };

    const int nqb = (Sp + 32 * NW - 1) / (32 * NW);
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int bh = xcd + 8 * (jj / nqb);
    const int Q0 = (jj % nqb) * 32 * NW;
    if (bh >= a.B * a.nh) return;
    const int b = bh / a.nh, hh = bh - b * a.nh;
    const int q0 = Q0 + 32 * wave;
    const bool active = q0 < Sp;
    const int q0m = active ? q0 : Sp - 32;
    const int klen = a.klen[b];
    if (Q0 >= klen && Q0 > 0) {
        // every query of this block lies past the row's last attended token: never read by an attended row; store zeros and leave
        if (active) {
            unsigned char* row = reinterpret_cast<unsigned char*>(a.CTX) + ((size_t)b * Sp + q0 + c) * 4 * a.H + (size_t)(2 * hh) * 128 + h * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(row + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
        }
        return;
    }

    const int nt = Sp >> 5, qt = q0m >> 5;
    const unsigned char* __restrict__ Qg = reinterpret_cast<const unsigned char*>(a.Qh) + ((size_t)bh * nt + qt) * TILEB;
    const unsigned char* __restrict__ Kg = reinterpret_cast<const unsigned char*>(a.Kh) + (size_t)bh * nt * TILEB;
    const unsigned char* __restrict__ Vg = reinterpret_cast<const unsigned char*>(a.Vt) + (size_t)bh * nt * TILEB;
    const unsigned char* __restrict__ PKg = reinterpret_cast<const unsigned char*>(a.PK) + (size_t)hh * (a.P >> 5) * TILEB;
    const unsigned char* __restrict__ PQg = reinterpret_cast<const unsigned char*>(a.PQ) + (size_t)hh * (a.P >> 5) * TILEB;
    const unsigned char* __restrict__ IDXg = reinterpret_cast<const unsigned char*>(a.idx16);
    const float* __restrict__ kb = a.kbias + (size_t)b * Sp;

    int nkt = (klen + 31) >> 5;
    nkt = nkt < 1 ? 1 : (nkt > nt ? nt : nkt);
    const int kfirst = a.kfirst[b];
    const int foff = 8 * h;

    // ---- LDS: per-wave c2p table T (32 x 64 floats) and p2c image I (32 x 32 floats), both swizzled; the K and V^T rings ----
    unsigned char* const Tw = smem_mx2 + T_OFF + wave * 8192;
    unsigned char* const Iw = smem_mx2 + I_OFF + wave * 4096;
    unsigned char* const kring = smem_mx2 + K_OFF;
    unsigned char* const vring = smem_mx2 + V_OFF;
    const unsigned swzc = (unsigned)(c & 7) << 4;
    const unsigned Tc = (unsigned)(T_OFF + wave * 8192 + 256 * c) | swzc;                 // c2p gather: T[c][col] at Tc ^ (4 col), col = delta & 63
    const unsigned Ih = (unsigned)(I_OFF + wave * 4096 + 512 * h + 64 * h);               // p2c gather: I[e + 4h + 8g][col] at (Ih ^ (4 col ^ 16 e)) + 128 e + 1024 g, col = delta & 31

    // fragments of one contiguous 8-KiB tile in memory (the query tile; an aligned block of the PK table)
    auto load_tile = [&](const unsigned char* tile, MxFrag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(tile + s * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) f.x[m] = cat8(*reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + lane * 32), *reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + lane * 32 + 16));
    };
    // 32 consecutive rows of the PQ table from row r0 on (Q layout: row rho = tile rho >> 5, slot rho & 31), clamped at the table's end
    // (rows past `last`, the highest table row the window needs, repeat it: their lanes fall into cache lines the instruction fetches anyway)
    auto load_rows_win = [&](int r0, int last, MxFrag& f) __attribute__((always_inline)) {
        int rho = r0 + c;
        rho = rho > last ? last : rho;
        const unsigned tb = (unsigned)(rho & ~31) << 8, sl = (unsigned)(rho & 31);
        const unsigned vf = tb + sl * 16 + h * 512;
        const unsigned vx = tb + 4096 + sl * 32 + h * 1024;
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(PQg + (size_t)vf + s * 1024);
#pragma unroll
        for (int m = 0; m < 2; ++m)
            f.x[m] = cat8(*reinterpret_cast<const i32x4*>(PQg + (size_t)vx + m * 2048), *reinterpret_cast<const i32x4*>(PQg + (size_t)vx + (m * 2048 + 16)));
    };
    // K tile t from ring slot t & 1: f16 units as they are, MX steps re-arranged by the DMA into [64 lanes x first | 64 lanes x second] (attention_mx.hip)
    auto k_tile = [&](int t, MxFrag& f) __attribute__((always_inline)) {
        const unsigned char* tile = kring + (size_t)(t & 1) * TILEB;
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(tile + s * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) f.x[m] = cat8(*reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + lane * 16), *reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + 1024 + lane * 16));
    };
    // LDS-DMA: wave w moves the 1-KiB piece pair w of a tile.  K tile: pairs 0-1 the f16 units, 2-3 the two MX steps; V^T tile: per 4-KiB
    // sub-tile [2 f16 units | one MX step].  An MX step's two pieces are its lanes' first / second 16 bytes (per-lane source address).
    const unsigned off16 = lane * 16, off32 = lane * 32;
    const int piece_src = wave * 2048;
    auto uniform_ptr = [](const unsigned char* q) -> const unsigned char* {
        const unsigned long long v = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
    };
    auto dma_pair = [&](const unsigned char* src, unsigned char* dst, const bool mx_piece) __attribute__((always_inline)) {
        if (mx_piece) { glds16_sv(uniform_ptr(src), off32, dst); glds16_sv(uniform_ptr(src + 16), off32, dst + 1024); }
        else { glds16_sv(uniform_ptr(src), off16, dst); glds16_sv(uniform_ptr(src + 1024), off16, dst + 1024); }
    };
    auto dma_k = [&](int t) __attribute__((always_inline)) { dma_pair(Kg + (size_t)t * TILEB + piece_src, kring + (size_t)(t & 1) * TILEB + piece_src, wave >= 2); };
    auto dma_v = [&](int t) __attribute__((always_inline)) { dma_pair(Vg + (size_t)t * TILEB + piece_src, vring + (size_t)(t & 1) * TILEB + piece_src, (wave & 1) != 0); };
    // after barrier(t): every wave holds its K(t) fragments and has left V^T(t - 1)
    auto ring_advance = [&](int t) __attribute__((always_inline)) {
        if constexpr ((ABL & 16) != 0) return;
        if (t + 1 < nkt) dma_v(t + 1);
        if (t + 2 < nkt) dma_k(t + 2);
    };
    // tile info (host table, glc_mx2_build_tables): x = first table row of the p2c window, y = its 32-row blocks (1: x = delta_min & ~3; 2: x = the
    // aligned block z), z = lowest aligned c2p block, w = bit 0: the window also touches block z + 1; bits 8..: its last table row
    auto tinfo = [&](int kt) -> int4 { int4 t = a.tinfo[__builtin_amdgcn_readfirstlane(qt - kt + nt)]; if constexpr ((ABL & 32) != 0) t.y = 1; return t; };      // (ABL 32: every window one block — timing only)
    // packed gather indices of this lane for key tile kt: byte i = 4 * (delta(q - k_i) & 63)
    const unsigned idx_lane = (unsigned)(c - 8 * h + Sp) * 16u;
    auto load_idx = [&](int kt) -> i32x4 { return *reinterpret_cast<const i32x4*>(IDXg + ((ptrdiff_t)512 * (qt - kt) + (ptrdiff_t)idx_lane)); };

    MxFrag qf, kf;
    load_tile(Qg, qf);
    dma_k(0);
    dma_v(0);
    if (nkt > 1) dma_k(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // K(0) is in the ring for everyone (K(1) and V^T(0) are published by barrier(0))

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    float one_f = 1.0f;
    asm volatile("" : "+s"(one_f));      // opaque to the optimiser: fma(p, 1, -half) stays a v_fma_mix_f32

    // Shared tail of every key tile: key bias, online softmax (log2 units, deferred rescale), P*V with V^T from the ring (attention_mx.hip).
    auto softmax_pv = [&](float (&sv)[16], int kt) __attribute__((always_inline)) {
        const int k0 = kt * 32;
        const unsigned char* vtile = vring + (size_t)(kt & 1) * TILEB;
        f16x8 vf[2];
        i32x8 vx;
        auto load_v = [&](int d) __attribute__((always_inline)) {
            vf[0] = *reinterpret_cast<const f16x8*>(vtile + d * 4096 + lane * 16);
            vf[1] = *reinterpret_cast<const f16x8*>(vtile + d * 4096 + 1024 + lane * 16);
            vx = cat8(*reinterpret_cast<const i32x4*>(vtile + d * 4096 + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(vtile + d * 4096 + 3072 + lane * 16));
        };
        load_v(0);
        if (k0 + 32 > kfirst) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(kb + k0 + foff);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(kb + k0 + foff + 4);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff);
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sv[i] += b0[i]; sv[4 + i] += b1[i]; sv[8 + i] += b2[i]; sv[12 + i] += b3[i]; }
        }
        if constexpr (!(ABL & 4)) {
        float mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, sv[i]), sv[i + 1]);
        mx = fmaxf(mx, sv[15]);
        if (__builtin_amdgcn_ballot_w64(mx - m > RESCALE_THR) != 0ull) {     // deferred rescale (attention.hip)
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
        const f32x2 m2 = {m, m};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 d = (f32x2){sv[i], sv[i + 1]} - m2;
            sv[i] = __builtin_amdgcn_exp2f(d[0]); sv[i + 1] = __builtin_amdgcn_exp2f(d[1]);
            ps2 += (f32x2){sv[i], sv[i + 1]};
        }
        l += ps2[0] + ps2[1];
        } else l += sv[0];
        // P travels as (hi8 | lo8): f16(p) for the f16 MFMAs (k-step t = keys 16 t + 8 h + j), fp8 parts of the 16 keys for the scaled one
        f16x8 pf[2];
        i32x8 px;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[t][j] = (f16_t)sv[8 * t + j];
        }
        if constexpr ((ABL & 4) != 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) px[q] = __builtin_bit_cast(int, sv[q]);
        } else
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * q], sv[4 * q + 1], 0, false);
            wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * q + 2], sv[4 * q + 3], wh, true);
            px[q] = wh;
            // lo8 = e4m3((p - f16(p)) 2^SHIFT): one mixed-precision FMA per value on the halves of the packed f16 operand (attention_mx.hip)
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
#pragma unroll
        for (int t = 0; t < 2; ++t) o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t], pf[t], o0, 0, 0, 0);      // O^T[dd][query c]
        if constexpr (!(ABL & 8)) o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o0, 0, 0, 0, SC, 1, SC);
        __builtin_amdgcn_sched_barrier(0);
        load_v(1);
#pragma unroll
        for (int t = 0; t < 2; ++t) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[t], pf[t], o1, 0, 0, 0);
        if constexpr (!(ABL & 8)) o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o1, 0, 0, 0, SC, 1, SC);
    };

    // key-tile ranges of THIS wave's query tile: [0, kt_a) saturated at delta = P - 1, [kt_a, kt_b) generic, [kt_b, nkt) saturated at delta = 0
    int kt_a = q0m - 31 - a.rsat_pos >= 0 ? (q0m - 31 - a.rsat_pos) / 32 + 1 : 0;
    kt_a = kt_a > nkt ? nkt : kt_a;
    int kt_b = (q0m + 31 - a.rsat_neg + 31) / 32;
    kt_b = kt_b < kt_a ? kt_a : (kt_b > nkt ? nkt : kt_b);

    // Position-table operands in registers: pqw = the (first) 32 rows of the current tile's p2c window (the broadcast row on saturated
    // tiles); aux = its second 32 rows when the window has two blocks, and before that — from the middle of the previous tile to its end —
    // the PK block of the next c2p block.  A tile that needs both gets its second window block late (requested at the end of the
    // previous tile, used by the last MFMA group of this one): three operand sets in flight under the softmax do not fit 256 registers.
    MxFrag pqw, aux;

    // Saturated key tiles: delta is ONE value d*: c2p = Q_q.PK[d*] a per-query constant, p2c = K_k.PQ[d*] a second product on the same K tile
    // with every query column reading row d* (attention_mx.hip).  Every wave runs every key tile: one workgroup barrier per tile (the rings).
    auto sat_tiles = [&](int kt_lo, int kt_hi, int dstar) {
        if (kt_lo >= kt_hi) return;
        float cq;
        {
            MxFrag pkb;                         // broadcast fragments: every row / column is table row d*
            const int so = (dstar >> 5) * 8192 + (dstar & 31) * 32;        // split-form offsets of row d* in the Q layout / the K layout (slot pi)
            const int sk = (dstar >> 5) * 8192 + glc_pi32(dstar & 31) * 32;
            auto bc_rows = [&](const unsigned char* base, int off, MxFrag& f) __attribute__((always_inline)) {
                const unsigned vf = (unsigned)((off & ~8191) + ((off & 8191) >> 1) + h * 512);
                const unsigned vx = (unsigned)(off + 4096 + h * 1024);
#pragma unroll
                for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(base + (size_t)vf + s * 1024);
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
                    f.x[m2] = cat8(*reinterpret_cast<const i32x4*>(base + (size_t)vx + m2 * 2048), *reinterpret_cast<const i32x4*>(base + (size_t)vx + (m2 * 2048 + 16)));
            };
            bc_rows(PQg, so, pqw);
            bc_rows(PKg, sk, pkb);
            f32x16 t;
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = 0.f;
            mm_lh_hl(pkb, qf, t);                // every row = PK[d*] . Q_c
            cq = t[0];
        }
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            k_tile(kt, kf);
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = cq;
            mm_lh_hl(kf, qf, sacc);
            mm_lh_hl(kf, pqw, sacc);             // + K_k . PQ[d*] (same for every query column)
            float sv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = sacc[i];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my DMA pieces of V^T(kt) and K(kt + 1), requested a whole tile ago, have landed
            if constexpr (!(ABL & 16)) wg_barrier_lds();           // barrier(kt)
            ring_advance(kt);
            softmax_pv(sv, kt);
        }
    };

    sat_tiles(0, kt_a, a.P - 1);

    if (kt_a < kt_b) {
        auto byte_of = [](const i32x4& v, int i) -> unsigned { return ((unsigned)v[i >> 2] >> (8 * (i & 3))) & 255u; };
        // one aligned block B of the c2p table: T[q][32 (B & 1) + (row & 31)] = Q_q . PK[row]   (PK tile B; slot r holds row 32 B + pi(r))
        auto c2p_store = [&](const f32x16& v, int B) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned cb = (unsigned)(128 * (B & 1) + 64 * (g >> 1) + 32 * h + 16 * (g & 1));
                *reinterpret_cast<f32x4*>(Tw + 256 * c + (cb ^ swzc)) = (f32x4){v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            }
        };
        // one p2c block whose first table row is r0 (a multiple of 4): I[key slot c][(r0 + row) & 31] = K_c . PQ[r0 + row]
        auto p2c_store = [&](const f32x16& v, int r0) __attribute__((always_inline)) {
            const unsigned cb0 = (unsigned)(4 * r0 + 16 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned cb = (cb0 + 32 * g) & 127u;
                *reinterpret_cast<f32x4*>(Iw + 128 * c + (cb ^ swzc)) = (f32x4){v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            }
        };
        unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tiles = 0, tlast = 0;
        const unsigned long long clk0 = DIAG ? __builtin_amdgcn_s_memtime() : 0, rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0;
        auto stamp = [&](int k) __attribute__((always_inline)) {      // time since the previous stamp goes to segment k (k < 0: start)
            if constexpr (DIAG) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (k >= 0) seg[k] += t - tlast;
                tlast = t;
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // ---- entry: the c2p blocks, the p2c window rows and the indices of the first generic tile (synchronous: once per wave) ----
        int4 ti = tinfo(kt_a);
        f32x16 cacc;
        {
            load_tile(PKg + (size_t)ti.z * TILEB, aux);
#pragma unroll
            for (int i = 0; i < 16; ++i) cacc[i] = 0.f;
            mm_lh_hl(aux, qf, cacc);
            c2p_store(cacc, ti.z);
            if (ti.w & 1) {
                load_tile(PKg + (size_t)(ti.z + 1) * TILEB, aux);
#pragma unroll
                for (int i = 0; i < 16; ++i) cacc[i] = 0.f;
                mm_lh_hl(aux, qf, cacc);
                c2p_store(cacc, ti.z + 1);
            }
            wave_lds_sync();
        }
        load_rows_win(ti.x, ti.w >> 8, pqw);
        if (ti.y == 2) load_rows_win(ti.x + 32, ti.w >> 8, aux);
        i32x4 idx = load_idx(kt_a);
        bool pend = false;       // cacc holds the c2p block ti.z of the tile about to start (computed under the previous tile's tail)

        for (int kt = kt_a; kt < kt_b; ++kt) {
            const bool has_next = kt + 1 < kt_b;
            const int4 tn = tinfo(has_next ? kt + 1 : kt);
            const bool newblk = has_next && tn.z != ti.z;          // the next tile's window reaches one block further down the table
            stamp(-1);
            stamp(0);
            const bool pk_early = newblk && ti.y == 1;             // aux is free on a one-block tile: the PK block gets the whole tile to arrive
            if (pk_early && !(ABL & 1)) load_tile(PKg + (size_t)tn.z * TILEB, aux);
            __builtin_amdgcn_sched_barrier(0);
            k_tile(kt, kf);
            if (pend) { c2p_store(cacc, ti.z); wave_lds_sync(); }
            // ---- c2p: the initial accumulator of S^T ----
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = (ABL & 2) ? __builtin_bit_cast(float, idx[i & 3]) * 1e-30f : *reinterpret_cast<const float*>(smem_mx2 + (Tc ^ byte_of(idx, i)));
            // ---- p2c on this wave's window, S^T = K Q^T + c2p ----
            f32x16 bacc, bacc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
            mm_hl_lh(pqw, kf, bacc);
            mm_lh_hl(kf, qf, sacc);
            if (ti.y == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) bacc2[i] = 0.f;
                mm_hl_lh(aux, kf, bacc2);
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);                                              // seg 1: K fragments, c2p gather, p2c + S^T MFMA issue
            // my DMA pieces of V^T(kt) and K(kt + 1), requested a whole tile ago, have landed (only an early PK block may still be in flight)
            if (pk_early && !(ABL & 1)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (!(ABL & 16)) wg_barrier_lds();           // barrier(kt): every wave holds its K(kt) fragments and has left V^T(kt - 1)
            stamp(2);                                              // seg 2: request wait + the ring barrier
            ring_advance(kt);
            // ---- the next tile's rows into the registers this tile is done with ----
            if (has_next && !(ABL & 1)) load_rows_win(tn.x, tn.w >> 8, pqw);
            __builtin_amdgcn_sched_barrier(0);
            if (ABL & 1) { }
            else if (newblk && ti.y == 2) load_tile(PKg + (size_t)tn.z * TILEB, aux);
            else if (!newblk && has_next && tn.y == 2) load_rows_win(tn.x + 32, tn.w >> 8, aux);
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);                                              // seg 3: DMA and row requests
            // ---- image: store, gather (index bytes: 4 (delta & 31) ^ 16 e), add ----
            i32x4 idxi;
#pragma unroll
            for (int w = 0; w < 4; ++w) idxi[w] = (idx[w] & 0x7C7C7C7C) ^ 0x30201000;
            float sv[16];
            if constexpr ((ABL & 2) != 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sv[i] = sacc[i] + bacc[i] + (ti.y == 2 ? bacc2[i] : 0.f);
            } else {
            p2c_store(bacc, ti.x);
            wave_lds_sync();
            if (ti.y == 2) {
                // two aligned blocks: the first pass holds block z, the second block z + 1; an element belongs to the pass of its delta's block parity
                float g0[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) g0[i] = *reinterpret_cast<const float*>(smem_mx2 + (Ih ^ byte_of(idxi, i)) + (128 * (i & 3) + 1024 * (i >> 2)));
                wave_lds_sync();
                p2c_store(bacc2, ti.x + 32);
                wave_lds_sync();
                const unsigned flip = (ti.z & 1) ? 0u : 0x80808080u;          // bit 7 of an index byte = (delta >> 5) & 1; after the flip: set = first pass
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float g1 = *reinterpret_cast<const float*>(smem_mx2 + (Ih ^ byte_of(idxi, i)) + (128 * (i & 3) + 1024 * (i >> 2)));
                    const bool first = (((unsigned)idx[i >> 2] ^ flip) >> (8 * (i & 3)) & 128u) != 0;
                    sv[i] = sacc[i] + (first ? g0[i] : g1);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    sv[i] = sacc[i] + *reinterpret_cast<const float*>(smem_mx2 + (Ih ^ byte_of(idxi, i)) + (128 * (i & 3) + 1024 * (i >> 2)));
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) idx = load_idx(kt + 1);
            stamp(4);                                              // seg 4: image stores (wait for the p2c MFMAs), gathers (wait for S^T)
            softmax_pv(sv, kt);
            stamp(5);                                              // seg 5: softmax, P.V issue
            pend = newblk;
            if (newblk) {
#pragma unroll
                for (int i = 0; i < 16; ++i) cacc[i] = 0.f;
                mm_lh_hl(aux, qf, cacc);                           // c2p block of the next tile: queued behind P.V, stored at the top of that tile
                __builtin_amdgcn_sched_barrier(0);
                if (tn.y == 2 && !(ABL & 1)) load_rows_win(tn.x + 32, tn.w >> 8, aux);      // the late second window block, in place of the PK block
            }
            stamp(6);                                              // seg 6: c2p issue (waits for the PK block)
            ti = tn;
            if constexpr (DIAG) ++tiles;
        }
        if constexpr (DIAG) {
            if (a.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {     // 64 workgroups of XCD 0
                unsigned long long* o = a.stamps + ((size_t)(blockIdx.x >> 3) * 8 + wave) * 10;
                for (int k = 0; k < 8; ++k) o[k] = seg[k];
                const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk0, dr = __builtin_amdgcn_s_memrealtime() - rt0;
                o[8] = dr ? dc * 1000 / dr : 0; o[9] = tiles;
            }
        }
    }

    sat_tiles(kt_b, nkt, 0);

    if (!active) return;
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
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

// Host side: the two tables the kernel reads, from the distance -> delta table of this padded length (engine.hip, cached per Sp).
//   idx16 [2 Sp + 32][16 bytes]: entry u + Sp, byte i = 4 * (delta(u - kc_i) & 63), kc_i = (i & 7) + 16 (i >> 3) — the 16 keys a lane of the
//         32x32 S^T accumulator holds (lane (c, h) of query tile qt against key tile kt: u = 32 (qt - kt) + c - 8 h)
//   tinfo [2 nt + 1] int4: entry (qt - kt) + nt: y = 32-row blocks of the p2c window (1 / 2), x = its first table row (y = 1: delta_min & ~3;
//         y = 2: 32 z, the two aligned blocks), z = lowest aligned c2p block (delta_min >> 5), w = bit 0: block z + 1 is touched too, bits 8..: delta_max (the window's last row)
// Returns false when the table does not have the properties the kernel builds on (non-decreasing, steps of at most one, a tile pair's
// window inside two aligned blocks, the lowest block dropping by at most one per key tile): the caller then keeps attention_mx.hip's
// band kernel.
bool glc_mx2_build_tables(int Sp, int P, const int32_t* dtab, std::vector<unsigned char>& idx16, std::vector<int4>& tinfo) {
    if (Sp <= 0 || Sp % 64 || P <= 0 || P % 32 || !dtab) return false;
    const int nt = Sp >> 5, n = 2 * Sp - 1;
    for (int i = 0; i < n; ++i) {
        if (dtab[i] < 0 || dtab[i] >= P) return false;
        if (i && (dtab[i] < dtab[i - 1] || dtab[i] > dtab[i - 1] + 1)) return false;
    }
    auto d = [&](int r) { r = r < -(Sp - 1) ? -(Sp - 1) : (r > Sp - 1 ? Sp - 1 : r); return dtab[r + Sp - 1]; };
    tinfo.assign(2 * nt + 1, int4{0, 1, 0, 0});
    int prev_z = -1;
    for (int dq = nt - 1; dq >= -(nt - 1); --dq) {          // the order a wave walks its key tiles in
        const int dmin = d(32 * dq - 31), dmax = d(32 * dq + 31);
        const int z = dmin >> 5;
        const bool one = dmax - (dmin & ~3) < 32;                 // the window fits one block of 32 rows from delta_min & ~3 on
        if ((dmax >> 5) > z + 1) return false;
        if (prev_z >= 0 && (z > prev_z || z < prev_z - 1)) return false;
        prev_z = z;
        tinfo[dq + nt] = int4{one ? (dmin & ~3) : 32 * z, one ? 1 : 2, z, ((dmax >> 5) > z ? 1 : 0) | (dmax << 8)};
    }
    idx16.assign((size_t)(2 * Sp + 32) * 16, 0);
    for (int j = 0; j < 2 * Sp + 32; ++j)
        for (int i = 0; i < 16; ++i) {
            const int kc = (i & 7) + 16 * (i >> 3);
            idx16[(size_t)j * 16 + i] = (unsigned char)(4 * (d(j - Sp - kc) & 63));
        }
    return true;
}

// Same contract as glc_launch_attention_mx, plus the two tables of glc_mx2_build_tables in AttnArgs::idx16 / tinfo.
const char* glc_launch_attention_mx2(hipStream_t st, const AttnArgs& a_in) {
    AttnArgs a = a_in;
    if (!a.gx_sat) a.gx_sat = glc_gx_sat_ptr();              // fp8 range guard of the GX context rows
    if (!a.act_sc) a.act_sc = glc_gx_act_sc();               // ... and the exponent of the activation rows (engine.hip act_sc)
    if (!a.Qh || !a.Kh || !a.Vt || !a.PK || !a.PQ || !a.kbias || !a.klen || !a.kfirst || !a.CTX || !a.idx16 || !a.tinfo) return "attention(mx2): null pointer";
    if (a.B <= 0 || a.nh <= 0 || a.Sp <= 0 || a.Sp % 64 || a.H != a.nh * 64 || a.P <= 0 || a.P % 32) return "attention(mx2): bad shape";
    if (a.sel_b || a.tile_flag) return "attention(mx2): no row selection in this kernel";
    static_assert(2 * LDS_BYTES <= 160 * 1024, "LDS budget: two workgroups per CU");
    const int nqb = (a.Sp + 32 * NW - 1) / (32 * NW), bh8 = (a.B * a.nh + 7) / 8 * 8;
    auto go = [&](auto kern, std::atomic<unsigned>& r) -> const char* {
        if (!glc_raise_lds_limit(kern, (int)LDS_BYTES, r)) return "attention(mx2): cannot raise the dynamic LDS limit";
        hipLaunchKernelGGL(kern, dim3(nqb * bh8), dim3(64 * NW), LDS_BYTES, st, a);
        return nullptr;
    };
    static std::atomic<unsigned> r0{0}, r1{0}, r2{0}, r3{0}, r4{0};
    if (a.stamps) return go(attn_mx2_kernel<true>, r1);
    static std::atomic<unsigned> r5{0}, r6{0}, r7{0}, r8{0};
    const int abl = (a.variant >> 8) & 31;       // (glc_debug_attn_bench passes bits 8-12)
    if (abl == 1) return go(attn_mx2_kernel<false, 1>, r2);
    if (abl == 2) return go(attn_mx2_kernel<false, 2>, r3);
    if (abl == 3) return go(attn_mx2_kernel<false, 3>, r4);
    if (abl == 4) return go(attn_mx2_kernel<false, 7>, r5);
    if (abl == 8) return go(attn_mx2_kernel<false, 11>, r6);
    if (abl == 16) return go(attn_mx2_kernel<false, 19>, r7);
    if (abl == 31) return go(attn_mx2_kernel<false, 31>, r8);
    static std::atomic<unsigned> r9{0};
    if (abl == 30) return go(attn_mx2_kernel<false, 32>, r9);
    return go(attn_mx2_kernel<false>, r0);
}
