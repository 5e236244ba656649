// DeBERTa-v2/v3 disentangled self-attention, WORKGROUP-SHARED band kernel (round 2).  Same algebra, operand layouts and per-element
// arithmetic as attn_band_kernel (attention.hip, whose header derives score = Q.K + Q.PK[delta(q-k)] + K.PQ[delta(q-k)] and the
// Toeplitz-band form); what changes is who loads and who computes what:
//
//   * A workgroup = NW waves = NW consecutive 32-query tiles of ONE (batch, head).  Every key tile (K fragments 4 units, V^T 4 units)
//     is fetched ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, 2 per wave and tile) into a
//     3-slot LDS ring, one tile ahead, and read by all waves as ds_read_b128 fragments — the band kernel had every wave load
//     its own copy through the L1 path (20 global_load_dwordx4 per wave and tile).
//   * The p2c band is shared.  Wave w needs the relative-distance blocks b = w - t ("low") and b + 1 ("high") against key tile t;
//     the high block of wave w IS the low block of wave w + 1.  So every wave computes only its low block (4 MFMAs instead of 8)
//     into a workgroup-shared LDS image [32 keys][32 (NW + 1) distances], and the one block nobody owns (the high block of the last
//     wave) is computed by wave t mod NW: NW + 1 blocks per key tile instead of 2 NW  (17 instead of 20 MFMAs per wave and tile
//     at NW = 4, 16.5 at NW = 8; 12 are algorithmic).
//   * The c2p term enters the S^T MFMA chain as its INITIAL ACCUMULATOR (gathered from the wave's ring at the top of the tile), so
//     the score assembly is one add per element instead of two and the accumulator needs no zeroing.
//   * Saturated key tiles (every q - k beyond the bucket clamp for ALL waves of the workgroup) run as before, per wave, on the shared
//     K / V^T tiles.
// Synchronisation per band tile: barrier X (every wave has finished gathering the previous tile's p2c image) -> p2c stores ->
// barrier Y (image complete; every wave's DMA pieces of tile t + 1 have landed: each wave waited for its own at the TOP of the tile,
// half a tile after requesting them, see "Vector-memory pipeline" at the band loop) -> gathers.  Saturated
// tiles: one barrier.  The DMA runs up to two tiles ahead: tile t + 2 is requested right after barrier Y of tile t (top of a
// saturated tile t) into slot (t + 2) % 3 = (t - 1) % 3, whose last readers are the P.V products of tile t - 1, which every wave has
// retired (lgkmcnt(0)) before a barrier that the requesting wave has passed; its pieces are waited for (vmcnt(0)) before a barrier
// of tile t + 1, i.e. a whole tile after the request, and first read at the top of tile t + 2.
// NW = 4 (16-bit operands; 80 KiB LDS: two workgroups per CU, i.e. two independent waves per SIMD) or NW = 8 (split-f16 operands
// of the fp32 mode, whose units are twice as large: 153 KiB, one workgroup of 8 waves per CU).
#include <stdio.h>
#include <stdlib.h>
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

constexpr float RESCALE_THR = 8.0f;   // log2 units (as attention.hip)
constexpr int LROW = 68;              // floats per c2p ring row (2 blocks of 32 + 4 pad)

template <bool SPLIT, typename T> struct WgFrag { typedef typename AFrag<T>::type type; };
template <typename T> struct WgFrag<true, T> { typedef f16x8s type; };

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}
// The same request with the address as (wave-uniform SGPR base) + (32-bit per-lane offset register).  Written as asm because the
// builtin's address lands in a temporary VGPR pair inside loops (the zero-extension of the lane offset is hoisted out of the block and
// the saddr form no longer matches), and the compiler guards every later write to an LDS-DMA's address registers with a vmcnt(0): here
// the lane offset register is loop-invariant and never rewritten.  The compiler does not count these requests: its own vmcnt waits can
// only come out stricter than needed, and every reader of the landed bytes sits behind an explicit vmcnt wait + workgroup barrier.
__device__ __forceinline__ void glds16_sv(const unsigned char* ubase, unsigned lane_off, void* l) {
    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)l;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la), "v"(lane_off), "s"(ubase) : "memory");
}
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wg_barrier_lds() {          // all my LDS traffic retired, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ void wg_barrier_all() {          // ... and all my vector-memory traffic (LDS-DMA pieces included)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];

// KVG = K / V^T fragments straight from global memory, per wave (no ring); the LDS it frees double-buffers the p2c image, which
// removes barrier X: ONE workgroup barrier per band tile, none in saturated tiles.
// STAG (NW = 8): the two halves of the workgroup (waves 0-3 / 4-7, SIMD partners pairwise) each share their OWN p2c image (5 blocks per
// 4 waves) and run the band loop HALF A TILE apart: the tile body is two phases, P1 = [S^T, p2c MFMAs, image stores] and P2 = [gathers,
// softmax, P.V, next c2p block], with one workgroup barrier after each; the late half enters the loop one barrier later, so on every SIMD
// one wave is in the matrix-heavy phase while its partner is in the VALU / LDS-heavy one (the staggered-wave-group idea of gemm256s.hip).
// VGL (split units, NW = 4): only K goes through the ring (8 KiB per tile = two DMA pieces per wave); V^T fragments are loaded per wave
// from global memory.  80 KiB of LDS: TWO workgroups of four waves per CU, i.e. the two waves of a SIMD belong to different workgroups.
// DIAG: s_memtime stamps at the phase boundaries of a band tile, summed per wave in SGPRs (glc_debug_attn_bench prints them).
// PD (split units): precision-budget build — AttnArgs::prec rounds operand tensors to f16 at run time by zeroing their lo halves
// (bits: 1 Q, 2 K, 4 V^T, 8 P, 16 PQ rows, 32 PK rows); numerically the kernel that never fetches / forms them, at unchanged cost.
// NMM (split units): MFMAs per split product — 3; 2 = TIMING-ONLY build without the a_lo * b_hi MFMA (wrong results: what a cheaper
// cross-term form could gain at most, scripts/attn_bench.py variant bit 6).
template <typename T, bool SPLIT, int NW, bool KVG, bool STAG = false, bool VGL = false, bool DIAG = false, bool PD = false, int NMM = 3>
__global__ __launch_bounds__(64 * NW, 2) void attn_wg_kernel(AttnArgs a) {
    static_assert(!PD || SPLIT, "precision switches act on split units");
    static_assert(!STAG || (NW == 8 && !KVG), "the stagger pairs the two 4-wave halves of an 8-wave workgroup");
    static_assert(!VGL || (SPLIT && NW == 4 && !KVG && !STAG), "V^T from global: the 4-wave split-unit variant");
    static_assert(!SPLIT || sizeof(T) == 4, "split operands live in the fp32 layouts");
    static_assert(SPLIT || sizeof(T) == 2, "16-bit operands or split-f16 units");
    typedef typename WgFrag<SPLIT, T>::type frag_t;
    auto mm = [](const frag_t& x, const frag_t& y, f32x16& acc) __attribute__((always_inline)) {
        if constexpr (SPLIT && NMM == 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.hi, y.lo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.hi, y.hi, acc, 0, 0, 0);
        } else mma32(x, y, acc);
    };
    constexpr int UNITB = 512 * (int)sizeof(T);      // bytes of one fragment unit (64 lanes x 8 elements)
    constexpr int TILEB = 4 * UNITB;                 // one K tile, or one V^T tile
    constexpr int NPIECE = (VGL ? 1 : 2) * TILEB / 1024;     // 1-KiB DMA pieces per key tile (K then V^T, or K only): 2 per wave
    constexpr int SLOTB = (VGL ? 1 : 2) * TILEB;             // bytes of one ring slot
    static_assert(KVG || NPIECE == 2 * NW, "two DMA pieces per wave and tile");
    constexpr int NSH = STAG ? 4 : NW;               // waves that share one p2c image
    constexpr int NIMG = (KVG || STAG) ? 2 : 1;      // images: KVG alternates two, STAG keeps one per half
    constexpr int LROWP = 32 * (NSH + 1) + 4;        // floats per p2c image row
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int Sp = a.Sp;

    float* c2p_l = reinterpret_cast<float*>(smem_wg) + (size_t)wave * 32 * LROW;                     // this wave's ring [32 q][64 + 4]
    float* p2c_img = reinterpret_cast<float*>(smem_wg) + (size_t)NW * 32 * LROW;                     // shared [32 keys][LROWP]
    unsigned char* kv_ring = smem_wg + ((size_t)NW * 32 * LROW + NIMG * 32 * LROWP) * sizeof(float);  // 3 x (K tile | V^T tile)
    const int grp = STAG ? wave >> 2 : 0, wl = STAG ? wave & 3 : wave;      // image-sharing group and this wave's slot in it

    // XCD-aware decode of the 1-D grid: every query block of one (batch, head) gets the same id % 8 (shared L2 for its K / V^T)
    const int nqb = (Sp + 32 * NW - 1) / (32 * NW);
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int bh = xcd + 8 * (jj / nqb);
    const int Q0 = (jj % nqb) * 32 * NW;
    const int QX = Q0 + 32 * (NSH * grp + NSH);           // the query tile whose LOW block is the group's unowned (last high) block
    if (bh >= a.B * a.nh) return;                       // workgroup-uniform
    const int b = bh / a.nh, hh = bh - b * a.nh;
    const int q0 = Q0 + 32 * wave;                      // this wave's query tile (may lie past Sp in the last block: computes, never stores)
    const bool active = q0 < Sp;
    const int q0m = active ? q0 : Sp - 32;              // memory-safe tile for the Q fragments of an inactive wave
    if (a.tile_flag) {                                  // pruned last layer: skip workgroups without a selected query tile
        bool any = false;
        for (int w = 0; w < NW; ++w) { const int qw = Q0 + 32 * w; any = any || (qw < Sp && a.tile_flag[(size_t)b * (Sp >> 5) + (qw >> 5)]); }
        if (!any) return;
    }
    const int klen = a.klen[b];
    typedef T OutT;                                     // CTX rows carry the operand type (fp32 in the split mode)
    OutT* outp = reinterpret_cast<OutT*>(a.CTX) + ((size_t)b * Sp + q0m + c) * a.H + hh * 64;
    if (Q0 >= klen && Q0 > 0) {
        // every query of this block lies past the row's last attended token (padding of a ragged batch): never read by an
        // attended row; store zeros (finite) and leave
        if (active) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { store4<OutT>(outp + 8 * g + 4 * h, 0.f, 0.f, 0.f, 0.f); store4<OutT>(outp + 32 + 8 * g + 4 * h, 0.f, 0.f, 0.f, 0.f); }
        }
        return;
    }

    const unsigned char* __restrict__ Qg = reinterpret_cast<const unsigned char*>(a.Qh) + ((size_t)bh * Sp + q0m) * 64 * sizeof(T) + lane * 8 * sizeof(T);
    const unsigned char* __restrict__ Kg = reinterpret_cast<const unsigned char*>(a.Kh) + (size_t)bh * Sp * 64 * sizeof(T);
    const unsigned char* __restrict__ Vg = reinterpret_cast<const unsigned char*>(a.Vt) + (size_t)bh * Sp * 64 * sizeof(T);
    const unsigned char* __restrict__ PKg = reinterpret_cast<const unsigned char*>(a.PK) + ((size_t)hh * a.P * 64 + 32 * 8 * h) * sizeof(T);
    const unsigned char* __restrict__ PQg = reinterpret_cast<const unsigned char*>(a.PQ) + ((size_t)hh * a.P * 64 + 32 * 8 * h) * sizeof(T);
    const float* __restrict__ kb = a.kbias + (size_t)b * Sp;

    int nkt = (klen + 31) >> 5;                                    // key tiles beyond the last valid key add exactly 0
    nkt = nkt < 1 ? 1 : (nkt > (Sp >> 5) ? (Sp >> 5) : nkt);
    const int kfirst = a.kfirst[b];
    const int foff = 8 * h;

    // Position rows: otab entry (q - k) + Sp - 1 + 64 holds the byte offsets of row delta(q - k) in the PQ (x) / PK (y) layouts; 64
    // clamped entries pad each end.  Row c of the LOW block of (query tile at qb, key tile t): rel = qb - 32 t - 31 + c.
    const int otab_max = 2 * Sp - 2 + 128;
    auto block_delta = [&](int qb, int t) -> int2 {
        int idx = qb - 32 * t - 31 + c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return a.otab[idx];
    };
    auto block_x = [&](int qb, int t) -> int {                    // the PQ-layout offset only (one register to carry a tile ahead)
        int idx = qb - 32 * t - 31 + c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return reinterpret_cast<const int*>(a.otab)[2 * idx];
    };
    auto pk_of_pq = [&](int x) -> int {                           // same table row in the PK layout: its slot within the 32-row unit is pi32-permuted
        constexpr int SH = sizeof(T) == 4 ? 5 : 4;               // log2(bytes of one 8-element row entry)
        const int r = (x >> SH) & 31;
        return x + ((glc_pi32(r) - r) << SH);
    };
    auto drop_lo = [&](frag_t& f, int bit) __attribute__((always_inline)) {
        if constexpr (PD) { if (a.prec & bit) f.lo = (f16x8)(f16_t)0; }
    };
    auto load_rows = [&](const unsigned char* base, int off, frag_t (&f)[4]) {       // 4 fragment units of gathered table rows
#pragma unroll
        for (int s = 0; s < 4; ++s) { f[s] = *reinterpret_cast<const frag_t*>(base + off + s * UNITB); drop_lo(f[s], base == PQg ? 16 : 32); }
    };
    // One fragment unit out of the ring.  16-bit operands: the unit is the linear copy of its 1 KiB in HBM (16 B per lane).  Split units
    // (32 B per lane in HBM: [8 hi | 8 lo]) are re-arranged by the DMA into [64 lanes x hi | 64 lanes x lo]: read lane-strided at 32 B
    // the two ds_read_b128 of a fragment were 2-way bank conflicts (SQ_LDS_BANK_CONFLICT = 26 % of the kernel's LDS cycles).
    auto ring_unit = [&](const unsigned char* unit) -> frag_t {
        if constexpr (SPLIT) {
            frag_t f;
            f.hi = *reinterpret_cast<const f16x8*>(unit + lane * 16);
            f.lo = *reinterpret_cast<const f16x8*>(unit + 1024 + lane * 16);
            return f;
        } else return *reinterpret_cast<const frag_t*>(unit + lane * 16);
    };
    auto k_tile = [&](int t, frag_t (&f)[4]) {                  // the 4 fragment units of key tile t: from the ring, or (KVG) from global
        if constexpr (KVG) {
#pragma unroll
            for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const frag_t*>(Kg + (size_t)t * TILEB + s * UNITB + lane * (UNITB / 64));
        } else {
            const unsigned char* tile = kv_ring + (size_t)(t % 3) * SLOTB;
#pragma unroll
            for (int s = 0; s < 4; ++s) f[s] = ring_unit(tile + s * UNITB);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) drop_lo(f[s], 2);
    };
    auto band_store = [&](float* dst, const f32x16& v) {           // 4 consecutive rr per register group
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(dst + 8 * g + 4 * h) = (f32x4){v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    };
    // LDS-DMA of key tile t into ring slot t % 3: piece p = 2 wave + i is 1 KiB of [K tile | V^T tile]
    const unsigned dma_lane_off = lane * (SPLIT ? 32 : 16);
    auto uniform_ptr = [](const unsigned char* q) -> const unsigned char* {       // tell the compiler what it cannot prove: wave-uniform, lives in SGPRs
        const unsigned long long v = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
    };
    auto dma_tile = [&](int t) {
        if constexpr (KVG) return;
        unsigned char* slot = kv_ring + (size_t)(t % 3) * SLOTB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = 2 * wave + i;
            if constexpr (SPLIT) {
                // piece p = (unit `wave` of [K | V^T], part i): every lane fetches the hi (i = 0) or lo (i = 1) 16 bytes of ITS 32-byte
                // entry, so the unit lands as [64 x hi | 64 x lo] (the per-lane source address makes the DMA a gather).
                // Address = wave-uniform base + the loop-invariant lane offset register: a per-tile address register would be dead
                // right after the request, and the compiler guards the re-use of an LDS-DMA's address registers with a vmcnt(0).
                const unsigned char* ubase = (VGL || wave < 4) ? Kg + (size_t)t * TILEB + wave * UNITB : Vg + (size_t)t * TILEB + (wave - 4) * UNITB;
                glds16_sv(uniform_ptr(ubase + i * 16), dma_lane_off, slot + p * 1024);
            } else {
                glds16_sv(uniform_ptr(p < NPIECE / 2 ? Kg + (size_t)t * TILEB + p * 1024 : Vg + (size_t)t * TILEB + (p - NPIECE / 2) * 1024), dma_lane_off, slot + p * 1024);
            }
        }
    };

    frag_t qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { qf[s] = *reinterpret_cast<const frag_t*>(Qg + s * UNITB); drop_lo(qf[s], 1); }
    dma_tile(0);
    if (nkt > 1) dma_tile(1);
    frag_t kf[4];                           // KVG: always holds K(kt) at the top of tile kt (re-loaded in place after its last MFMA)
    if constexpr (KVG) k_tile(0, kf);

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    const int rr_base = c - 8 * h + 31;

    // Shared tail of every key tile: key bias, online softmax (log2 units, deferred rescale), P*V with V^T from the ring.
    auto softmax_pv = [&](float (&sv)[16], int kt) {
        const int k0 = kt * 32;
        frag_t vt[2][2];
        if constexpr (KVG || VGL) {
            const unsigned char* vtile = Vg + (size_t)kt * TILEB + lane * (UNITB / 64);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                vt[0][t] = *reinterpret_cast<const frag_t*>(vtile + t * UNITB);
                vt[1][t] = *reinterpret_cast<const frag_t*>(vtile + (2 + t) * UNITB);
            }
        } else {
            const unsigned char* vtile = kv_ring + (size_t)(kt % 3) * SLOTB + TILEB;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                vt[0][t] = ring_unit(vtile + t * UNITB);
                vt[1][t] = ring_unit(vtile + (2 + t) * UNITB);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) { drop_lo(vt[0][t], 4); drop_lo(vt[1][t], 4); }
        if (k0 + 32 > kfirst) {                                             // wave-uniform: tile holds masked keys
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
        if (__builtin_amdgcn_ballot_w64(mx - m > RESCALE_THR) != 0ull) {     // deferred rescale (attention.hip)
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
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
                for (int j = 0; j < 8; ++j) {
                    const f16_t ph = (f16_t)sv[8 * t + j];
                    pfr.hi[j] = ph;
                    pfr.lo[j] = (f16_t)(sv[8 * t + j] - (float)ph);
                }
                drop_lo(pfr, 8);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) pfr[j] = (T)sv[8 * t + j];
            }
            mm(vt[0][t], pfr, o0);     // O^T[dd][query c], dd = (i&3) + 8*(i>>2) + 4h
            mm(vt[1][t], pfr, o1);     //                   dd + 32
        }
    };

    // key-tile ranges, WORKGROUP-uniform: [0, kt_a) saturated high for every wave (wave 0 has the smallest q - k), [kt_a, kt_b)
    // band path (correct for any tile), [kt_b, nkt) saturated low for every wave (the last wave has the largest q - k)
    int kt_a = Q0 - 31 - a.rsat_pos >= 0 ? (Q0 - 31 - a.rsat_pos) / 32 + 1 : 0;
    kt_a = kt_a > nkt ? nkt : kt_a;
    int kt_b = (Q0 + 32 * (NW - 1) + 31 - a.rsat_neg + 31) / 32;
    kt_b = kt_b < kt_a ? kt_a : (kt_b > nkt ? nkt : kt_b);

    // Saturated key tiles: delta is ONE value d*: c2p = Q_q.PK[d*] is a per-query constant, p2c = K_k.PQ[d*] a second product on the
    // same K fragments (8 + 4 MFMA, no band).  One workgroup barrier per tile (ring hand-over).
    auto sat_tiles = [&](int kt_lo, int kt_hi, int dstar) {
        if (kt_lo >= kt_hi) return;
        frag_t pqb[4], pkb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // broadcast fragments: every row / column is table row d*
            pqb[s] = *reinterpret_cast<const frag_t*>(PQg + ((size_t)(dstar >> 5) * 2048 + (dstar & 31) * 8) * sizeof(T) + s * UNITB);
            pkb[s] = *reinterpret_cast<const frag_t*>(PKg + ((size_t)(dstar >> 5) * 2048 + glc_pi32(dstar & 31) * 8) * sizeof(T) + s * UNITB);
            drop_lo(pqb[s], 16); drop_lo(pkb[s], 32);
        }
        float cq;
        {
            f32x16 t;
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mm(pkb[s], qf[s], t);            // every row = PK[d*] . Q_c
            cq = t[0];
        }
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            if (kt + 2 < nkt) dma_tile(kt + 2);          // slot (kt - 1) % 3: its last readers retired before the barrier that ended tile kt - 1
            if constexpr (!KVG) k_tile(kt, kf);
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = cq;
#pragma unroll
            for (int s = 0; s < 4; ++s) mm(kf[s], qf[s], sacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) mm(kf[s], pqb[s], sacc);        // + K_k . PQ[d*] (same for every query column)
            if constexpr (KVG) k_tile(kt + 1 < nkt ? kt + 1 : kt, kf);
            float sv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = sacc[i];
            softmax_pv(sv, kt);
            if constexpr (!KVG) wg_barrier_all();        // tiles kt + 1 (and kt + 2) are in the ring for everyone
        }
    };

    if constexpr (!KVG) wg_barrier_all();  // tile 0 is in the ring
    sat_tiles(0, kt_a, a.P - 1);

    if (kt_a < kt_b) {
        // ---- band prologue: this wave's c2p blocks L(kt_a - 1) (high block of the first band tile) and L(kt_a) (its low block) ----
        {
            frag_t pk[4];
            f32x16 bacc;
            load_rows(PKg, block_delta(q0, kt_a - 1).y, pk);
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mm(pk[s], qf[s], bacc);
            band_store(c2p_l + c * LROW + 32, bacc);            // ring half 1
            load_rows(PKg, block_delta(q0, kt_a).y, pk);
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mm(pk[s], qf[s], bacc);
            band_store(c2p_l + c * LROW, bacc);                 // ring half 0
            wave_lds_sync();
        }
        // Gather addresses of the c2p band.  xr = 32 (step parity): the ring half that holds this tile's LOW block; the column of band
        // row rr is rr ^ xr.  On even steps that is lane base + immediate; on odd steps the xor depends on the lane, so the 16
        // addresses are computed once here instead of 3 VALU per element and tile.
        const float* c2p_even = c2p_l + c * LROW + rr_base;
        const float* c2p_odd[SPLIT ? 1 : 16];              // (split operands: no registers to spare, the xor is recomputed)
        if constexpr (!SPLIT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c2p_odd[i] = c2p_l + c * LROW + ((rr_base - (16 * (i >> 3) + (i & 7))) ^ 32);
        }
        // One band tile.  xr is a literal at both call sites (the loop is unrolled by two).  `pq` holds the rows of this wave's LOW
        // block for tile kt, requested one tile ago: it is re-loaded IN PLACE for tile kt + 1 as soon as its last MFMA has issued.
        // Vector-memory pipeline of the loop (every request has a whole phase to land; no wait sits right behind its own request):
        //   top of tile kt   wait for everything requested during tile kt - 1: the PQ rows of this tile, the offset-table entries of
        //                    tile kt + 1 and this wave's DMA pieces of key tile kt + 1 (requested after barrier Y, ~half a tile ago);
        //                    request the offset-table entries of tile kt + 2
        //   after S^T        request the PK rows of L(kt + 1) (used after barrier Y and the gather; they take the registers K(kt) has
        //                    just left) and the PQ rows of tile kt + 1 in place;  barrier Y is an LDS-only barrier (the DMA was waited for above:
        //                    every wave is past its own wait when it arrives), then the DMA of tile kt + 2 goes out
        // Before: the entries were fetched where they were used (a dependent-load stall per tile), the rows right before barrier Y,
        // whose vmcnt(0) then waited for them, and the DMA was waited for right after its request by the PK rows' own wait.
        unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, tiles = 0, tlast = 0;
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
        frag_t pq[4], pqx[4];
        load_rows(PQg, block_delta(q0, kt_a).x, pq);
        if ((kt_a % NSH) == wl) load_rows(PQg, block_delta(QX, kt_a).x, pqx);
        int od_n = block_x(q0, kt_a + 1);                       // PQ-layout row offsets of the NEXT tile's low block
        int odx_n = block_x(QX, kt_a + 1);                      // ... and of the unowned block (one wave per tile needs it: that wave would otherwise
                                                                // sit in a dependent load right before barrier X, every tile, with all others waiting)
        auto band_tile = [&](const int kt, const int xr) __attribute__((always_inline)) {
            const bool extra = (kt % NSH) == wl;                // wave-uniform: this wave also computes the block nobody owns
            stamp(-1);
            if constexpr (!KVG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(0);                                           // seg 0: wait for last tile's requests
            frag_t pk[4];
            const int od = od_n, odx = odx_n;
            od_n = block_x(q0, kt + 2);
            odx_n = block_x(QX, kt + 2);
            if constexpr (!KVG) k_tile(kt, kf);
            float* img = p2c_img + (KVG ? (size_t)(kt & 1) * 32 * LROWP : (size_t)grp * 32 * LROWP);      // KVG: two images, alternating; STAG: one per half
            // the gathered c2p band is the initial accumulator of S^T; reg i <-> key k0 + 16 (i>>3) + 8h + (i&7)
            f32x16 sacc;
            int rbo = rr_base;            // (round 4, as attention_mx.hip: the odd step's 16 gather addresses recomputed from an opaque copy of the
            if constexpr (SPLIT) asm volatile("" : "+v"(rbo));      //  base instead of living as spilled loop invariants; bit-identical)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kc = 16 * (i >> 3) + (i & 7);
                if constexpr (SPLIT) sacc[i] = xr ? c2p_l[c * LROW + ((rbo - kc) ^ 32)] : c2p_even[-kc];
                else sacc[i] = xr ? *c2p_odd[i] : c2p_even[-kc];
            }
            // ---- p2c (needs only K and PQ): low block of this wave, and, one wave per tile, the high block of the last wave ----
            f32x16 bacc, bacc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mm(pq[s], kf[s], bacc);
            if (extra) {
#pragma unroll
                for (int i = 0; i < 16; ++i) bacc2[i] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) mm(pqx[s], kf[s], bacc2);
            }
            // ---- S^T = K Q^T + c2p ----
#pragma unroll
            for (int s = 0; s < 4; ++s) mm(kf[s], qf[s], sacc);
            // requests for the next tile, into the registers whose last MFMA has just issued
            // (request order pinned — conditional rows, PK rows, PQ rows, later the DMA: the PK rows are consumed first, and the wait the
            // compiler places there counts the younger requests; a conditional request between them made that wait a vmcnt(0))
            __builtin_amdgcn_sched_barrier(0);
            if (((kt + 1) % NSH) == wl) load_rows(PQg, odx, pqx);
            __builtin_amdgcn_sched_barrier(0);
            load_rows(PKg, pk_of_pq(od), pk);                   // PK rows of L(kt + 1), into the registers K(kt) has just left: consumed after the gather
            __builtin_amdgcn_sched_barrier(0);
            load_rows(PQg, od, pq);
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);                                           // seg 1: K fragments, c2p gather, p2c + S^T MFMA issue, row requests
            if constexpr (KVG) k_tile(kt + 1 < nkt ? kt + 1 : kt, kf);     // in place: K(kt + 1)
            else if constexpr (!STAG) wg_barrier_lds();         // X: every wave has finished gathering the previous tile's image
            stamp(2);                                           // seg 2: barrier X
            // (STAG: the barrier that ended this half's previous P2 already separates those gathers from these stores)
            band_store(img + c * LROWP + 32 * wl, bacc);        // row = key lane (conflict-free); the gather applies pi
            if (extra) band_store(img + c * LROWP + 32 * NSH, bacc2);
            wg_barrier_lds();                                   // Y: image complete; tile kt + 1 is in the ring for everyone (each wave waited for its pieces at the top)
            stamp(3);                                           // seg 3: image stores (wait for the p2c MFMA results) + barrier Y
            if constexpr (KVG) { if (kt + 2 < nkt) dma_tile(kt + 2); }
            else dma_tile(kt + 2 < nkt ? kt + 2 : nkt - 1);     // slot (kt - 1) % 3: every wave is past tile kt - 1.  Unconditional (past the end: the
                                                                // last tile once more, into a slot nobody reads again) so that the request count is fixed
            __builtin_amdgcn_sched_barrier(0);                  // (the PK-row MFMAs stay below: their rows were requested half a phase ago)
            float sv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kc = 16 * (i >> 3) + (i & 7);                         // key offset minus 8h
                const int prow = 16 * (i >> 3) + 8 * ((i >> 2) & 1) + (i & 3);  // pi(key offset) minus 4h: image rows are in lane order
                sv[i] = sacc[i] + img[(prow + 4 * h) * LROWP + 32 * wl + rr_base - kc];
            }
            f32x16 cacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) cacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mm(pk[s], qf[s], cacc);             // c2p of L(kt + 1)  [rr][query c]
            stamp(4);                                           // seg 4: DMA request, image gather (waits for S^T), c2p MFMA issue
            softmax_pv(sv, kt);
            band_store(c2p_l + c * LROW + (xr ^ 32), cacc);     // over the old high block (its gather is long retired)
            stamp(5);                                           // seg 5: softmax, P.V, c2p block store
            if constexpr (DIAG) ++tiles;
            if constexpr (STAG) wg_barrier_lds();               // end of P2: this half's image may be rewritten; V^T(kt) is retired
        };
        if constexpr (STAG) { if (grp == 1) wg_barrier_lds(); } // the late half starts one phase later
        for (int kt = kt_a;;) {
            band_tile(kt, 0);
            if (++kt >= kt_b) break;
            band_tile(kt, 32);
            if (++kt >= kt_b) break;
        }
        if constexpr (STAG) { if (grp == 0) wg_barrier_lds(); } // ... and the early half waits for it at the end
        if constexpr (DIAG) {
            if (a.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {     // 64 workgroups of XCD 0
                unsigned long long* o = a.stamps + ((size_t)(blockIdx.x >> 3) * NW + wave) * 8;
                for (int k = 0; k < 6; ++k) o[k] = seg[k];
                // s_memtime ticks per 100 MHz s_memrealtime tick over the band loop, x1000 (i.e. the clock s_memtime counts, in 0.1 MHz)
                const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk0, dr = __builtin_amdgcn_s_memrealtime() - rt0;
                o[6] = dr ? dc * 1000 / dr : 0; o[7] = tiles;
            }
        }
    }

    sat_tiles(kt_b, nkt, 0);

    if (!active) return;
    if (a.tile_flag && !a.tile_flag[(size_t)b * (Sp >> 5) + (q0 >> 5)]) return;     // pruned last layer: this tile is not read
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if constexpr (sizeof(T) == 2) {
        store_acc32_wide<T>(o0, inv, reinterpret_cast<T*>(outp), h);
        store_acc32_wide<T>(o1, inv, reinterpret_cast<T*>(outp) + 32, h);
    } else if (a.ctx_gs == 2) {
        // GX context rows (glc_common.h: the A operand of the MX cross-term GEMM): per 32 columns [32 hi | 32 lo8 | 32 hi8].  Lane c
        // (h = 0) gets the 8 consecutive columns 16 p .. 16 p + 7 of its query row, lane c + 32 the columns 16 p + 8 .. 16 p + 15: each
        // lane keeps one of its two register groups and receives the matching group of its partner lane (one cross-half exchange per
        // register).  (v_permlane32_swap through its builtin lost its second result here — hipcc, ROCm 7.2, stored the first one twice;
        // the shuffle form is the one that compiled correctly.)
        unsigned char* row = reinterpret_cast<unsigned char*>(a.CTX) + ((size_t)b * Sp + q0 + c) * 4 * a.H;
        auto store_gx = [&](const f32x16& o, int col0) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float own_a = o[8 * p + e] * inv, own_b = o[8 * p + 4 + e] * inv;      // group 2p (columns 16p + 4h + e), group 2p + 1 (+ 8)
                    const float got = __shfl_xor(h ? own_a : own_b, 32, 64);                      // h = 0: the partner's group 2p; h = 1: the partner's group 2p + 1
                    v[e] = h ? got : own_a;           // h = 0: columns 16p + e (own), 16p + 4 + e (partner);  h = 1: 16p + 8 + e (partner), 16p + 12 + e (own)
                    v[4 + e] = h ? own_b : got;
                }
                gx_store8(row, col0 + 16 * p + 8 * h, v, gx_act_khi(a.act_sc), gx_act_klo(a.act_sc), a.gx_sat);
            }
        };
        store_gx(o0, 64 * hh);
        store_gx(o1, 64 * hh + 32);
    } else if (a.ctx_gs) {
        // group-split context rows (glc_kernels.h): the head's 64 columns are groups 2 hh (o0) and 2 hh + 1 (o1) of the row's
        // [32 hi halves | 32 lo halves] groups; hi = f16(v), lo = f16(v - hi), 16-byte stores through the 16-bit store helper
        f16_t* row = reinterpret_cast<f16_t*>(a.CTX) + ((size_t)b * Sp + q0 + c) * 2 * a.H + (size_t)(2 * hh) * 64;
        f32x16 l0, l1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v0 = o0[i] * inv, v1 = o1[i] * inv;
            l0[i] = v0 - (float)(f16_t)v0;
            l1[i] = v1 - (float)(f16_t)v1;
        }
        store_acc32_wide<f16_t>(o0, inv, row, h);
        store_acc32_wide<f16_t>(l0, 1.0f, row + 32, h);
        store_acc32_wide<f16_t>(o1, inv, row + 64, h);
        store_acc32_wide<f16_t>(l1, 1.0f, row + 96, h);
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            store4<OutT>(outp + 8 * g + 4 * h, o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            store4<OutT>(outp + 32 + 8 * g + 4 * h, o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

template <typename T, bool SPLIT, int NW, bool KVG, bool STAG, bool VGL> constexpr size_t wg_lds_bytes() {
    return VGL ? ((size_t)NW * 32 * LROW + 32 * (32 * (NW + 1) + 4)) * sizeof(float) + 3 * 4 * 512 * sizeof(T) :
           KVG ? ((size_t)NW * 32 * LROW + 2 * 32 * (32 * (NW + 1) + 4)) * sizeof(float)
         : STAG ? ((size_t)NW * 32 * LROW + 2 * 32 * (32 * 5 + 4)) * sizeof(float) + 3 * 2 * 4 * 512 * sizeof(T)
                : ((size_t)NW * 32 * LROW + 32 * (32 * (NW + 1) + 4)) * sizeof(float) + 3 * 2 * 4 * 512 * sizeof(T);
}

template <typename T, bool SPLIT, int NW, bool KVG, bool STAG = false, bool VGL = false, bool DIAG = false, bool PD = false, int NMM = 3> const char* launch_wg(hipStream_t st, const AttnArgs& a) {
    static std::atomic<unsigned> raised{0};
    constexpr size_t lds = wg_lds_bytes<T, SPLIT, NW, KVG, STAG, VGL>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (!glc_raise_lds_limit(attn_wg_kernel<T, SPLIT, NW, KVG, STAG, VGL, DIAG, PD, NMM>, (int)lds, raised)) return "attention(wg): cannot raise the dynamic LDS limit";
    static const bool dbg = glc_dev_env("GLC_ATTN_DEBUG") != nullptr;
    if (dbg) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, attn_wg_kernel<T, SPLIT, NW, KVG, STAG, VGL, DIAG, PD, NMM>, 64 * NW, lds);
        fprintf(stderr, "[attn_wg] NW=%d lds=%zu bytes, occupancy API: %d workgroup(s) per CU\n", NW, lds, nb);
    }
    const int nqb = (a.Sp + 32 * NW - 1) / (32 * NW), bh8 = (a.B * a.nh + 7) / 8 * 8;
    hipLaunchKernelGGL((attn_wg_kernel<T, SPLIT, NW, KVG, STAG, VGL, DIAG, PD, NMM>), dim3(nqb * bh8), dim3(64 * NW), lds, st, a);
    return nullptr;
}

}  // namespace

// Same shape contract as glc_launch_attention (attention.hip): head_dim == 64, Sp % 64 == 0, P % 32 == 0, fragment-major operands, offset table.
const char* glc_launch_attention_wg(hipStream_t st, int dtype, const AttnArgs& a_in) {
    AttnArgs a = a_in;
    if (!a.gx_sat) a.gx_sat = glc_gx_sat_ptr();              // fp8 range guard of the GX context rows (ctx_gs == 2)
    if (!a.act_sc) a.act_sc = glc_gx_act_sc();               // ... and the exponent of the activation rows (engine.hip act_sc)
    if (!a.Qh || !a.Kh || !a.Vt || !a.PK || !a.PQ || !a.kbias || !a.klen || !a.kfirst || !a.CTX || !a.otab) return "attention(wg): null pointer";
    if (a.B <= 0 || a.nh <= 0 || a.Sp <= 0 || a.Sp % 64 || a.H != a.nh * 64 || a.P <= 0 || a.P % 32) return "attention(wg): bad shape";
    if (a.sel_b) return "attention(wg): no row selection in this kernel";
    if (a.stamps && !(dtype == GLC_DT_F32 && a.split && !(a.variant & 57))) return "attention(wg): the stamped build exists for the split-f16 8-wave kernel only";
    if (dtype == GLC_DT_F32) {
        if (!a.split) return "attention(wg): the fp32 mode runs this kernel on split-f16 units only";
#ifdef GLC_DEVELOPER      // stamped and timing-only (WRONG results) builds: developer libraries only
        if (a.stamps) return launch_wg<float, true, 8, false, false, false, true>(st, a);
        if (a.variant & 64) return launch_wg<float, true, 8, false, false, false, false, false, 2>(st, a);      // timing-only: two MFMAs per product
#else
        if (a.stamps || (a.variant & 64)) return "attention(wg): stamped and timing-only builds exist in developer builds only (make DEV=1)";
#endif
        if (a.prec) return launch_wg<float, true, 8, false, false, false, false, true>(st, a);      // precision-budget build
        // half-tile stagger: measured same-box 1.42-1.51 vs 1.44-1.45 ms per launch at c3 — no gain, off by default (GLC_ATTN_STAG=1 / variant bit 4)
        static const bool stag_default = glc_dev_env("GLC_ATTN_STAG") != nullptr && atoi(glc_dev_env("GLC_ATTN_STAG")) != 0;
        if (a.variant & 8) return launch_wg<float, true, 8, true>(st, a);
        if (a.variant & 1) return launch_wg<float, true, 4, false, false, true>(st, a);      // diagnostic: 4 waves, K ring only, two workgroups per CU
        const bool stag = (a.variant & 16) ? true : ((a.variant & 32) ? false : stag_default);
        return stag ? launch_wg<float, true, 8, false, true>(st, a) : launch_wg<float, true, 8, false, false>(st, a);
    }
    if (a.split) return "attention(wg): split operands belong to the fp32 mode";
    if (a.variant & 8) return dtype == GLC_DT_BF16 ? launch_wg<bf16_t, false, 4, true>(st, a) : launch_wg<f16_t, false, 4, true>(st, a);
    return dtype == GLC_DT_BF16 ? launch_wg<bf16_t, false, 4, false>(st, a) : launch_wg<f16_t, false, 4, false>(st, a);
}
