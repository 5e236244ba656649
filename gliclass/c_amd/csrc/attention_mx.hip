// DeBERTa-v2/v3 disentangled self-attention, workgroup-shared band kernel on "MX tiles" (round 3) — the attention of the MX pipeline.
//
// The algebra, the work split (8 waves = 8 consecutive 32-query tiles of one (batch, head); K / V^T tiles once per workgroup through an
// LDS-DMA ring; the p2c band shared through one LDS image; c2p as the initial accumulator of S^T; saturated tiles per wave), the
// synchronisation and the vector-memory pipeline are those of attention_wg.hip (split units, NW = 8): read that header first.  What
// changes is the arithmetic of a product and the operand format that goes with it:
//   split units      a*b = a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        3 x v_mfma_f32_32x32x16_f16 per 16 columns       (96 cycles)
//   MX tiles         a*b = a_hi*b_hi                                 1 x v_mfma_f32_32x32x16_f16 per 16 columns       (32 cycles)
//                        + (a_hi*b_lo + a_lo*b_hi)                   1 x v_mfma_scale_f32_32x32x64_f8f6f4 per 32      (64 cycles)
// i.e. 2/3 of the matrix-pipe time and half the MFMA instructions (a timing-only build of the split kernel with two f16 MFMAs per
// product ran 1.10 vs 1.40 ms at c3: scripts/attn_bench.py variant 68).  An operand tile (glc_layout.h "MX tiles") is its four f16
// units + two MX steps whose lane carries 32 bytes [first | second] = the fp8 parts of 16 columns in the order its tensor always
// travels in — K, PK, V^T as (lo8 | hi8), Q, PQ, P as (hi8 | lo8) — so every product pairs a (lo8 | hi8) operand with a (hi8 | lo8)
// one: MX block 0 multiplies lo8 x hi8, block 1 hi8 x lo8, and the 2^-GLC_GX_SHIFT of the lo8 parts comes back through the e8m0 scale
// of the block that holds them (a lane of half h supplies the scale of block h).  Cross terms are ~2^-11 of a product and come out
// to ~4 bits: ~2^-15 relative per product (split units: 2^-21; single f16: 2^-11).  The probabilities are split on the fly:
// P = f16(p) for the f16 MFMAs, p_hi8 = e4m3(p), p_lo8 = e4m3((p - f16(p)) 2^SHIFT) for the scaled one (p <= 2^8 with the deferred
// rescale: inside the e4m3 range).  Everything else — scores, softmax, the position gathers, the fp32 accumulators — is unchanged.
//
// Two workgroup shapes of the same code (template parameter NW):
//   NW = 8  one workgroup per CU: 8 query tiles, ring of three [K | V^T] slots (156 KB of LDS), key tile t + 2 requested after barrier Y
//           of tile t, two barriers per band tile.
//   NW = 4  TWO workgroups per CU (default since the end of round 3): 4 query tiles, 80 KB of LDS each — one K slot and two V^T slots:
//           K(t + 1) and V^T(t + 1) are requested once every wave holds its K(t) fragments (after barrier X) and published by a third
//           barrier Z at the start of tile t + 1.  Twice the K / V^T bytes per query and a quarter instead of an eighth of the tiles
//           carrying the unowned block, but the two workgroups of a CU are independent: their phases do not coincide, and a barrier
//           holds 4 waves instead of 8.  Same box, c3: 1.22 -> 1.15 ms per launch, forward 37.84 -> 37.19 / 37.47 ms.
#include <stdio.h>
#include <stdlib.h>
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"
#include "glc_pfrag.h"

namespace {

constexpr float RESCALE_THR = 8.0f;   // log2 units (as attention.hip)
constexpr int LROW = 68;              // floats per c2p ring row (2 blocks of 32 + 4 pad)
constexpr int TILEB = GLC_MXT_BYTES;  // one K tile, or one V^T tile
constexpr int SLOTB = 2 * TILEB;      // NW = 8 ring slot: K tile | V^T tile
template <int NW> constexpr size_t mx_lds_bytes() {        // c2p rings + p2c image + [NW = 8: 3 x (K | V^T); NW = 4: K | V^T even | V^T odd]
    return ((size_t)NW * 32 * LROW + 32 * (32 * (NW + 1) + 4)) * sizeof(float) + (NW == 8 ? 3 * SLOTB : 3 * TILEB);
}
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
__device__ __forceinline__ void wg_barrier_all() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ i32x8 cat8(const i32x4& a, const i32x4& b) {
    i32x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_mx[];

// ABL: timing-only builds (wrong results; AttnArgs::variant bits 8 / 9 through glc_debug_attn_bench): 1 = without the block-scaled MFMAs,
// 2 = without the fp8 conversion of the probabilities.  ABL = 3 is a MEASUREMENT build with right results at lower precision: P and V^T at
// single f16 in the P.V product (no P split, no scaled MFMA there) — what the precision budget's one affordable cut (docs/LOG_r01-r05.md §2) buys in
// time; GLC_ATTN_PV16=1 or variant bit 12, never the default.
// DIAG: s_memtime stamps at the phase boundaries of a band tile, summed per wave in SGPRs (glc_debug_attn_bench prints them; the stamps
// pin the instruction order at each boundary, so the stamped build is slower than the one it describes).
// FIXQ (round 5, default): a wave keeps ITS PQ block (rel-block g = slot - key tile, g mod NW == wave) in registers while it is image slot
// 0 .. NW - 1 and writes its p2c block to the slot the block has reached; the block that enters takes the registers in place (glc_pfrag.h:
// one asm block with tied operands) — 8 KB of position rows per wave every NW-th key tile instead of every tile (row requests 18 -> 12 KB per
// wave and tile; the L2 -> CU path is the band kernels' busiest resource: docs/LOG_r01-r05.md §3g).  Results bit-identical (the same products).
// DIET (round 6, default): the band loop's look-ups and row requests without per-lane address arithmetic.  AttnArgs::mtab holds, per lane half and
// distance entry, ONE ready-made offset of the row inside the PLANAR copy of its position table (glc_layout.h: every 16-byte piece of a row — four f16
// units, four MX planes — at the same per-lane offset, 1 KiB apart), the PK half stored backwards; the entry index splits into a wave-uniform base that
// moves by one block per key tile (scalar) and the lane's column, a 32-bit offset fixed at kernel entry, and the table is padded so that no index the
// loop forms needs a clamp.  A look-up is one global_load_dword, a row block eight global_load_dwordx4 on one offset register: 33 of round 5's 245
// vector instructions per key tile gone, one loop-carried register fewer.  With it the c2p ring holds its blocks with the table rows in DESCENDING
// order (lane c of the PK fragment brings the row of lane 31 - c, the blocks swap ring halves): the even tile's start values S^T <- c2p come out of
// ascending ds_read2_b32 pairs in accumulator order — 16 v_mov per two tiles gone.  Same products, same sums: bit-identical context rows.
template <int NW, int ABL = 0, bool DIAG = false, bool RECOMP = false, bool FIXQ = true, bool XROT = FIXQ, bool DIET = true>
__global__ __launch_bounds__(64 * NW, 2) void attn_mx_kernel(AttnArgs a) {
    static_assert(!DIET || FIXQ, "the round-6 table serves the resident-block form only");
    constexpr bool RV = DIET;
    static_assert(NW == 8 || NW == 4, "workgroup shapes: 8 waves x 1 per CU, 4 waves x 2 per CU");
    constexpr int LROWP = 32 * (NW + 1) + 4;            // floats per p2c image row
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int Sp = a.Sp;
    // e8m0 scales of the block-scaled MFMA (lane half h supplies block h): an operand that travels as (lo8 | hi8) / as (hi8 | lo8)
    // (both in ONE register: byte 0 = the (lo8 | hi8) value, byte 1 = the (hi8 | lo8) value, picked by the instruction's op_sel)
    const int SC = h ? (127 | ((127 - GLC_GX_SHIFT) << 8)) : ((127 - GLC_GX_SHIFT) | (127 << 8));
    // a product on MX tiles: `lh` travels as (lo8 | hi8), `hl` as (hi8 | lo8); D[row of the FIRST argument][row of the second]
    // (ABL 4, timing only, wrong results: an f16 MFMA issued as two v_mfma_f32_16x16x32_f16 on the same operand and accumulator registers — the power
    // of the shape, profiles/r04/mfma_power_probe.txt)
    auto f16mm = [&](const f16x8& x, const f16x8& y, f32x16& acc, int s) __attribute__((always_inline)) {
        if constexpr (ABL == 4) {
            typedef float f32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int b = 8 * (s & 1) + 4 * t;
                f32x4_ sub = {acc[b], acc[b + 1], acc[b + 2], acc[b + 3]};
                sub = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, sub, 0, 0, 0);
                acc[b] = sub[0]; acc[b + 1] = sub[1]; acc[b + 2] = sub[2]; acc[b + 3] = sub[3];
            }
        } else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0);
    };
    auto mm_lh_hl = [&](const MxFrag& lh, const MxFrag& hl, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) f16mm(lh.f[s], hl.f[s], acc, s);
        if constexpr (ABL != 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(lh.x[m], hl.x[m], acc, 0, 0, 0, SC, 1, SC);
        }
    };
    auto mm_hl_lh = [&](const MxFrag& hl, const MxFrag& lh, f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) f16mm(hl.f[s], lh.f[s], acc, s);
        if constexpr (ABL != 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hl.x[m], lh.x[m], acc, 0, 0, 1, SC, 0, SC);
        }
    };

    float* c2p_l = reinterpret_cast<float*>(smem_mx) + (size_t)wave * 32 * LROW;                      // this wave's ring [32 q][64 + 4]
    float* p2c_img = reinterpret_cast<float*>(smem_mx) + (size_t)NW * 32 * LROW;                      // shared [32 keys][LROWP]
    unsigned char* kv_ring = smem_mx + ((size_t)NW * 32 * LROW + 32 * LROWP) * sizeof(float);         // 3 x (K tile | V^T tile)

    const int nqb = (Sp + 32 * NW - 1) / (32 * NW);
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int bh = xcd + 8 * (jj / nqb);
    const int Q0 = (jj % nqb) * 32 * NW;
    const int QX = Q0 + 32 * NW;                        // the query tile whose LOW block is the workgroup's unowned (last high) block
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

    const int nt = Sp >> 5;
    const unsigned char* __restrict__ Qg = reinterpret_cast<const unsigned char*>(a.Qh) + ((size_t)bh * nt + (q0m >> 5)) * TILEB;
    const unsigned char* __restrict__ Kg = reinterpret_cast<const unsigned char*>(a.Kh) + (size_t)bh * nt * TILEB;
    const unsigned char* __restrict__ Vg = reinterpret_cast<const unsigned char*>(a.Vt) + (size_t)bh * nt * TILEB;
    const unsigned char* __restrict__ PKg = reinterpret_cast<const unsigned char*>(a.PK) + (size_t)hh * (a.P >> 5) * TILEB;
    const unsigned char* __restrict__ PQg = reinterpret_cast<const unsigned char*>(a.PQ) + (size_t)hh * (a.P >> 5) * TILEB;
    const float* __restrict__ kb = a.kbias + (size_t)b * Sp;

    int nkt = (klen + 31) >> 5;
    nkt = nkt < 1 ? 1 : (nkt > nt ? nt : nkt);
    const int kfirst = a.kfirst[b];
    const int foff = 8 * h;

    // Position rows: otab entry (q - k) + Sp - 1 + 64 = byte offsets of row delta(q - k) in the SPLIT-unit PQ (x) / PK (y) layouts:
    // (delta >> 5) * 8192 + slot * 32.  In the MX tile of the same rows: f16 unit s at tile + s * 1024 + h * 512 + slot * 16,
    // MX step m at tile + 4096 + m * 2048 + h * 1024 + slot * 32.
    const int otab_max = 2 * Sp - 2 + 128;
    auto block_x = [&](int qb, int t) -> int {
        int idx = qb - 32 * t - 31 + c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return reinterpret_cast<const int*>(a.otab)[2 * idx];
    };
    auto block_xy = [&](int qb, int t) -> int2 {        // (x, y): the row's offsets in the PQ and in the PK layout (the table holds both)
        int idx = qb - 32 * t - 31 + c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return reinterpret_cast<const int2*>(a.otab)[idx];
    };
    auto pk_of_pq = [&](int x) -> int {                  // same table row in the K layout: its slot is pi32-permuted
        const int r = (x >> 5) & 31;
        return x + ((glc_pi32(r) - r) << 5);
    };
    // DIET: planar copies of this head's tables and the look-ups into AttnArgs::mtab.  Entry index of block (qb, t), lane column c: J0 + c with
    // J0 = qb - 32 t - 31 + Sp + 63 (0 <= J0 + c < NE for every t in [-1, Sp / 32 + 1], qb <= Sp + 32 NW); PQ rows: entry J0 + c (.x); PK rows:
    // lane c brings the row of column 31 - c = backwards entry (NE - 32 - J0) + c (.y).
    const int NE = 2 * Sp + 512;
    const unsigned char* __restrict__ mtb = reinterpret_cast<const unsigned char*>(a.mtab);
    const unsigned char* __restrict__ PKp = PKg + (size_t)a.nh * (a.P >> 5) * TILEB;
    const unsigned char* __restrict__ PQp = PQg + (size_t)a.nh * (a.P >> 5) * TILEB;
    const unsigned vl = (unsigned)((h * NE + c) * 8);
    auto look = [&](const unsigned char* tb) -> int {      // tb: wave-uniform, kept in scalar registers (opaque: hipcc would otherwise fold it into a 64-bit per-lane address)
        asm volatile("" : "+s"(tb));
        return *reinterpret_cast<const int*>(tb + (size_t)vl);
    };
    auto rows_q = [&](int qb, int t) -> int { return look(mtb + (ptrdiff_t)(qb - 32 * t - 31 + Sp + 63) * 8); };
    auto rows_k = [&](int qb, int t) -> int { return look(mtb + ((ptrdiff_t)(NE - 32 - (qb - 32 * t - 31 + Sp + 63)) * 8 + 4)); };
    auto load_rows2 = [&](const unsigned char* base, const int off, MxFrag& f) __attribute__((always_inline)) {      // base: a planar copy
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(base + (size_t)(unsigned)off + s * 1024);
#pragma unroll
        for (int m = 0; m < 2; ++m)
            f.x[m] = cat8(*reinterpret_cast<const i32x4*>(base + 4096 + (size_t)(unsigned)off + m * 2048), *reinterpret_cast<const i32x4*>(base + 4096 + (size_t)(unsigned)off + (m * 2048 + 1024)));
    };
    auto load_rows = [&](const unsigned char* base, int off, MxFrag& f) __attribute__((always_inline)) {       // gathered table rows: off = split-form offset
        // (uniform base + unsigned 32-bit lane offset + immediate: the scalar-base form of global_load, no 64-bit address arithmetic per lane)
        const unsigned vf = (unsigned)((off & ~8191) + ((off & 8191) >> 1) + h * 512);
        const unsigned vx = (unsigned)(off + 4096 + h * 1024);
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(base + (size_t)vf + s * 1024);
#pragma unroll
        for (int m = 0; m < 2; ++m)
            f.x[m] = cat8(*reinterpret_cast<const i32x4*>(base + (size_t)vx + m * 2048), *reinterpret_cast<const i32x4*>(base + (size_t)vx + (m * 2048 + 16)));
    };
    // ring images: f16 units as they are (16 B per lane), MX steps re-arranged by the DMA into [64 lanes x first | 64 lanes x second]
    auto k_tile = [&](int t, MxFrag& f) __attribute__((always_inline)) {
        const unsigned char* tile = NW == 8 ? kv_ring + (size_t)(t % 3) * SLOTB : kv_ring;      // (NW = 4: the one K slot)
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
    // LDS-DMA of key tile t into ring slot t % 3: wave w moves the 1-KiB pieces 2 w, 2 w + 1 of [K tile | V^T tile].  K tile: pieces 0-3 the
    // f16 units, 4-7 the two MX steps; V^T tile (pieces 8-15): per 4-KiB sub-tile [2 f16 units | one MX step].  An MX step's two pieces
    // are its lanes' first / second 16 bytes (per-lane source address: the DMA gathers).
    const unsigned off16 = lane * 16, off32 = lane * 32;
    const int piece_src = (wave & 3) * 2048;                          // byte offset of this wave's piece pair inside its tile (memory and LDS image alike)
    auto uniform_ptr = [](const unsigned char* q) -> const unsigned char* {
        const unsigned long long v = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
    };
    auto dma_pair = [&](const unsigned char* src, unsigned char* dst, const bool mx_piece) __attribute__((always_inline)) {
        if (mx_piece) { glds16_sv(uniform_ptr(src), off32, dst); glds16_sv(uniform_ptr(src + 16), off32, dst + 1024); }
        else { glds16_sv(uniform_ptr(src), off16, dst); glds16_sv(uniform_ptr(src + 1024), off16, dst + 1024); }
    };
    // NW = 8: key tile t into ring slot t % 3, waves 0-3 the K tile's four piece pairs, waves 4-7 the V^T tile's.
    // NW = 4: K into the one K slot, V^T into V slot t & 1; wave w moves the pairs the 8-wave form gives to its waves w and 4 + w.
    auto dma_tile = [&](int t) {
        if constexpr (NW == 8) {
            dma_pair((wave < 4 ? Kg : Vg) + (size_t)t * TILEB + piece_src, kv_ring + (size_t)(t % 3) * SLOTB + (wave < 4 ? 0 : TILEB) + piece_src,
                     wave < 4 ? wave >= 2 : (wave & 1));
        } else {
            dma_pair(Kg + (size_t)t * TILEB + piece_src, kv_ring + piece_src, wave >= 2);
            dma_pair(Vg + (size_t)t * TILEB + piece_src, kv_ring + TILEB + (size_t)(t & 1) * TILEB + piece_src, (wave & 1) != 0);
        }
    };

    MxFrag qf;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf.f[s] = *reinterpret_cast<const f16x8*>(Qg + s * 1024 + lane * 16);
#pragma unroll
    for (int m = 0; m < 2; ++m) qf.x[m] = cat8(*reinterpret_cast<const i32x4*>(Qg + 4096 + m * 2048 + lane * 32), *reinterpret_cast<const i32x4*>(Qg + 4096 + m * 2048 + lane * 32 + 16));
    dma_tile(0);
    if (NW == 8 && nkt > 1) dma_tile(1);
    MxFrag kf;

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    const int rr_base = c - 8 * h + 31;
    float one_f = 1.0f;
    asm volatile("" : "+s"(one_f));      // opaque to the optimiser: fma(p, 1, -half) stays a v_fma_mix_f32

    // Shared tail of every key tile: key bias, online softmax (log2 units, deferred rescale), P*V with V^T from the ring.
    auto softmax_pv = [&](float (&sv)[16], int kt) __attribute__((always_inline)) {
        const int k0 = kt * 32;
        const unsigned char* vtile = NW == 8 ? kv_ring + (size_t)(kt % 3) * SLOTB + TILEB : kv_ring + TILEB + (size_t)(kt & 1) * TILEB;
        // (the two 32-row halves of V^T one after the other: the second half's fragments are read under the first half's MFMAs —
        //  both resident at once cost 16 registers the loop does not have)
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
        // (pairs as 2-vectors: v_pk_add_f32 — this file is built without SLP packing, so the pairs are explicit)
        const f32x2 m2 = {m, m};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 d = (f32x2){sv[i], sv[i + 1]} - m2;
            sv[i] = __builtin_amdgcn_exp2f(d[0]); sv[i + 1] = __builtin_amdgcn_exp2f(d[1]);
            ps2 += (f32x2){sv[i], sv[i + 1]};
        }
        l += ps2[0] + ps2[1];
        // P travels as (hi8 | lo8): f16(p) for the f16 MFMAs (k-step t = keys 16 t + 8 h + j), fp8 parts of the 16 keys for the scaled one
        f16x8 pf[2];
        i32x8 px;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[t][j] = (f16_t)sv[8 * t + j];
        }
        if constexpr (ABL == 2) { px = qf.x[0]; }
        else if constexpr (ABL == 3) { }
        else
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int wh0;
            asm volatile("" : "=v"(wh0));      // (both halves are written below: no start value to materialise)
            int wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * q], sv[4 * q + 1], wh0, false);
            wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * q + 2], sv[4 * q + 3], wh, true);
            px[q] = wh;
            // lo8 = e4m3((p - f16(p)) 2^SHIFT): the residual as ONE mixed-precision FMA per value (p * 1 - f16 half, read from the packed
            // operand) and the 2^SHIFT inside the conversion (v_cvt_scalef32_pk_fp8_f32 divides by its scale; |residual| 2^SHIFT <= 256: no overflow)
            // (inline asm: hipcc converts every value to f16 a second time for a C-level fma(p, 1, -half) instead of reading the halves of the
            //  packed operand; all three inputs are VALU results — v_exp_f32, an SGPR, v_cvt_pk_f16_f32 — so no LDS / MFMA hazard is involved)
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
            { int wl0; asm volatile("" : "=v"(wl0)); wl2 = __builtin_bit_cast(v2i16, wl0); }
            wl2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl2, r[0], r[1], 1.0f / (float)(1 << GLC_GX_SHIFT), false);
            wl2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl2, r[2], r[3], 1.0f / (float)(1 << GLC_GX_SHIFT), true);
            px[4 + q] = __builtin_bit_cast(int, wl2);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) f16mm(vf[t], pf[t], o0, t);      // O^T[dd][query c]
        if constexpr (ABL != 1 && ABL != 3) o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o0, 0, 0, 0, SC, 1, SC);
        __builtin_amdgcn_sched_barrier(0);
        load_v(1);
#pragma unroll
        for (int t = 0; t < 2; ++t) f16mm(vf[t], pf[t], o1, t);
        if constexpr (ABL != 1 && ABL != 3) o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o1, 0, 0, 0, SC, 1, SC);
    };

    // key-tile ranges, workgroup-uniform (attention_wg.hip)
    int kt_a = Q0 - 31 - a.rsat_pos >= 0 ? (Q0 - 31 - a.rsat_pos) / 32 + 1 : 0;
    kt_a = kt_a > nkt ? nkt : kt_a;
    int kt_b = (Q0 + 32 * (NW - 1) + 31 - a.rsat_neg + 31) / 32;
    kt_b = kt_b < kt_a ? kt_a : (kt_b > nkt ? nkt : kt_b);

    // Saturated key tiles: delta is ONE value d*: c2p = Q_q.PK[d*] a per-query constant, p2c = K_k.PQ[d*] a second product on the same K tile
    auto sat_tiles = [&](int kt_lo, int kt_hi, int dstar) {
        if (kt_lo >= kt_hi) return;
        MxFrag pqb, pkb;                         // broadcast fragments: every row / column is table row d*
        load_rows(PQg, (dstar >> 5) * 8192 + (dstar & 31) * 32, pqb);
        load_rows(PKg, (dstar >> 5) * 8192 + glc_pi32(dstar & 31) * 32, pkb);
        float cq;
        {
            f32x16 t;
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = 0.f;
            mm_lh_hl(pkb, qf, t);                // every row = PK[d*] . Q_c
            cq = t[0];
        }
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            if constexpr (NW == 8) { if (kt + 2 < nkt) dma_tile(kt + 2); }
            else wg_barrier_all();               // Z: key tile kt (K and V^T) is in LDS for everyone
            k_tile(kt, kf);
            if constexpr (NW == 4) {
                wg_barrier_lds();                // every wave holds its K fragments: the K slot is free
                if (kt + 1 < nkt) dma_tile(kt + 1);
            }
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = cq;
            mm_lh_hl(kf, qf, sacc);
            mm_lh_hl(kf, pqb, sacc);             // + K_k . PQ[d*] (same for every query column)
            float sv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = sacc[i];
            softmax_pv(sv, kt);
            if constexpr (NW == 8) wg_barrier_all();
        }
    };

    if constexpr (NW == 8) wg_barrier_all();       // tile 0 is in the ring
    sat_tiles(0, kt_a, a.P - 1);

    if (kt_a < kt_b) {
        // ---- band prologue: this wave's c2p blocks L(kt_a - 1) and L(kt_a) ----
        {
            MxFrag pk;
            f32x16 bacc;
            if constexpr (DIET) load_rows2(PKp, rows_k(q0, kt_a - 1), pk);
            else load_rows(PKg, pk_of_pq(block_x(q0, kt_a - 1)), pk);
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
            mm_lh_hl(pk, qf, bacc);
            band_store(c2p_l + c * LROW + (RV ? 0 : 32), bacc);      // ring half 1 (RV: half 0)
            if constexpr (DIET) load_rows2(PKp, rows_k(q0, kt_a), pk);
            else load_rows(PKg, pk_of_pq(block_x(q0, kt_a)), pk);
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
            mm_lh_hl(pk, qf, bacc);
            band_store(c2p_l + c * LROW + (RV ? 32 : 0), bacc);      // ring half 0 (RV: half 1)
            wave_lds_sync();
        }
        // (RV: element rr of the logical [L(t) | L(t - 1)] pair sits at float 63 - rr of the ring row on even tiles, at (63 - rr) ^ 32 on odd ones)
        const float* c2p_even = c2p_l + c * LROW + (RV ? 63 - rr_base : rr_base);
        MxFrag pq, pqx;
        PFrag pqr;                                              // FIXQ: the resident block
        int eq_n = 0;                                           // FIXQ: table offset (x) of the block that enters at the next key tile but one
        auto rows_vf = [&](int off) -> unsigned { return (unsigned)((off & ~8191) + ((off & 8191) >> 1) + h * 512); };
        auto rows_vx = [&](int off) -> unsigned { return (unsigned)(off + 4096 + h * 1024); };
        if constexpr (FIXQ && DIET) {
            const unsigned vo = (unsigned)rows_q(Q0 + 32 * ((wave + kt_a) & (NW - 1)), kt_a);
#pragma unroll
            for (int s = 0; s < 4; ++s) pqr.f[s] = *reinterpret_cast<const f16x8*>(PQp + (size_t)vo + s * 1024);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                pqr.xa[m] = *reinterpret_cast<const glc_i32x4*>(PQp + 4096 + (size_t)vo + m * 2048);
                pqr.xb[m] = *reinterpret_cast<const glc_i32x4*>(PQp + 4096 + (size_t)vo + (m * 2048 + 1024));
            }
            eq_n = rows_q(Q0, kt_a + 1);
        } else if constexpr (FIXQ) {
            const int offp = block_x(Q0 + 32 * ((wave + kt_a) & (NW - 1)), kt_a);
            const unsigned vf = rows_vf(offp), vx = rows_vx(offp);
#pragma unroll
            for (int s = 0; s < 4; ++s) pqr.f[s] = *reinterpret_cast<const f16x8*>(PQg + (size_t)vf + s * 1024);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                pqr.xa[m] = *reinterpret_cast<const glc_i32x4*>(PQg + (size_t)vx + m * 2048);
                pqr.xb[m] = *reinterpret_cast<const glc_i32x4*>(PQg + (size_t)vx + (m * 2048 + 16));
            }
            eq_n = block_x(Q0, kt_a + 1);
        } else load_rows(PQg, block_x(q0, kt_a), pq);
        // the wave that computes the block nobody owns at key tile t: t mod NW in round 4's form; with resident blocks the wave NW - 2 - t (mod NW) —
        // never the wave that requests its entering block during the tile before (that one already issues 16 row requests, with these 24)
        auto extra_wave = [&](int t) -> int { return XROT ? ((NW - 2 - t) & (NW - 1)) : (t % NW); };
        if (extra_wave(kt_a) == wave) { if constexpr (DIET) load_rows2(PQp, rows_q(QX, kt_a), pqx); else load_rows(PQg, block_x(QX, kt_a), pqx); }
        int2 od_n = DIET ? (int2){0, rows_k(q0, kt_a + 1)} : block_xy(q0, kt_a + 1);      // (DIET: .y = the PK rows' planar offset, .x unused)
        int odx_n = DIET ? rows_q(QX, kt_a + 1) : block_x(QX, kt_a + 1);
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
        auto band_tile = [&](const int kt, const int xr) __attribute__((always_inline)) {
            const bool extra = extra_wave(kt) == wave;          // wave-uniform: this wave also computes the block nobody owns
            stamp(-1);
            // everything requested during tile kt - 1 has arrived (rows, offsets, my DMA pieces); NW = 4: barrier Z — key tile kt is in LDS for everyone
            if constexpr (NW == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else wg_barrier_all();
            stamp(0);                                           // seg 0: wait for last tile's requests
            if constexpr (FIXQ) asm volatile("" : "+v"(pqr.f[0]), "+v"(pqr.f[1]), "+v"(pqr.f[2]), "+v"(pqr.f[3]), "+v"(pqr.xa[0]), "+v"(pqr.xb[0]), "+v"(pqr.xa[1]), "+v"(pqr.xb[1]));      // (the block's uses stay behind the wait: its in-place request is not tracked by the compiler)
            const int eq = eq_n;                                // x offset of block_x(Q0, kt + 1)
            if constexpr (FIXQ) eq_n = DIET ? rows_q(Q0, kt + 2) : block_x(Q0, kt + 2);
            const int jm = FIXQ ? ((wave + kt) & (NW - 1)) : wave;      // image slot of this wave's p2c block
            MxFrag pk;
            const int2 od = od_n;
            const int odx = odx_n;
            if constexpr (DIET) { od_n.y = rows_k(q0, kt + 2); odx_n = rows_q(QX, kt + 2); }
            else { od_n = block_xy(q0, kt + 2); odx_n = block_x(QX, kt + 2); }
            k_tile(kt, kf);
            float* img = p2c_img;
            f32x16 sacc;
            // (odd steps: the 16 gather addresses (rr_base - kc) ^ 32 are no affine function of the lane's base; as loop invariants the compiler
            //  spills them — 16 of the kernel's 19 spilled registers, reloaded through the vector-memory path every second tile, 119 MB of
            //  scratch writes per launch (profiles/r04/attn_hbm_bytes.txt).  RECOMP (default since round 4): recompute them from an opaque copy of the
            //  base instead — bit-identical, 7 spilled registers left, -1.5 % per launch in the microbenchmark)
            int rbo = RV ? 4 * (63 - rr_base) : rr_base;
            if constexpr (RECOMP) asm volatile("" : "+v"(rbo));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kc = 16 * (i >> 3) + (i & 7);
                // (odd tiles: byte address = ((4 (63 - rr_base) + 4 kc) ^ 128) + row base: one add, one v_xad_u32 per element — hipcc's own form of
                //  the index expression takes three)
                if constexpr (RV) sacc[i] = xr ? *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(c2p_l + c * LROW) + (((unsigned)rbo + 4u * kc) ^ 128u)) : c2p_even[kc];
                else sacc[i] = xr ? c2p_l[c * LROW + (((RECOMP ? rbo : rr_base) - kc) ^ 32)] : c2p_even[-kc];
            }
            // ---- p2c: low block of this wave, and, one wave per tile, the high block of the last wave ----
            f32x16 bacc, bacc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
            if constexpr (FIXQ) {
#pragma unroll
                for (int s = 0; s < 4; ++s) f16mm(pqr.f[s], kf.f[s], bacc, s);
                if constexpr (ABL != 1) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) bacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(pqr.xa[m], pqr.xb[m]), kf.x[m], bacc, 0, 0, 1, SC, 0, SC);
                }
            } else mm_hl_lh(pq, kf, bacc);
            if (extra) {
#pragma unroll
                for (int i = 0; i < 16; ++i) bacc2[i] = 0.f;
                mm_hl_lh(pqx, kf, bacc2);
            }
            // ---- S^T = K Q^T + c2p ----
            mm_lh_hl(kf, qf, sacc);
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);                                           // seg 1: K fragments, c2p gather, p2c + S^T MFMA issue
            if (extra_wave(kt + 1) == wave) { if constexpr (DIET) load_rows2(PQp, odx, pqx); else load_rows(PQg, odx, pqx); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DIET) load_rows2(PKp, od.y, pk);
            else load_rows(PKg, od.y, pk);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FIXQ && DIET) pfrag_load_planar_if(jm == NW - 1 && kt + 1 < kt_b, PQp, PQp + 4096, (unsigned)eq, pqr);
            else if constexpr (FIXQ) pfrag_load_if(jm == NW - 1 && kt + 1 < kt_b, PQg, rows_vf(eq), rows_vx(eq), pqr);      // the block's last tile as a main slot: the block that enters takes its registers
            else load_rows(PQg, od.x, pq);
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);                                           // seg 2: row requests (8 waves x 16-24 KB through the CU's vector-memory path)
            wg_barrier_lds();                                   // X: every wave has finished gathering the previous tile's image
            stamp(3);                                           // seg 3: barrier X
            band_store(img + c * LROWP + 32 * jm, bacc);
            if (extra) band_store(img + c * LROWP + 32 * NW, bacc2);
            wg_barrier_lds();                                   // Y: image complete; tile kt + 1 is in the ring for everyone
            stamp(4);                                           // seg 4: image stores (wait for the p2c MFMA results) + barrier Y
            if constexpr (NW == 8) dma_tile(kt + 2 < nkt ? kt + 2 : nkt - 1);
            else if (kt + 1 < nkt) dma_tile(kt + 1);           // (the K slot is free since barrier X)
            __builtin_amdgcn_sched_barrier(0);
            float sv[16];
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const int kc = 16 * (i >> 3) + (i & 7);
                const int prow = 16 * (i >> 3) + 8 * ((i >> 2) & 1) + (i & 3);
                const f32x2 g = (f32x2){sacc[i], sacc[i + 1]} +
                                (f32x2){img[(prow + 4 * h) * LROWP + 32 * wave + rr_base - kc], img[(prow + 1 + 4 * h) * LROWP + 32 * wave + rr_base - kc - 1]};
                sv[i] = g[0]; sv[i + 1] = g[1];
            }
            f32x16 cacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) cacc[i] = 0.f;
            stamp(5);                                           // seg 5: DMA request, image gather (waits for S^T)
            mm_lh_hl(pk, qf, cacc);                             // c2p of L(kt + 1)  [rr][query c]
            softmax_pv(sv, kt);
            stamp(6);                                           // seg 6: c2p MFMA issue, softmax, P.V issue
            band_store(c2p_l + c * LROW + (RV ? xr : xr ^ 32), cacc);
            stamp(7);                                           // seg 7: c2p block store (waits for the matrix pipe to drain)
            if constexpr (DIAG) ++tiles;
        };
        for (int kt = kt_a;;) {
            band_tile(kt, 0);
            if (++kt >= kt_b) break;
            band_tile(kt, 32);
            if (++kt >= kt_b) break;
        }
        if constexpr (DIAG) {
            if (a.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {     // 64 workgroups of XCD 0
                unsigned long long* o = a.stamps + ((size_t)(blockIdx.x >> 3) * NW + wave) * 10;
                for (int k = 0; k < 8; ++k) o[k] = seg[k];
                // s_memtime ticks per 100 MHz s_memrealtime tick over the band loop, x1000 (the clock s_memtime counts, in 0.1 MHz)
                const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk0, dr = __builtin_amdgcn_s_memrealtime() - rt0;
                o[8] = dr ? dc * 1000 / dr : 0; o[9] = tiles;
            }
        }
    }

    sat_tiles(kt_b, nkt, 0);

    if (!active) return;
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // GX context rows (as attention_wg.hip, ctx_gs == 2)
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


// split-f16 units [8 hi | 8 lo] of a Q / K layout tensor (rows in tiles of 32, 64 columns: the position tables at load) -> MX tiles;
// hl != 0: the tensor travels as (hi8 | lo8) (PQ), else as (lo8 | hi8) (PK).  One thread per (tile, lane slot r): the row's 64 columns.
// planar != 0: the MX steps as two 16-byte planes per step ([64 lanes x first | 64 lanes x second], 1 KiB each — the form the K tiles take in the LDS ring)
// instead of 32 bytes per lane: every 16-byte piece of a row then sits at the same per-lane offset (attention_mx.hip DIET).
__global__ __launch_bounds__(64) void units_to_mxt_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int ntiles, int hl, unsigned* sat, int planar) {
    const int tile = blockIdx.x, r = threadIdx.x & 31, half = threadIdx.x >> 5;      // half: columns 32 half .. 32 half + 31
    if (tile >= ntiles) return;
    const unsigned char* st = src + (size_t)tile * 8192;
    unsigned char* dt = dst + (size_t)tile * 8192;
    for (int g = 4 * half; g < 4 * half + 4; ++g) {           // 8 columns at a time: e0 = 8 g -> unit s = g >> 1, lane half h = g & 1
        const int s = g >> 1, hh = g & 1;
        const f16_t* u = reinterpret_cast<const f16_t*>(st + s * 2048 + (32 * hh + r) * 32);
        float v[8];
        gs_h8 hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = u[j]; v[j] = (float)u[j] + (float)u[8 + j]; }
        gx_range_note(v, 1.0f, sat);
        *reinterpret_cast<gs_h8*>(dt + glc_mxt_f16(0, r, 8 * g)) = hi;
        float lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) lo[j] = (v[j] - (float)hi[j]) * (float)(1 << GLC_GX_SHIFT);
        const u32x2 l8 = {glc_fp8x4(lo[0], lo[1], lo[2], lo[3]), glc_fp8x4(lo[4], lo[5], lo[6], lo[7])};
        const u32x2 h8 = {glc_fp8x4(v[0], v[1], v[2], v[3]), glc_fp8x4(v[4], v[5], v[6], v[7])};
        unsigned char* px = dt + (planar ? glc_mxt_mx_planar(0, r, 8 * g) : glc_mxt_mx(0, r, 8 * g));
        *reinterpret_cast<u32x2*>(px) = hl ? h8 : l8;
        *reinterpret_cast<u32x2*>(px + (planar ? 1024 : 16)) = hl ? l8 : h8;
    }
}

}  // namespace

// Same contract as glc_launch_attention_wg with split units, except: Qh / Kh / Vt / PQ / PK hold MX tiles (glc_layout.h) and CTX is
// written as GX rows; otab is the split-unit offset table (engine.hip).  No row selection, no tile flags (the pruned layer keeps its kernel).
namespace {
template <int NW> const char* launch_mx(hipStream_t st, const AttnArgs& a) {
    constexpr size_t lds = mx_lds_bytes<NW>();
    static_assert((NW == 8 ? 1 : 2) * lds <= 160 * 1024, "LDS budget (NW = 4: two workgroups per CU)");
    const int nqb = (a.Sp + 32 * NW - 1) / (32 * NW), bh8 = (a.B * a.nh + 7) / 8 * 8;
    auto go = [&](auto kern, std::atomic<unsigned>& r) -> const char* {
        if (!glc_raise_lds_limit(kern, (int)lds, r)) return "attention(mx): cannot raise the dynamic LDS limit";
        hipLaunchKernelGGL(kern, dim3(nqb * bh8), dim3(64 * NW), lds, st, a);
        return nullptr;
    };
    static std::atomic<unsigned> r0{0};
#ifdef GLC_DEVELOPER      // stamped, timing-only (WRONG results) and measurement builds: developer libraries only
    static std::atomic<unsigned> r1{0}, r2{0}, r4{0}, r5{0}, r6{0}, r7{0};
    if (a.stamps) return go(attn_mx_kernel<NW, 0, true, true>, r4);
    if (a.variant & 256) return go(attn_mx_kernel<NW, 1>, r1);
    if (a.variant & 512) return go(attn_mx_kernel<NW, 2>, r2);
    static const bool pv16_env = glc_dev_env("GLC_ATTN_PV16") && atoi(glc_dev_env("GLC_ATTN_PV16")) != 0;
    if ((a.variant & 4096) || pv16_env) return go(attn_mx_kernel<NW, 3>, r5);
    if (a.variant & 65536) return go(attn_mx_kernel<NW, 4, false, true>, r7);    // bit 16: timing only, f16 MFMAs in the 16x16x32 shape
    static std::atomic<unsigned> r8{0};
    static std::atomic<unsigned> r9{0};
    if (a.variant & 262144) return go(attn_mx_kernel<NW, 0, false, true, true, false>, r9);     // bit 18: resident blocks, the extra block on wave t mod NW (A/B)
    if (a.variant & 131072) return go(attn_mx_kernel<NW, 0, false, true, false, false, false>, r8);     // bit 17: PQ rows requested every key tile (round 4's form; A/B)
    static std::atomic<unsigned> r10{0};
    if (a.variant & 1048576) return go(attn_mx_kernel<NW, 0, false, true, true, true, false>, r10);     // bit 20: round 5's look-ups (clamped index, offsets derived per request, 32-byte MX steps) and ascending ring (A/B)
    if (a.variant & 16384) return go(attn_mx_kernel<NW, 0, false, false>, r6);     // bit 14: the odd-step gather addresses as spilled loop invariants (round 3's build; A/B)
#else
    if (a.stamps || (a.variant & (256 | 512 | 4096 | 16384 | 65536 | 131072 | 262144 | 1048576))) return "attention(mx): stamped, timing-only and measurement builds exist in developer builds only (make DEV=1)";
#endif
    return go(attn_mx_kernel<NW, 0, false, true>, r0);
}
}  // namespace

const char* glc_launch_attention_mx(hipStream_t st, const AttnArgs& a_in) {
    AttnArgs a = a_in;
    if (!a.gx_sat) a.gx_sat = glc_gx_sat_ptr();              // fp8 range guard of the GX context rows
    if (!a.act_sc) a.act_sc = glc_gx_act_sc();               // ... and the exponent of the activation rows (engine.hip act_sc)
    if (!a.Qh || !a.Kh || !a.Vt || !a.PK || !a.PQ || !a.kbias || !a.klen || !a.kfirst || !a.CTX || !a.otab || !a.mtab) return "attention(mx): null pointer";
    if (a.B <= 0 || a.nh <= 0 || a.Sp <= 0 || a.Sp % 64 || a.H != a.nh * 64 || a.P <= 0 || a.P % 32) return "attention(mx): bad shape";
    if (a.sel_b || a.tile_flag) return "attention(mx): no row selection in this kernel";
    // workgroup shape (kernel header): 4 waves x two workgroups per CU by default; GLC_ATTN_MX_NW=8 or AttnArgs::variant bit 11: 8 waves x one (bit 10: 4)
    static const int nw_env = glc_dev_env("GLC_ATTN_MX_NW") ? atoi(glc_dev_env("GLC_ATTN_MX_NW")) : 4;
    const int nw = (a.variant & 1024) ? 4 : (a.variant & 2048) ? 8 : nw_env;     // (bits 10 / 11 choose the workgroup shape and nothing else)
    return nw == 8 ? launch_mx<8>(st, a) : launch_mx<4>(st, a);
}

// nrows rows (a multiple of 32) x 64 columns x nheads tensors in split units -> MX tiles (position tables at load)
const char* glc_launch_units_to_mxt(hipStream_t st, const void* src, void* dst, int ntiles, int hl, unsigned* sat, int planar) {
    if (!src || !dst || ntiles <= 0) return "units_to_mxt: bad args";
    hipLaunchKernelGGL(units_to_mxt_kernel, dim3(ntiles), dim3(64), 0, st, (const unsigned char*)src, (unsigned char*)dst, ntiles, hl, sat, planar);
    return nullptr;
}
