// 256x256 GEMM of the default mode with MX cross terms (round 3): C = A . W^T on "GX" operand rows (glc_common.h):
//   a * w = a_hi * w_hi            two v_mfma_f32_32x32x16_f16 per 32 k and 32x32 block            (2 x 32 cycles)
//         + a_hi * w_lo + a_lo * w_hi   ONE v_mfma_scale_f32_32x32x64_f8f6f4 on the fp8 parts         (64 cycles)
// instead of the six f16 MFMAs (192 cycles) of the split-f16 kernel (gemm256s.hip, GS).  Measured on the instruction mix alone
// (scripts/probes/mx32_probe.hip): 128 vs 192 matrix-pipe cycles per product and a clock the chip holds 6 % higher (1.77 vs 1.67 GHz);
// the round-3 stamps of the GS main loop showed its matrix pipe ~85 % busy, i.e. the pipe, not the loads, is what this removes.
// Operand error: the cross terms are ~2^-11 of a product and come out to ~4 bits, so a product is good to ~2^-15 relative
// (single f16: 2^-11; full split: 2^-21) — DESIGN.md has the measured per-label probability error of the whole forward.
//
// Structure = the full-line ring of gemm256s.hip (FL): ring stage 2s = A rows of 32-group s, stage 2s + 1 = W rows (one 128-byte GX
// group per row: 8 whole lines per LDS-DMA wave-instruction), slot = stage & 3, LDS image [256 rows][128 B] with the 16-byte chunk
// swizzle c ^ ((row >> 1) & 7); the same two-phase step and wave-group stagger, with EQUAL halves:
//   E: phase A = request group s + 1 (8 pieces per wave), read the f16 fragments (8 + 4); phase B = 16 x 32x32x16 f16      (512 cycles)
//   O: phase A = read the fp8 fragments (4 + 2 operands of 32 B), wait for my pieces;     phase B =  8 x 32x32x64 scaled   (512 cycles)
// Hazards: as argued in gemm256s.hip (FL).  Fragment maps: f16 32x32x16 lane (c = l & 31, h = l >> 5) holds row c, k = 16 ks + 8 h + j
// = chunk 2 ks + h of the group; fp8 32x32x64 lane (c, h) holds bytes [0,16) = slots 16 h .. 16 h + 15 of MX block 0 and [16,32) = the
// same slots of block 1 (probe: byte y <-> k-slot 16 h + (y & 15) + 32 (y >> 4)); the scale of block b of row c is taken from lane
// c + 32 b.  The lane's 32 bytes are chunks 4 + 2 h and 5 + 2 h of the group — the fp8 parts of elements 16 h .. 16 h + 15, A rows as
// [lo8 x 8 | hi8 x 8] per 8 elements, W rows as [hi8 x 8 | lo8 x 8] — so slot by slot a_lo8 meets w_hi8 and a_hi8 meets w_lo8 over
// the same element, and every slot's product carries 2^-(SHIFT + ws): one scale per operand for all blocks (glc_common.h).
// Accumulators are 32x32 blocks, acc[I][J]: non-transposed launches D[n][m] (lane = m, registers = n: 4 consecutive n per register
// quad), the V third D[m][n].  Epilogues as gemm256s.hip (LDS-staged 16-byte stores, LayerNorm fold, residual prefetch), reading and
// writing GX rows where that kernel has GS rows.
#include <stdlib.h>
#include <type_traits>
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int LINE = 128;                  // bytes per row and group
constexpr int STAGE = TM * LINE;           // 32 KiB: one operand's rows of one group
constexpr int NSLOT = 4;
constexpr int EPI_PATCH = 9216;            // bytes of wave-private fp32 epilogue staging
extern __shared__ __attribute__((aligned(16))) unsigned char smem256x[];
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#ifndef GLC_GX_A_AUX
#define GLC_GX_A_AUX 0
#endif
#ifndef GLC_GX_W_AUX
#define GLC_GX_W_AUX 0
#endif



// AUX: the cache-policy bits of the request (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)l, 16, 0, AUX);
}

// XCD-aware tile order (gemm256s.hip): this workgroup's (M-tile, N-tile) of a launch over ntn N-tiles
__device__ __forceinline__ void x_tile_of_block(const GemmArgs& p, int ntn, int& mt, int& nt) {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    mt = tile / ntn; nt = tile % ntn;
    if (p.n_group > 0) {
        const int mts = nwg / ntn, mpx = mts >> 3, nb = p.n_group;
        const int i = bid >> 3, per = mpx * nb;
        const int cg = i / per, r = i - cg * per;
        mt = xcd * mpx + r / nb;
        nt = cg * nb + r % nb;
    }
}

// W128 (round 6; the launcher takes it for the large bias / GELU / SwiGLU / residual shapes): the same tile by FOUR waves, one per SIMD with 512 registers (the
// 256 accumulator registers in AGPRs — hipcc's default; -amdgpu-mfma-vgpr-form would fill the loop with accumulator copies), each a 128 x 128 sub-tile = 16 blocks.  One in-order wave overlaps its MFMAs with its own loads when they alternate in program
// order (profiles/r06/overlap_bisect.txt): per 32-group the wave issues F = 32 x 32x32x16 f16 with the group's fp8 fragment reads between them, then X = 16 scaled
// 32x32x64 with the NEXT group's f16 fragment reads and its 16 DMA pieces of the group after that between them — every fragment register is loaded one phase
// (1024 matrix-pipe cycles) before its use and dies with its phase, so no fragment is double-buffered (a16 / w16 / xa / xw: 128 registers), ONE barrier per group.
// LDS reads per group and CU 128 KiB instead of 192; requests enter the vector-memory path one per MFMA gap instead of in bursts.
template <int EPI, bool VMODE, bool W128 = false>
__device__ __forceinline__ void gemm256x_tile(const GemmArgs& p, int n_tile0, int ntn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm_outer = W128 ? (wave >> 1) : (wave >> 2), wn_outer = W128 ? (wave & 1) : (wave & 3);
    const int c32 = lane & 31, h = lane >> 5;
    const int K = p.K, N = p.N;

    int mt, nt;
    x_tile_of_block(p, ntn, mt, nt);
    const int m0 = mt * TM, n0 = (n_tile0 + nt) * TN;
    if constexpr (EPI == EPI_QKV && !VMODE) {
        if (p.q_tile_flag && n0 < p.H) {                // Q third, pruned last layer: nobody reads query tiles without selected rows
            const unsigned long long f8 = *reinterpret_cast<const unsigned long long*>(p.q_tile_flag + (m0 >> 5));
            if (f8 == 0ull) return;
        }
    }

    const unsigned char* __restrict__ A = reinterpret_cast<const unsigned char*>(p.A);
    const unsigned char* __restrict__ W = reinterpret_cast<const unsigned char*>(p.W);
    const size_t rsb = (size_t)4 * K;          // row stride in bytes (GX rows)
    const int ng = K / 32;
    // DMA map (FL): lane L lands at (row L >> 3, physical chunk L & 7) of an 8-row piece and fetches logical chunk (L & 7) ^ ((row >> 1) & 7)
    const int lrow8 = lane >> 3, pch = lane & 7;
    const unsigned char* fa[2];
    const unsigned char* fw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * (W128 ? 64 : 32) + i * 8 + lrow8;
        const int lc = pch ^ ((row >> 1) & 7);
        fa[i] = A + (size_t)(m0 + row) * rsb + lc * 16;
        fw[i] = W + (size_t)(n0 + row) * rsb + lc * 16;
    }
    auto stage_a = [&](int grp) {           // this wave's 32 A rows of group grp: 4 one-KiB pieces
        unsigned char* sa = smem256x + ((2 * grp) & (NSLOT - 1)) * STAGE + (wave * 32) * LINE;
        const size_t o = (size_t)grp * LINE;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16<GLC_GX_A_AUX>(fa[i & 1] + (size_t)(i >> 1) * 16 * rsb + o, sa + i * 8 * LINE);
    };
    auto stage_w = [&](int grp) {           // ... and its 32 W rows
        unsigned char* sw = smem256x + ((2 * grp + 1) & (NSLOT - 1)) * STAGE + (wave * 32) * LINE;
        const size_t o = (size_t)grp * LINE;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16<GLC_GX_W_AUX>(fw[i & 1] + (size_t)(i >> 1) * 16 * rsb + o, sw + i * 8 * LINE);
    };
    auto stage_fl = [&](int grp) { stage_a(grp); stage_w(grp); };

    f32x16 accs[W128 ? 2 : 1][4][2];          // W128: [column half ch][row block][column block of the half] — the epilogue below runs once per half
#pragma unroll
    for (int ch = 0; ch < (W128 ? 2 : 1); ++ch)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[ch][i][j][r] = 0.f;

    // fragment read offsets inside a slot
    const int hsw = (c32 >> 1) & 7;
    constexpr int LINEF = LINE;
    const int arow = (wm_outer * 128 + c32) * LINEF, wrow = (wn_outer * (W128 ? 128 : 64) + c32) * LINEF;
    const int ck0 = ((0 + h) ^ hsw) * 16, ck1 = ((2 + h) ^ hsw) * 16;       // f16 k-steps 0 / 1: logical chunks h / 2 + h
    const int cx0 = ((4 + 2 * h) ^ hsw) * 16, cx1 = ((5 + 2 * h) ^ hsw) * 16;       // the fp8 parts of elements 16 h .. 16 h + 7 / + 8 .. + 15
    // e8m0 scales (one per operand, every block): A rows (activations, exponent 0) carry the 2^-SHIFT, W rows their 2^-ws
    const int sc_a = 127 - GLC_GX_SHIFT - p.act_sc;        // e8m0 scale of the A blocks: 2^-(SHIFT + sc) (glc_common.h)
    const int sc_w = 127 - p.mx_ws;

    if constexpr (W128) {
        f16x8 a16[4][2], w16[4][2];
        i32x8 xa[4], xw[4];
        auto slot_a = [&](int grp) -> unsigned char* { return smem256x + ((2 * grp) & (NSLOT - 1)) * STAGE; };
        auto slot_w = [&](int grp) -> unsigned char* { return smem256x + ((2 * grp + 1) & (NSLOT - 1)) * STAGE; };
        // piece q = 0 .. 15 of this wave's 64 A rows (q < 8) and 64 W rows of group grp
        // (the request in its scalar-base form: wave-uniform base + a 32-bit lane offset that never changes — no 64-bit address arithmetic per lane and piece)
        const unsigned voffs[2] = {(unsigned)((size_t)lrow8 * rsb + (pch ^ (((wave * 64 + lrow8) >> 1) & 7)) * 16), (unsigned)((size_t)(8 + lrow8) * rsb + (pch ^ (((wave * 64 + 8 + lrow8) >> 1) & 7)) * 16)};
        const unsigned char* const ubA = A + (size_t)(m0 + wave * 64) * rsb;
        const unsigned char* const ubW = W + (size_t)(n0 + wave * 64) * rsb;
        auto dma_piece = [&](int grp, int q) __attribute__((always_inline)) {
            const int i = q & 7;
            const unsigned char* ub = (q < 8 ? ubA : ubW) + (size_t)grp * LINE + (size_t)(i >> 1) * 16 * rsb;
            unsigned char* l = (q < 8 ? slot_a(grp) : slot_w(grp)) + (wave * 64 + i * 8) * LINE;
            const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)l;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la), "v"(voffs[i & 1]), "s"(ub) : "memory");
        };
        // fragment read r = 0 .. 15 of a group: the f16 k-steps (r < 8: W blocks, r >= 8: A blocks; two reads per block) ...
        auto rd_f16 = [&](int grp, int r) __attribute__((always_inline)) {
            const int b = (r & 7) >> 1, ks = r & 1;
            if (r < 8) w16[b][ks] = *reinterpret_cast<const f16x8*>(slot_w(grp) + wrow + b * 32 * LINEF + (ks ? ck1 : ck0));
            else a16[b][ks] = *reinterpret_cast<const f16x8*>(slot_a(grp) + arow + b * 32 * LINEF + (ks ? ck1 : ck0));
        };
        // ... and the fp8 operands (two 16-byte chunks each)
        auto rd_x = [&](int grp, int r) __attribute__((always_inline)) {
            const int b = (r & 7) >> 1, half = r & 1;
            const unsigned char* q0 = (r < 8 ? slot_w(grp) + wrow : slot_a(grp) + arow) + b * 32 * LINE + (half ? cx1 : cx0);
            const i32x4 t = *reinterpret_cast<const i32x4*>(q0);
            i32x8& d = r < 8 ? xw[b] : xa[b];
            d[4 * half] = t[0]; d[4 * half + 1] = t[1]; d[4 * half + 2] = t[2]; d[4 * half + 3] = t[3];
        };
#pragma unroll
        for (int q = 0; q < 16; ++q) dma_piece(0, q);
        if (ng > 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) dma_piece(1, q);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) rd_f16(0, r);
        // one group; M1 / M2 (compile time: the tail is peeled, a branch inside a phase would let hipcc sink the MFMAs behind the loads): groups s + 1 / s + 2 exist
        auto step = [&](const int s, auto m1, auto m2) __attribute__((always_inline)) {
            constexpr bool M1 = decltype(m1)::value, M2 = decltype(m2)::value;
            // ---- F(s): a_hi w_hi of group s; its fp8 fragments are read underneath ----
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int ks = k >> 4, i = (k >> 2) & 3, j = k & 3;
                f32x16& a = accs[j >> 1][i][j & 1];
                if (!VMODE) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(w16[j][ks], a16[i][ks], a, 0, 0, 0);      // D[n][m]
                else a = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[i][ks], w16[j][ks], a, 0, 0, 0);            // D[m][n]
                if ((k & 1) == 0) rd_x(s, k >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // group s + 1 (requested during X(s - 1)) has landed; every wave has read the last of group s: its slots are free
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---- X(s): both cross terms of group s; the f16 fragments of group s + 1 are read and group s + 2 is requested underneath ----
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = k >> 2, j = k & 3;
                f32x16& a = accs[j >> 1][i][j & 1];
                if (!VMODE) a = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xw[j], xa[i], a, 0, 0, 0, sc_w, 0, sc_a);
                else a = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[i], xw[j], a, 0, 0, 0, sc_a, 0, sc_w);
                if constexpr (M1) rd_f16(s + 1, k);
                if constexpr (M2) dma_piece(s + 2, k);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int s = 0; s + 2 < ng; ++s) step(s, std::true_type{}, std::true_type{});
        if (ng > 1) step(ng - 2, std::true_type{}, std::false_type{});
        step(ng - 1, std::false_type{}, std::false_type{});
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // every wave is through its last fragment read: the ring becomes the epilogue's staging space
    } else {
    const int wm = wm_outer;
    f32x16 (&acc)[4][2] = accs[0];
    f16x8 a16[4][2], w16[2][2];
    i32x8 xa[4], xw[2];
    
    
    {
    const int pm = p.prio_mode;     // 0: no priorities; 1: MFMA phase at priority 1; 2: load phase at priority 2; 3: the late wave group at priority 1 throughout
    if (pm == 3 && wm == 1) __builtin_amdgcn_s_setprio(1);
    stage_fl(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // group 0 is in LDS for everyone
    if (wm == 1) __builtin_amdgcn_s_barrier(); // the stagger
    auto ld32 = [&](const unsigned char* q0, int o0, int o1) __attribute__((always_inline)) {      // two 16-byte chunks -> one 32-byte MX operand
        const i32x4 t0 = *reinterpret_cast<const i32x4*>(q0 + o0);
        const i32x4 t1 = *reinterpret_cast<const i32x4*>(q0 + o1);
        i32x8 r;
        r[0] = t0[0]; r[1] = t0[1]; r[2] = t0[2]; r[3] = t0[3]; r[4] = t1[0]; r[5] = t1[1]; r[6] = t1[2]; r[7] = t1[3];
        return r;
    };
    auto sub = [&](const int s, const int odd) __attribute__((always_inline)) {
        // ---- phase A ----
        if (pm == 2) __builtin_amdgcn_s_setprio(2);
        if (!odd && s + 1 < ng && 0 != 5) stage_fl(s + 1);      // (ABL 4 / 5 / 6: timing-only builds — no fragment reads / no DMA / no MFMAs; wrong results)
        {
            const unsigned char* sa = smem256x + ((2 * s) & (NSLOT - 1)) * STAGE + arow;
            const unsigned char* sw = smem256x + ((2 * s + 1) & (NSLOT - 1)) * STAGE + wrow;
            if (!odd) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    w16[j][0] = *reinterpret_cast<const f16x8*>(sw + j * 32 * LINEF + ck0);
                    w16[j][1] = *reinterpret_cast<const f16x8*>(sw + j * 32 * LINEF + ck1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a16[i][0] = *reinterpret_cast<const f16x8*>(sa + i * 32 * LINEF + ck0);
                    a16[i][1] = *reinterpret_cast<const f16x8*>(sa + i * 32 * LINEF + ck1);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) xw[j] = ld32(sw + j * 32 * LINE, cx0, cx1);      // [w_hi8 w_lo8 | w_hi8 w_lo8] of 2 x 8 elements
#pragma unroll
                for (int i = 0; i < 4; ++i) xa[i] = ld32(sa + i * 32 * LINE, cx0, cx1);      // [a_lo8 a_hi8 | a_lo8 a_hi8]
            }
        }
        if (odd) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (pm == 2) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase B ----
        if (pm == 1) __builtin_amdgcn_s_setprio(1);
        if (!odd) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        
                        if (!VMODE) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w16[j][ks], a16[i][ks], acc[i][j], 0, 0, 0);      // D[n][m]
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[i][ks], w16[j][ks], acc[i][j], 0, 0, 0);            // D[m][n]
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!VMODE) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xw[j], xa[i], acc[i][j], 0, 0, 0, sc_w, 0, sc_a);
                    else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[i], xw[j], acc[i][j], 0, 0, 0, sc_a, 0, sc_w);
                }
        }
        if (pm == 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    for (int s = 0; s < ng; ++s) { sub(s, 0); sub(s, 1); }
    if (wm == 0) __builtin_amdgcn_s_barrier();   // pairs with the late group's last barrier
    }
    }

    // ---------------- epilogue ----------------
    // (W128: the wave's 128 x 128 as two 128 x 64 halves through the code of the 8-wave tiles — wave (wm, wn) there = (wave >> 1, 2 (wave & 1) + ch) here)
#pragma unroll
    for (int ch = 0; ch < (W128 ? 2 : 1); ++ch) {
    const int wm = wm_outer, wn = W128 ? 2 * wn_outer + ch : wn_outer;
    f32x16 (&acc)[4][2] = accs[ch];
    typedef f16_t T;
    typedef __attribute__((ext_vector_type(8))) T vec8T;
    const float* __restrict__ bias = p.bias;
    float* stg = reinterpret_cast<float*>(smem256x + wave * EPI_PATCH);
    const int qkv_b0 = (EPI == EPI_QKV || EPI == EPI_QKVR) ? m0 / p.Sp : 0;
    const float kHi = gx_act_khi(p.act_sc), kLo = gx_act_klo(p.act_sc), kInvLo = gx_pow2_inv(kLo);       // activation rows in and out: exponent act_sc
    constexpr float kInvLo0 = 1.0f / (float)(1 << GLC_GX_SHIFT);                                          // MX tiles (attention operands): exponent 0       // activation rows: exponent 0
    if constexpr (EPI == EPI_SWIGLU) {
        // W rows alternate 16 gate / 16 up features (engine.hip interleaves them at load): in D[n = 32 J + 8 q + 4 h + e][m] the register quads
        // q = 0, 1 hold gate features 8 q + 4 h + e of block J and q + 2 the matching up features — same lane, no exchange.  The wave's
        // 128 x 64 sub-tile becomes 128 x 32 outputs silu(gate) * up (Q2:47); patch [32 rows][32 features], row stride 36 floats.
        // RMSNorm folded into this GEMM (a_stats: W holds W diag(gain), the rows are raw): gate and up scale by the row's rstd first.
        const int Iw = N >> 1;
        const bool lnf = p.a_stats != nullptr;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float rs = lnf ? p.a_stats[m0 + wm * 128 + c * 32 + c32].y : 1.0f;
            
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gt = acc[c][J][4 * q + e] * rs, up = acc[c][J][4 * (q + 2) + e] * rs;
                        v[e] = gt * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt)) * up;
                    }
                    *reinterpret_cast<f32x4*>(stg + c32 * 36 + 16 * J + 8 * q + 4 * h) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = lane + 64 * k, row = idx >> 2, g4 = idx & 3;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * 36 + g4 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * 36 + g4 * 8 + 4);
                const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const int m = m0 + wm * 128 + c * 32 + row;
                gx_store8(reinterpret_cast<unsigned char*>(p.C) + (size_t)m * 4 * Iw, (n0 >> 1) + wn * 32 + g4 * 8, v, kHi, kLo, m < p.gx_rows ? p.gx_sat : nullptr);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    } else if constexpr (!VMODE) {
        // D[n = 32 J + 8 q + 4 h + e][m = 32 I + c32]; patch [32 rows m][64 cols n], row stride 68 floats
        const int which = (EPI == EPI_QKV) ? n0 / p.H : 0;
        const bool lnf = EPI != EPI_RESID && p.a_stats != nullptr;
        f32x4 bj[2][4], cj[2][4];
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nn = n0 + wn * 64 + 32 * J + 8 * q + 4 * h;      // Z16: entries q = 0 / 2 = columns 16 (2 J + q / 2) + 4 qz ..
                bj[J][q] = bias ? *reinterpret_cast<const f32x4*>(bias + nn) : (f32x4){0.f, 0.f, 0.f, 0.f};
                cj[J][q] = (lnf && p.ln_c) ? *reinterpret_cast<const f32x4*>(p.ln_c + nn) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        float rg[8], rb[8];
        const bool rln = EPI == EPI_RESID && p.r_stats != nullptr;
        const bool gxout = EPI == EPI_RESID && p.ln_part != nullptr;      // raw GX rows + statistics partials out
        if constexpr (EPI == EPI_RESID) {
            if (rln) {
                const int nb = n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { rg[e] = p.r_gamma[nb + e]; rb[e] = p.r_beta[nb + e]; }
            }
        }
        // residual rows (GX) one 32-row chunk ahead of their use: lane = 8 consecutive columns
        gs_h8 rpre[4]; u32x2 rpre_lo[4]; float2 rst_pre[4];
        auto load_resid = [&](int c, gs_h8 (&r)[4], u32x2 (&rl)[4], float2 (&rst)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, row = idx >> 3, g8 = idx & 7;
                if (rln) rst[k] = p.r_stats[m0 + wm * 128 + c * 32 + row];
                const int n = n0 + wn * 64 + g8 * 8;
                const unsigned char* rp = reinterpret_cast<const unsigned char*>(p.resid) + (size_t)(m0 + wm * 128 + c * 32 + row) * 4 * N + (n >> 5) * 128;
                r[k] = *reinterpret_cast<const gs_h8*>(rp + (n & 31) * 2);
                rl[k] = *reinterpret_cast<const u32x2*>(rp + 64 + (n & 31) * 2);
            }
        };
        if constexpr (EPI == EPI_RESID) { load_resid(0, rpre, rpre_lo, rst_pre); }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            gs_h8 rcur[4]; u32x2 rcur_lo[4]; float2 rst_cur[4];
            
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { rcur[k] = rpre[k]; rcur_lo[k] = rpre_lo[k]; rst_cur[k] = rst_pre[k]; }
                if (c + 1 < 4) load_resid(c + 1, rpre, rpre_lo, rst_pre);
            }
            const float2 sm = lnf ? p.a_stats[m0 + wm * 128 + c * 32 + c32] : make_float2(0.f, 1.f);
            
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[c][J][4 * q], acc[c][J][4 * q + 1], acc[c][J][4 * q + 2], acc[c][J][4 * q + 3]};
                    if constexpr (EPI != EPI_RESID) {
                        if (lnf) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = sm.y * (v[r] - sm.x * cj[J][q][r]);
                        }
                    }
                    v += bj[J][q];
                    if (EPI == EPI_GELU) { const f32x2 g0 = glc_gelu2_f32((f32x2){v[0], v[1]}), g1 = glc_gelu2_f32((f32x2){v[2], v[3]}); v = (f32x4){g0[0], g0[1], g1[0], g1[1]}; }
                    *reinterpret_cast<f32x4*>(stg + c32 * 68 + 32 * J + 8 * q + 4 * h) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if constexpr (EPI == EPI_QKVR) {
                // Decoder Q / K heads (head_dim 128): this wave's 64 columns are features [32 hf, 32 hf + 32) (patch columns 0 .. 31) and their
                // rotate-half partners 64 + the same (columns 32 .. 63) of one head (W rows in glc_rope_perm128 order).  A lane takes 8
                // consecutive pairs of one row: RoPE (Q2:211) and, on Q, the softmax scale in fp32, then the two 8-value pieces of the MX
                // tile (decoder_mx.hip layout; Q as (hi8 | lo8) at slot r, K as (lo8 | hi8) at slot pi(r)).
                const int col0 = n0 + wn * 64, head = col0 >> 7, hf = (col0 >> 6) & 1;
                const bool isq = head < p.nq;
                const int ntl = p.Sp >> 5;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int idx = lane + 64 * k, row = idx >> 2, g4 = idx & 3;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + row * 68 + g4 * 8), a1 = *reinterpret_cast<const f32x4*>(stg + row * 68 + g4 * 8 + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(stg + row * 68 + 32 + g4 * 8), b1 = *reinterpret_cast<const f32x4*>(stg + row * 68 + 32 + g4 * 8 + 4);
                    const int m = m0 + wm * 128 + c * 32 + row;
                    if (m >= p.Mvalid) continue;
                    int b = qkv_b0, sq = m - qkv_b0 * p.Sp;
                    while (sq >= p.Sp) { sq -= p.Sp; ++b; }
                    const float x1[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, x2[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    const int d1 = 32 * hf + g4 * 8, d2 = d1 + 64;
                    const f32x4* cp = reinterpret_cast<const f32x4*>(p.rope_cs + ((size_t)sq * 64 + d1) * 2);
                    const f32x4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
                    const float cs[16] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3], c2[0], c2[1], c2[2], c2[3], c3[0], c3[1], c3[2], c3[3]};
                    const float sc = isq ? p.qscale : 1.f;
                    float o1[8], o2[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float co = cs[2 * j], sn = cs[2 * j + 1];
                        o1[j] = (x1[j] * co - x2[j] * sn) * sc;
                        o2[j] = (x2[j] * co + x1[j] * sn) * sc;
                    }
                    const int r = sq & 31, slot = isq ? r : glc_pi32(r);
                    unsigned char* base = isq ? reinterpret_cast<unsigned char*>(p.Qh) + ((size_t)(b * p.nq + head) * ntl + (sq >> 5)) * 16384
                                              : reinterpret_cast<unsigned char*>(p.Kh) + ((size_t)(b * p.nkv + (head - p.nq)) * ntl + (sq >> 5)) * 16384;
                    unsigned* sat = m < p.gx_rows ? p.gx_sat : nullptr;
                    store_mx8(base + (d1 >> 4) * 1024 + (32 * ((d1 >> 3) & 1) + slot) * 16, base + 8192 + (d1 >> 5) * 2048 + (32 * ((d1 >> 4) & 1) + slot) * 32 + 8 * ((d1 >> 3) & 1), o1, isq, sat);
                    store_mx8(base + (d2 >> 4) * 1024 + (32 * ((d2 >> 3) & 1) + slot) * 16, base + 8192 + (d2 >> 5) * 2048 + (32 * ((d2 >> 4) & 1) + slot) * 32 + 8 * ((d2 >> 3) & 1), o2, isq, sat);
                }
            } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, row = idx >> 3, g8 = idx & 7;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * 68 + g8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * 68 + g8 * 8 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const int m = m0 + wm * 128 + c * 32 + row;
                const int n = n0 + wn * 64 + g8 * 8;
                if constexpr (EPI == EPI_RESID) {
                    float r[8];
                    gx_decode8(rcur[k], rcur_lo[k], kInvLo, r);
                    if (rln) {           // raw residual row: LayerNorm on the fly
                        const float2 rs = rst_cur[k];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (r[e] - rs.x) * rs.y * rg[e] + rb[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
                    if (gxout) {
                        // raw GX row out + this 64-column block's (sum, squared deviations from the block mean) of the row (gemm256s.hip)
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) s1 += v[e];
#pragma unroll
                        for (int o = 1; o < 8; o <<= 1) s1 += __shfl_xor(s1, o, 64);
                        const float bm = s1 * (1.0f / 64.0f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float dv = v[e] - bm; s2 += dv * dv; }
#pragma unroll
                        for (int o = 1; o < 8; o <<= 1) s2 += __shfl_xor(s2, o, 64);
                        if (g8 == 0) p.ln_part[(size_t)m * (N >> 6) + ((n0 + wn * 64) >> 6)] = make_float2(s1, s2);
                        gx_store8(reinterpret_cast<unsigned char*>(p.C) + (size_t)m * 4 * N, n, v, kHi, kLo, m < p.gx_rows ? p.gx_sat : nullptr);
                    } else {             // plain fp32 row (LayerNorm input)
                        float* cp = reinterpret_cast<float*>(p.C) + (size_t)m * N + n;
                        *reinterpret_cast<f32x4*>(cp) = (f32x4){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    }
                } else if constexpr (EPI == EPI_QKV) {
                    if (m < p.Mvalid) {
                        vec8T o, ol;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { o[e] = (T)v[e]; ol[e] = (T)(v[e] - (float)o[e]); }
                        int b = qkv_b0, sq = m - qkv_b0 * p.Sp;
                        while (sq >= p.Sp) { sq -= p.Sp; ++b; }
                        const int nn = n - which * p.H, hh = nn >> 6, dd = nn & 63;
                        const int bh = b * p.nh + hh;
                        if (p.qkv_mxt) {        // MX tiles (glc_layout.h): f16 unit piece + the fp8 parts, Q as (hi8 | lo8), K as (lo8 | hi8)
                            gx_range_note(v, 1.0f, p.gx_sat && m < p.gx_rows ? p.gx_sat + 1 : nullptr);      // (tiles: the guard's second word; no counter outside a GxScope, padding rows not counted)
                            u32x2 l8, h8;
                            gs_h8 oh;
                            gx_split8(v, 1.0f, kInvLo0, oh, l8, h8);
                            const int tile = bh * (p.Sp >> 5) + (sq >> 5), slot = which == 0 ? (sq & 31) : glc_pi32(sq & 31);
                            unsigned char* bq = reinterpret_cast<unsigned char*>(which == 0 ? p.Qh : p.Kh);
                            unsigned char* px = bq + glc_mxt_mx(tile, slot, dd);
                            *reinterpret_cast<vec8T*>(bq + glc_mxt_f16(tile, slot, dd)) = o;
                            *reinterpret_cast<u32x2*>(px) = which == 0 ? h8 : l8;
                            *reinterpret_cast<u32x2*>(px + 16) = which == 0 ? l8 : h8;
                            continue;
                        }
                        const size_t off = which == 0 ? glc_qoff(p.Sp, bh, sq, dd) : glc_koff(p.Sp, bh, sq, dd);
                        T* base = reinterpret_cast<T*>(which == 0 ? p.Qh : p.Kh);
                        *reinterpret_cast<vec8T*>(base + 2 * off) = o;          // split-f16 unit [8 hi | 8 lo]
                        *reinterpret_cast<vec8T*>(base + 2 * off + 8) = ol;
                    }
                } else {
                    if (p.gs_c_plain) {
                        const int nl = n < p.perm_cols ? (n & ~127) | glc_rope_perm128(n & 127) : n;      // (W rows in the EPI_QKVR order: back to the logical column)
                        float* cp = reinterpret_cast<float*>(p.C) + (size_t)m * N + nl;
                        *reinterpret_cast<f32x4*>(cp) = (f32x4){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    } else
                    gx_store8<false, true>(reinterpret_cast<unsigned char*>(p.C) + (size_t)m * 4 * N, n, v, kHi, kLo, m < p.gx_rows ? p.gx_sat : nullptr);      // FFN1's intermediate: streams (non-temporal)
                }
            }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // V third: D[m = 32 I + 8 q + 4 h + e][n = 32 J + c32]; patch [64 rows dd][32 cols key], row stride 36 floats
        float bn[2], cn[2] = {0.f, 0.f};
        const bool lnf = p.a_stats != nullptr;
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            bn[J] = bias ? bias[n0 + wn * 64 + 32 * J + c32] : 0.f;
            if (lnf && p.ln_c) cn[J] = p.ln_c[n0 + wn * 64 + 32 * J + c32];
        }
        const int hh = (n0 + wn * 64 - 2 * p.H) >> 6;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[c][J][4 * q], acc[c][J][4 * q + 1], acc[c][J][4 * q + 2], acc[c][J][4 * q + 3]};
                    if (lnf) {      // accumulator rows m0 + 128 wm + 32 c + 8 q + 4 h + r
                        const float2* sp = p.a_stats + m0 + wm * 128 + c * 32 + 8 * q + 4 * h;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float2 sm = sp[r]; v[r] = sm.y * (v[r] - sm.x * cn[J]); }
                    }
                    v[0] += bn[J]; v[1] += bn[J]; v[2] += bn[J]; v[3] += bn[J];
                    *reinterpret_cast<f32x4*>(stg + (32 * J + c32) * 36 + 8 * q + 4 * h) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, dd = idx >> 2, kg = idx & 3;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + dd * 36 + kg * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + dd * 36 + kg * 8 + 4);
                vec8T o, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (T)lo[e]; o[4 + e] = (T)hi[e];
                    ol[e] = (T)(lo[e] - (float)o[e]); ol[4 + e] = (T)(hi[e] - (float)o[4 + e]);
                }
                const int m = m0 + wm * 128 + c * 32 + kg * 8;           // first of 8 consecutive keys
                if (m < p.Mvalid) {
                    int b = qkv_b0, sq = m - qkv_b0 * p.Sp;
                    while (sq >= p.Sp) { sq -= p.Sp; ++b; }
                    if constexpr (EPI == EPI_QKVR) {      // decoder V^T MX tiles (decoder_mx.hip): D / 32 sub-tiles of 4 KiB per 32-key tile, (lo8 | hi8)
                        const float x8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const int col0 = n0 + wn * 64, g = (col0 >> 7) - p.nq - p.nkv, ddl = 64 * ((col0 >> 6) & 1) + dd, kgt = (sq & 31) >> 3;
                        unsigned char* sub = reinterpret_cast<unsigned char*>(p.Vt) + ((size_t)(b * p.nkv + g) * (p.Sp >> 5) + (sq >> 5)) * 16384 + (ddl >> 5) * 4096;
                        store_mx8(sub + (kgt >> 1) * 1024 + (32 * (kgt & 1) + (ddl & 31)) * 16, sub + 2048 + (32 * (kgt & 1) + (ddl & 31)) * 32 + 8 * (kgt >> 1), x8, false,
                                  m < p.gx_rows ? p.gx_sat : nullptr);
                        continue;
                    }
                    if (p.qkv_mxt) {            // V^T MX tiles: (lo8 | hi8)
                        const float x8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        gx_range_note(x8, 1.0f, p.gx_sat && m < p.gx_rows ? p.gx_sat + 1 : nullptr);
                        u32x2 l8, h8;
                        gs_h8 oh;
                        gx_split8(x8, 1.0f, kInvLo0, oh, l8, h8);
                        const int tile = (b * p.nh + hh) * (p.Sp >> 5) + (sq >> 5);
                        unsigned char* bv = reinterpret_cast<unsigned char*>(p.Vt);
                        unsigned char* px = bv + glc_mxt_v_mx(tile, dd, sq);
                        *reinterpret_cast<vec8T*>(bv + glc_mxt_v_f16(tile, dd, sq)) = o;
                        *reinterpret_cast<u32x2*>(px) = l8;
                        *reinterpret_cast<u32x2*>(px + 16) = h8;
                        continue;
                    }
                    const size_t off = glc_voff(p.Sp, b * p.nh + hh, dd, sq);
                    *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.Vt) + 2 * off) = o;
                    *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.Vt) + 2 * off + 8) = ol;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    }
    
}

template <int EPI, bool VMODE>
__global__ __launch_bounds__(512, 2) void gemm256x_kernel(GemmArgs p, int n_tile0, int ntn) { gemm256x_tile<EPI, VMODE>(p, n_tile0, ntn); }
// four waves, one per SIMD, 128 x 128 per wave (W128 above)
template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm256w_kernel(GemmArgs p, int n_tile0, int ntn) { gemm256x_tile<EPI, false, true>(p, n_tile0, ntn); }
// Decoder QKV with the RoPE / MX-tile epilogue in ONE launch: the N-tiles of the V heads (nt >= nqk) run the transposed tile (c5: 7 + 1 N-tiles x
// 128 M-tiles = 4 full rounds of the chip; as two launches the V heads' 128 workgroups would be a fifth, half-empty round)
__global__ __launch_bounds__(512, 2) void gemm256x_qkvr_kernel(GemmArgs p, int nqk, int ntn) {
    int mt, nt;
    x_tile_of_block(p, ntn, mt, nt);
    if (nt < nqk) gemm256x_tile<EPI_QKVR, false>(p, 0, ntn);
    else gemm256x_tile<EPI_QKVR, true>(p, 0, ntn);
}

template <int EPI, bool VMODE> const char* launch_x(hipStream_t st, const GemmArgs& a, int n_tile0, int ntn) {
    static std::atomic<unsigned> lds_ok{0};
    constexpr int lds_bytes = NSLOT * STAGE;
    if (!glc_raise_lds_limit(gemm256x_kernel<EPI, VMODE>, lds_bytes, lds_ok)) return "gemm256x: cannot raise the dynamic LDS limit";
    const int grid = (a.Mpad / TM) * ntn;
    GemmArgs b = a;
    b.n_group = 0;
    static const int prio_env = glc_dev_env("GLC_GEMM_PRIO") ? atoi(glc_dev_env("GLC_GEMM_PRIO")) : 1;      // developer A/B switch
    b.prio_mode = a.prio_mode >= 0 ? a.prio_mode : prio_env;
    if (ntn >= 8 && (a.Mpad / TM) % 8 == 0) b.n_group = ntn % 4 == 0 ? 4 : (ntn % 3 == 0 ? 3 : 0);      // wide N: as gemm256s.hip
#ifdef GLC_GX_NB
    { constexpr int nbv = (GLC_GX_NB) > 0 ? (GLC_GX_NB) : 1;          // build-time A/B of the tile order (scripts/gemm_cache_policy_ab.sh; profiles/r05/gemm_cache_policy.txt)
      if ((a.Mpad / TM) % 8 == 0) b.n_group = (GLC_GX_NB) > 0 && ntn % nbv == 0 ? nbv : 0; }
#endif
    // Wave-tile choice (round 6, measured in the forwards of c3 / c4 / c5: profiles/r06/gemm_w128.txt): the one-wave-per-SIMD 128 x 128 tile reads a third less
    // LDS per MAC and pays its own request issue; it wins where the K loop and the tile count amortise its double epilogue — N K >= 768 x 3072 (FFN1 / FFN2
    // of every encoder shape: -2...-4 %; the decoder's gate|up, down and o_proj: -4...-12 %) — and loses below (attn-out of base / large: +1...+3 %) and on the
    // QKV epilogues (more registers than it has left: spills).  GLC_GEMM_W128 = 0 / 1 (developer builds) forces the 8-wave / the one-wave tile.
    if constexpr (!VMODE && (EPI == EPI_BIAS || EPI == EPI_GELU || EPI == EPI_RESID || EPI == EPI_SWIGLU)) {
        static const int w128_env = glc_dev_env("GLC_GEMM_W128") ? atoi(glc_dev_env("GLC_GEMM_W128")) : -1;
        const bool big = (long long)a.N * a.K >= 768LL * 3072LL;
        if (w128_env == 1 || (w128_env < 0 && big)) {
            static std::atomic<unsigned> lds_okw{0};
            if (!glc_raise_lds_limit(gemm256w_kernel<EPI>, lds_bytes, lds_okw)) return "gemm256x: cannot raise the dynamic LDS limit";
            hipLaunchKernelGGL((gemm256w_kernel<EPI>), dim3(grid), dim3(256), lds_bytes, st, b, n_tile0, ntn);
            return nullptr;
        }
    }
    hipLaunchKernelGGL((gemm256x_kernel<EPI, VMODE>), dim3(grid), dim3(512), lds_bytes, st, b, n_tile0, ntn);
    return nullptr;
}

// fp32 values -> GX rows in place (weights at load): exponent sc on the fp8 parts
__global__ __launch_bounds__(256) void to_gx_kernel(float* __restrict__ w, size_t ngroups, float k_hi, float k_lo, int worder) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    float* base = w + gi * 32;
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(base + 4 * i);      // the whole group is read before it is overwritten
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x[8] = {v[2 * i][0], v[2 * i][1], v[2 * i][2], v[2 * i][3], v[2 * i + 1][0], v[2 * i + 1][1], v[2 * i + 1][2], v[2 * i + 1][3]};
        if (worder) gx_store8<true>(reinterpret_cast<unsigned char*>(base), 8 * i, x, k_hi, k_lo);
        else gx_store8<false>(reinterpret_cast<unsigned char*>(base), 8 * i, x, k_hi, k_lo);
    }
}

// group-split rows ([32 hi | 32 lo] f16 halves per 32 values) -> the largest |hi + lo| (as float bits: non-negative floats order like integers)
__global__ __launch_bounds__(256) void gs_absmax_kernel(const f16_t* __restrict__ gs, size_t ngroups, unsigned* __restrict__ out) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    float m = 0.f;
    if (gi < ngroups) {
        const gs_h8* p = reinterpret_cast<const gs_h8*>(gs + gi * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const gs_h8 hi = p[i], lo = p[4 + i];
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)hi[e] + (float)lo[e]));
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
// group-split rows -> GX rows (weight order, fp8 exponent sc): the MX copies of the projection weights from their split-f16 copies on the device
__global__ __launch_bounds__(256) void gs_to_gx_kernel(const f16_t* __restrict__ gs, unsigned char* __restrict__ gx, size_t ngroups, float k_hi, float k_lo) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    const gs_h8* p = reinterpret_cast<const gs_h8*>(gs + gi * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const gs_h8 hi = p[i], lo = p[4 + i];
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)hi[e] + (float)lo[e];
        gx_store8<true>(gx + gi * 128, 8 * i, x, k_hi, k_lo);
    }
}

}  // namespace

// n values (n % 32 == 0) as group-split rows: *d_bits (zeroed by the caller) = float bits of the largest magnitude
const char* glc_launch_gs_absmax(hipStream_t st, const void* gs, size_t n, unsigned* d_bits) {
    if (!gs || !d_bits || n % 32) return "gs_absmax: bad args";
    const size_t groups = n / 32;
    if (groups) hipLaunchKernelGGL(gs_absmax_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, (const f16_t*)gs, groups, d_bits);
    return nullptr;
}
// ... -> GX rows in weight order with fp8 exponent sc (what glc_launch_to_gx(.., sc, 1) makes of the fp32 values)
const char* glc_launch_gs_to_gx(hipStream_t st, const void* gs, void* gx, size_t n, int sc) {
    if (!gs || !gx || n % 32) return "gs_to_gx: bad args";
    if (sc < -40 || sc > 60) return "gs_to_gx: exponent out of range";
    const size_t groups = n / 32;
    if (groups) hipLaunchKernelGGL(gs_to_gx_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, (const f16_t*)gs, (unsigned char*)gx, groups, ldexpf(1.0f, sc), ldexpf(1.0f, sc + GLC_GX_SHIFT));
    return nullptr;
}

bool glc_gemm256x_supported(const GemmArgs& a, int epi) {
    if (!(a.Mpad > 0 && a.Mpad % TM == 0 && a.N > 0 && a.N % TN == 0 && a.K > 0 && a.K % 32 == 0)) return false;
    if (a.mx_ws < -40 || a.mx_ws > 60) return false;
    if (epi == EPI_QKV) return a.H % 256 == 0 && a.N == 3 * a.H && a.Sp % 64 == 0 && a.Sp >= 64 && a.nh * 64 == a.H;
    if (epi == EPI_QKVR) return a.nq > 0 && a.nkv > 0 && a.nq % 2 == 0 && a.nkv % 2 == 0 && a.N == (a.nq + 2 * a.nkv) * 128 && a.Sp % 32 == 0 && a.Sp >= 32 &&
                                a.Mvalid > 0 && a.Mvalid % a.Sp == 0 && a.Mvalid <= a.Mpad;
    return epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_SWIGLU;
}

// Developer builds (make DEV=1) keep the measured-and-rejected forms of this kernel in their own translation unit (csrc/dev/gemm256x_dev.hip:
// GY images, the 16 x 16 MFMA shapes, stamped and timing-only builds); the product library has no path to them.
#ifdef GLC_DEVELOPER
const char* glc_launch_gemm256x_dev(hipStream_t st, int epi, const GemmArgs& a);
#endif

const char* glc_launch_gemm256x(hipStream_t st, int epi, const GemmArgs& a_in) {
    GemmArgs a = a_in;
    if (a.gy || a.z16 || a.stamps || a.epi_abl || a.prio_mode >= 4) {
#ifdef GLC_DEVELOPER
        return glc_launch_gemm256x_dev(st, epi, a_in);
#else
        return "gemm256x: GY images, the 16 x 16 MFMA shapes, stamps and timing-only builds exist in developer builds only (make DEV=1)";
#endif
    }
#ifdef GLC_DEVELOPER
    {
        static const bool z16_env = glc_dev_env("GLC_GEMM_Z16") && atoi(glc_dev_env("GLC_GEMM_Z16")) != 0;      // developer A/B: the 16 x 16 MFMA shapes
        if (z16_env && a.K % 64 == 0) { a.z16 = 1; return glc_launch_gemm256x_dev(st, epi, a); }
    }
#endif
    if (!a.gx_sat) a.gx_sat = glc_gx_sat_ptr();              // fp8 range guard of the activation images this launch writes
    if (!a.act_sc) a.act_sc = glc_gx_act_sc();               // ... and the exponent of the activation rows (engine.hip act_sc)
    if (a.gx_rows <= 0) a.gx_rows = a.Mvalid > 0 ? a.Mvalid : a.Mpad;     // ... over the rows that exist (slack rows up to Mpad hold leftovers)
    if (!glc_gemm256x_supported(a, epi)) return "gemm256x: unsupported shape";
    if (!a.A || !a.W) return "gemm256x: null operand";
    if (epi == EPI_QKV || epi == EPI_QKVR) { if (!a.Qh || !a.Kh || !a.Vt) return "gemm256x: null QKV output"; if (epi == EPI_QKVR && !a.rope_cs) return "gemm256x: null RoPE table"; }
    else if (!a.C) return "gemm256x: null output";
    if (epi == EPI_RESID && !a.resid) return "gemm256x: null residual";
    const int ntn = a.N / TN;
    switch (epi) {
        case EPI_BIAS: return launch_x<EPI_BIAS, false>(st, a, 0, ntn);
        case EPI_GELU: return launch_x<EPI_GELU, false>(st, a, 0, ntn);
        case EPI_RESID: return launch_x<EPI_RESID, false>(st, a, 0, ntn);
        case EPI_SWIGLU: return a.bias ? "gemm256x: the SwiGLU epilogue takes no bias" : launch_x<EPI_SWIGLU, false>(st, a, 0, ntn);
        case EPI_QKV: {
            const int nqk = 2 * a.H / TN, nq = a.qkv_skip_q ? a.H / TN : 0;
            const char* m = launch_x<EPI_QKV, false>(st, a, nq, nqk - nq);      // (one launch for both, as EPI_QKVR below: measured +-0 at c3 — 6 + 3 full rounds either way)
            return m ? m : launch_x<EPI_QKV, true>(st, a, nqk, ntn - nqk);
        }
        case EPI_QKVR: {
            const int nqk = (a.nq + a.nkv) / 2;              // 256-column tiles of the Q and K heads; the V heads run transposed
            static std::atomic<unsigned> lds_ok{0};
            if (!glc_raise_lds_limit(gemm256x_qkvr_kernel, NSLOT * STAGE, lds_ok)) return "gemm256x: cannot raise the dynamic LDS limit";
            GemmArgs b = a;
            b.n_group = 0;
            b.prio_mode = a.prio_mode >= 0 ? a.prio_mode : 1;
            if (ntn >= 8 && (a.Mpad / TM) % 8 == 0) b.n_group = ntn % 4 == 0 ? 4 : (ntn % 3 == 0 ? 3 : 0);
            hipLaunchKernelGGL(gemm256x_qkvr_kernel, dim3((a.Mpad / TM) * ntn), dim3(512), NSLOT * STAGE, st, b, nqk, ntn);
            return nullptr;
        }
    }
    return "gemm256x: bad epilogue";
}

// In place: n fp32 values (n % 32 == 0) -> GX rows with fp8 exponent sc (glc_common.h): weights at load (sc = glc_gx_weight_exponent,
// worder = 1: [hi8 | lo8] per 8 elements), activations in tests (sc = 0, worder = 0: [lo8 | hi8]).
const char* glc_launch_to_gx(hipStream_t st, void* w, size_t n, int sc, int worder) {
    if (!w || n % 32) return "to_gx: element count must be a multiple of 32";
    if (sc < -40 || sc > 60) return "to_gx: exponent out of range";
    const size_t groups = n / 32;
    if (groups) hipLaunchKernelGGL(to_gx_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, (float*)w, groups, ldexpf(1.0f, sc), ldexpf(1.0f, sc + GLC_GX_SHIFT), worder);
    return nullptr;
}
