// DEVELOPER translation unit (make DEV=1 only; never part of the product library): the measured-and-rejected forms of the MX cross-term
// GEMM of gemm256x.hip, kept compilable with their checks (scripts/gemm_gy_check.py, gemm_z16_check.py, gemm_mx_ablate.py, gemm_stamps.py):
//   GY    e2m3 cross terms with per-16-element block scales (round 4: correct in every epilogue, +-0 ... -4 % on real data)
//   Z16   the main loop on the 16 x 16 MFMA shapes (round 4: correct, -8 ... +5 %)
//   DIAG  s_memtime stamps of the main loop; ABL 4-8: timing-only builds (WRONG results)
// (W128, the one-wave-per-SIMD 128 x 128 wave tile, was deleted in round 5: docs/LOG_r01-r05.md §9 post-mortem.)
// Entry: glc_launch_gemm256x_dev — reached from glc_launch_gemm256x only when a developer field of GemmArgs asks for one of these.
#define glc_launch_gemm256x glc_launch_gemm256x_dev
#define glc_gemm256x_supported glc_gemm256x_supported_dev
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
#include "../glc_common.h"
#include "../glc_kernels.h"
#include "../glc_layout.h"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int LINE = 128;                  // bytes per row and group
constexpr int STAGE = TM * LINE;           // 32 KiB: one operand's rows of one group
constexpr int NSLOT = 4;
constexpr int EPI_PATCH = 9216;            // bytes of wave-private fp32 epilogue staging
constexpr int YST0 = 8 * EPI_PATCH, YSTW = 7168 + 256;   // GY epilogue: behind the fp32 patches, per wave two groups x 32 rows x 112 bytes + the rows' scale dwords,
constexpr int LDS_GY = YST0 + 8 * YSTW;            // written in the image's own layout and flushed as whole KiB (main loop: 28 KiB of every 32 KiB slot, scale ring in slot 0's slack)
extern __shared__ __attribute__((aligned(16))) unsigned char smem256x[];
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)l, 16, 0, 0);
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

template <int EPI, bool VMODE, bool DIAG = false, int ABL = 0, bool GY = false, bool Z16 = false>
__device__ __forceinline__ void gemm256x_tile(const GemmArgs& p, int n_tile0, int ntn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int c32 = lane & 31, h = lane >> 5;
    const int K = p.K, N = p.N;
    const unsigned long long t_entry = DIAG ? __builtin_amdgcn_s_memtime() : 0;

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
    // GY images (group-major, glc_common.h): what a K step stages is 256 rows x 112 bytes = 28 consecutive KiB per operand — 28 one-KiB pieces:
    // waves 0-3 move A's, waves 4-7 W's, seven each — landing linearly (28-dword row pitch: the 16-byte fragment reads of 16 consecutive rows
    // hit 64 distinct banks).  The scale bytes of two groups (one dword per row, 1 KiB per operand) follow as one dword piece per wave.
    // piece pi = wave + 8 j (j = 0 .. 6) of the 56 (A's 28, then W's 28): every wave moves pieces of both operands, as in the GX loop
    const int yop = wave >> 2, ywq = wave & 3;      // (scale pieces: waves 0-3 A's rows 64 ywq .., waves 4-7 W's)
    const unsigned char* const ya = GY ? reinterpret_cast<const unsigned char*>(p.A) + (size_t)m0 * 112 + lane * 16 : nullptr;
    const unsigned char* const yw = GY ? reinterpret_cast<const unsigned char*>(p.W) + (size_t)n0 * 112 + lane * 16 : nullptr;
    const size_t ystep_a = (size_t)p.Mpad * 112, ystep_w = (size_t)N * 112;
    const size_t yrows = GY ? (yop ? (size_t)N : (size_t)p.Mpad) : 0;
    const unsigned char* const ysbase = GY ? (yop ? reinterpret_cast<const unsigned char*>(p.W) + (size_t)ng * N * 112 + (size_t)n0 * 4
                                                  : reinterpret_cast<const unsigned char*>(p.A) + (size_t)ng * p.Mpad * 112 + (size_t)m0 * 4) + (size_t)(ywq * 64 + lane) * 4 : nullptr;
    constexpr int YSC = 256 * 112;              // scale ring: 2 entries x (A 1 KiB | W 1 KiB) in the 4 KiB a 112-byte-pitch slot 0 leaves free

    // DMA map (FL): lane L lands at (row L >> 3, physical chunk L & 7) of an 8-row piece and fetches logical chunk (L & 7) ^ ((row >> 1) & 7)
    const int lrow8 = lane >> 3, pch = lane & 7;
    const unsigned char* fa[2];
    const unsigned char* fw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 8 + lrow8;
        const int lc = pch ^ ((row >> 1) & 7);
        fa[i] = A + (size_t)(m0 + row) * rsb + lc * 16;
        fw[i] = W + (size_t)(n0 + row) * rsb + lc * 16;
    }
    auto stage_fl = [&](int grp) {
        if constexpr (GY) {
            unsigned char* da = smem256x + ((2 * grp) & (NSLOT - 1)) * STAGE;
            unsigned char* dw = smem256x + ((2 * grp + 1) & (NSLOT - 1)) * STAGE;
            const unsigned char* sa_ = ya + (size_t)grp * ystep_a;
            const unsigned char* sw_ = yw + (size_t)grp * ystep_w;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int pi = wave + 8 * j;               // (wave-uniform)
                if (pi < 28) glds16(sa_ + pi * 1024, da + pi * 1024);
                else glds16(sw_ + (pi - 28) * 1024, dw + (pi - 28) * 1024);
            }
            if (!(grp & 1))
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(ysbase + (size_t)(grp >> 1) * yrows * 4),
                                                 (void __attribute__((address_space(3)))*)(smem256x + YSC + ((grp >> 1) & 1) * 2048 + yop * 1024 + ywq * 256), 4, 0, 0);
            return;
        }
        unsigned char* sa = smem256x + ((2 * grp) & (NSLOT - 1)) * STAGE + (wave * 32) * LINE;
        unsigned char* sw = smem256x + ((2 * grp + 1) & (NSLOT - 1)) * STAGE + (wave * 32) * LINE;
        const size_t o = (size_t)grp * LINE;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(fa[i & 1] + (size_t)(i >> 1) * 16 * rsb + o, sa + i * 8 * LINE);
#pragma unroll
        for (int i = 0; i < (ABL == 7 ? 3 : 4); ++i) glds16(fw[i & 1] + (size_t)(i >> 1) * 16 * rsb + o, sw + i * 8 * LINE);      // (ABL 7: 7 of 8 KiB per wave and group, as 112-byte row groups would move)
    };

    // Z16 (round 4): the same products on v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4 — under the chip's power envelope the 16x16
    // f16 shape sustains 1925 TFLOP/s on random operands where 32x32x16 sustains 1622 (profiles/r04/mfma_power_probe.txt: half the accumulator
    // registers read and written per MAC).  zacc[I][J]: 16 x 16 blocks; non-transposed D[n = 16 J + 4 qz + t][m = 16 I + c16], lane = (c16, qz).
    constexpr int ZJ = 4;
    f32x4 zacc[8][ZJ];
    if constexpr (Z16) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < ZJ; ++j) zacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets inside a slot
    const int hsw = (c32 >> 1) & 7;
    constexpr int LINEF = GY ? 112 : LINE;
    const int arow = (wm * 128 + c32) * LINEF, wrow = (wn * 64 + c32) * LINEF;
    const int ck0 = GY ? h * 16 : ((0 + h) ^ hsw) * 16, ck1 = GY ? 32 + h * 16 : ((2 + h) ^ hsw) * 16;       // f16 k-steps 0 / 1: logical chunks h / 2 + h
    const int cx0 = ((4 + 2 * h) ^ hsw) * 16, cx1 = ((5 + 2 * h) ^ hsw) * 16;       // the fp8 parts of elements 16 h .. 16 h + 7 / + 8 .. + 15
    // e8m0 scales (one per operand, every block): A rows (activations, exponent 0) carry the 2^-SHIFT, W rows their 2^-ws
    const int sc_a = 127 - GLC_GX_SHIFT - p.act_sc;        // e8m0 scale of the A blocks: 2^-(SHIFT + sc) (glc_common.h)
    const int sc_w = 127 - p.mx_ws;

    unsigned long long seg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, t_loop1 = 0;
    const unsigned long long clk0 = DIAG ? __builtin_amdgcn_s_memtime() : 0, rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if constexpr (DIAG) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) seg[k] += t - tlast;
            tlast = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f16x8 a16[4][2], w16[2][2];
    i32x8 xa[4], xw[2];
    int ysa[4] = {0, 0, 0, 0}, ysw[2] = {0, 0};      // GY: the lanes' e8m0 scale bytes of the group's blocks
    if constexpr (ABL == 4) {      // (timing-only build without fragment reads: defined operands)
#pragma unroll
        for (int i = 0; i < 4; ++i) { a16[i][0] = a16[i][1] = (f16x8)(f16_t)(0.001f * lane); xa[i] = (i32x8)(0x38383838 + lane); }
#pragma unroll
        for (int j = 0; j < 2; ++j) { w16[j][0] = w16[j][1] = (f16x8)(f16_t)(0.002f * lane); xw[j] = (i32x8)(0x38383838 - lane); }
    }
    
    if constexpr (Z16) {
        // LDS: an f16 ring of 2 groups and an fp8 ring of 3 (the scaled MFMA takes 128 k-slots = the cross terms of TWO groups), 32 KiB each:
        // [A 256 rows x 64 B | W 256 rows x 64 B], the four 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3 (16 consecutive rows' reads of one
        // logical chunk hit 64 distinct banks).  Every wave moves its own 32 rows of A and of W: 8 one-KiB pieces per group.  ONE workgroup
        // barrier per group: behind it group s + 1 is visible and the slots of group s - 1 (f16) / s - 2 (fp8) are free.
        constexpr int ZX = 2 * 32768;
        const int c16 = lane & 15, qz = lane >> 4, swz = (c16 >> 2) & 3;
        const unsigned char* za[2];
        const unsigned char* zw[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = wave * 32 + rb * 16 + (lane >> 2);
            const int lc = (lane & 3) ^ ((row >> 2) & 3);
            za[rb] = A + (size_t)(m0 + row) * rsb + lc * 16;
            zw[rb] = W + (size_t)(n0 + row) * rsb + lc * 16;
        }
        int xs_next = 0;                              // fp8 ring slot of the next group to stage
        auto stage_zf = [&](int grp) {                // the f16 halves of group grp: 4 pieces per wave
            unsigned char* fs = smem256x + (grp & 1) * 32768 + (wave * 2) * 1024;
            const size_t o = (size_t)grp * LINE;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                glds16(za[rb] + o, fs + rb * 1024);
                glds16(zw[rb] + o, fs + 16384 + rb * 1024);
            }
        };
        auto stage_zx = [&](int grp) {                // the fp8 halves: 4 pieces per wave
            unsigned char* xs = smem256x + ZX + xs_next * 32768 + (wave * 2) * 1024;
            const size_t o = (size_t)grp * LINE + 64;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                glds16(za[rb] + o, xs + rb * 1024);
                glds16(zw[rb] + o, xs + 16384 + rb * 1024);
            }
            xs_next = xs_next == 2 ? 0 : xs_next + 1;
        };
        const int zarow = (wm * 128 + c16) * 64, zwrow = 16384 + (wn * 64 + c16) * 64;
        const int zck = (qz ^ swz) * 16;                                         // f16: logical chunk qz (k = 8 qz .. 8 qz + 7 of the group)
        const int zx0 = ((2 * (qz & 1)) ^ swz) * 16, zx1 = ((2 * (qz & 1) + 1) ^ swz) * 16;      // fp8 parts of elements 16 (qz & 1) .. + 7 / + 8 .. + 15
        const int sc_a = 127 - GLC_GX_SHIFT - p.act_sc, sc_w = 127 - p.mx_ws;
        auto ldx = [&](const unsigned char* q0) __attribute__((always_inline)) {
            const i32x4 t0 = *reinterpret_cast<const i32x4*>(q0 + zx0);
            const i32x4 t1 = *reinterpret_cast<const i32x4*>(q0 + zx1);
            i32x8 r;
            r[0] = t0[0]; r[1] = t0[1]; r[2] = t0[2]; r[3] = t0[3]; r[4] = t1[0]; r[5] = t1[1]; r[6] = t1[2]; r[7] = t1[3];
            return r;
        };
        // Per group four barrier slots (the rhythm of the 32 x 32 loop), waves 4-7 — the SIMDs' second waves — one slot behind: A1 requests of group
        // s + 1 and the f16 fragment reads of s | B1 a_hi w_hi (32 MFMAs) | A2 requests landed; odd s: every fp8 fragment of groups s - 1 and s (all
        // of them here, none in B2: the partner group places its next requests into the slot of s - 2 while this group is in B2) | B2 (odd s) both
        // cross terms of the two groups, 32 MFMAs of 128 k-slots: lanes qz < 2 bring group s - 1, qz >= 2 bring s.
        stage_zf(0); stage_zx(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // group 0 is in LDS for everyone
        if (wm == 1) __builtin_amdgcn_s_barrier(); // the stagger
        int xs_prev = 0, xs_cur = 0;               // fp8 ring slots of groups s - 1 and s
        auto zstep = [&](const int s, const bool odd) __attribute__((always_inline)) {
            // ---- A1 ----
            if (s + 1 < ng) { stage_zf(s + 1); stage_zx(s + 1); }
            const unsigned char* fs = smem256x + (s & 1) * 32768;
            f16x8 za16[8], zw16[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) zw16[j] = *reinterpret_cast<const f16x8*>(fs + zwrow + j * 16 * 64 + zck);
#pragma unroll
            for (int i = 0; i < 8; ++i) za16[i] = *reinterpret_cast<const f16x8*>(fs + zarow + i * 16 * 64 + zck);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- B1 ----
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!VMODE) zacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zw16[j], za16[i], zacc[i][j], 0, 0, 0);      // D[n][m]
                    else zacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(za16[i], zw16[j], zacc[i][j], 0, 0, 0);            // D[m][n]
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- A2 ----
            i32x8 zxw[4], zxa[8];
            if (odd) {
                const unsigned char* xb = smem256x + ZX + ((qz >> 1) ? xs_cur : xs_prev) * 32768;
#pragma unroll
                for (int j = 0; j < 4; ++j) zxw[j] = ldx(xb + zwrow + j * 16 * 64);
#pragma unroll
                for (int i = 0; i < 8; ++i) zxa[i] = ldx(xb + zarow + i * 16 * 64);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- B2 ----
            if (odd) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!VMODE) zacc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(zxw[j], zxa[i], zacc[i][j], 0, 0, 0, sc_w, 0, sc_a);
                        else zacc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(zxa[i], zxw[j], zacc[i][j], 0, 0, 0, sc_a, 0, sc_w);
                    }
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            xs_prev = xs_cur; xs_cur = xs_cur == 2 ? 0 : xs_cur + 1;
        };
        for (int s = 0; s < ng; s += 2) { zstep(s, false); zstep(s + 1, true); }      // (ng is even: K % 64 == 0)
        if (wm == 0) __builtin_amdgcn_s_barrier();   // pairs with the late group's last barrier
    } else {
    const int pm = ABL ? 1 : p.prio_mode;     // (timing-only builds: the default policy)  0: no priorities; 1: MFMA phase at priority 1; 2: load phase at priority 2; 3: the late wave group at priority 1 throughout
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
    auto ld24 = [&](const unsigned char* q0, int o0, int o1) __attribute__((always_inline)) {      // (ABL 7) one 16-byte and one 8-byte chunk
        typedef __attribute__((ext_vector_type(2))) int i32x2;
        const i32x4 t0 = *reinterpret_cast<const i32x4*>(q0 + o0);
        const i32x2 t1 = *reinterpret_cast<const i32x2*>(q0 + o1);
        i32x8 r;
        r[0] = t0[0]; r[1] = t0[1]; r[2] = t0[2]; r[3] = t0[3]; r[4] = t1[0]; r[5] = t1[1]; r[6] = 0; r[7] = 0;
        return r;
    };
    auto sub = [&](const int s, const int odd) __attribute__((always_inline)) {
        // ---- phase A ----
        stamp(-1);
        if (pm == 2) __builtin_amdgcn_s_setprio(2);
        if (!odd && s + 1 < ng && ABL != 5) stage_fl(s + 1);      // (ABL 4 / 5 / 6: timing-only builds — no fragment reads / no DMA / no MFMAs; wrong results)
        stamp(5 * odd + 0);
        {
            const unsigned char* sa = smem256x + ((2 * s) & (NSLOT - 1)) * STAGE + arow;
            const unsigned char* sw = smem256x + ((2 * s + 1) & (NSLOT - 1)) * STAGE + wrow;
            if constexpr (ABL == 4) {
            } else if (!odd) {
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
            } else if constexpr (GY) {        // the lane's block (elements 16 h ..): 16 + 8 bytes of e2m3 parts and its scale byte
                const unsigned char* ysc = smem256x + YSC + ((s >> 1) & 1) * 2048 + 2 * (s & 1) + h;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    xw[j] = ld24(sw + j * 32 * LINEF, 64 + 16 * h, 96 + 8 * h);
                    ysw[j] = ysc[1024 + (wn * 64 + j * 32 + c32) * 4];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xa[i] = ld24(sa + i * 32 * LINEF, 64 + 16 * h, 96 + 8 * h);
                    ysa[i] = ysc[(wm * 128 + i * 32 + c32) * 4];
                }
            } else if constexpr (ABL == 7) {      // (timing only: 24 instead of 32 bytes per lane and block, as fp6 parts would be read)
#pragma unroll
                for (int j = 0; j < 2; ++j) xw[j] = ld24(sw + j * 32 * LINE, cx0, cx1);
#pragma unroll
                for (int i = 0; i < 4; ++i) xa[i] = ld24(sa + i * 32 * LINE, cx0, cx1);
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
        stamp(5 * odd + 1);
        __builtin_amdgcn_s_barrier();
        stamp(5 * odd + 2);
        // ---- phase B ----
        if (pm == 1) __builtin_amdgcn_s_setprio(1);
        if constexpr (ABL == 6) {
        } else if (!odd) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (ABL == 8) {       // (timing only, wrong results: the same MACs, operand and accumulator registers as two v_mfma_f32_16x16x32_f16 — the shape's power)
                            typedef float f32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                f32x4_ sub = {acc[i][j][8 * ks + 4 * t], acc[i][j][8 * ks + 4 * t + 1], acc[i][j][8 * ks + 4 * t + 2], acc[i][j][8 * ks + 4 * t + 3]};
                                sub = __builtin_amdgcn_mfma_f32_16x16x32_f16(w16[j][t], a16[i][ks], sub, 0, 0, 0);
                                acc[i][j][8 * ks + 4 * t] = sub[0]; acc[i][j][8 * ks + 4 * t + 1] = sub[1]; acc[i][j][8 * ks + 4 * t + 2] = sub[2]; acc[i][j][8 * ks + 4 * t + 3] = sub[3];
                            }
                        } else
                        if (!VMODE) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w16[j][ks], a16[i][ks], acc[i][j], 0, 0, 0);      // D[n][m]
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[i][ks], w16[j][ks], acc[i][j], 0, 0, 0);            // D[m][n]
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (ABL == 7) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xw[j], xa[i], acc[i][j], 2, 2, 0, sc_w, 0, sc_a);      // (timing only: both operands read as e2m3)
                    else if constexpr (GY) {       // e2m3 parts, per-lane block scales
                        if (!VMODE) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xw[j], xa[i], acc[i][j], 2, 2, 0, ysw[j], 0, ysa[i]);
                        else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[i], xw[j], acc[i][j], 2, 2, 0, ysa[i], 0, ysw[j]);
                    } else
                    if (!VMODE) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xw[j], xa[i], acc[i][j], 0, 0, 0, sc_w, 0, sc_a);
                    else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[i], xw[j], acc[i][j], 0, 0, 0, sc_a, 0, sc_w);
                }
        }
        if (pm == 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(5 * odd + 3);
        __builtin_amdgcn_s_barrier();
        stamp(5 * odd + 4);
    };
    for (int s = 0; s < ng; ++s) { sub(s, 0); sub(s, 1); }
    if (wm == 0) __builtin_amdgcn_s_barrier();   // pairs with the late group's last barrier
    }
    if constexpr (DIAG) {
        if (p.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {
            unsigned long long* o = p.stamps + ((size_t)(blockIdx.x >> 3) * 8 + wave) * 12;
            for (int k = 0; k < 10; ++k) o[k] = seg[k];
            const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk0, dr = __builtin_amdgcn_s_memrealtime() - rt0;
            o[10] = dr ? dc * 1000 / dr : 0; o[11] = ng;
        }
        t_loop1 = __builtin_amdgcn_s_memtime();
    }

    // ---------------- epilogue ----------------
    {
    constexpr int zj0 = 0;
    typedef f16_t T;
    typedef __attribute__((ext_vector_type(8))) T vec8T;
    const float* __restrict__ bias = p.bias;
    float* stg = reinterpret_cast<float*>(smem256x + wave * EPI_PATCH);
    unsigned char* const yst = smem256x + YST0 + wave * YSTW;       // (GY output)
    // GY output of one 32-row chunk: the lanes have written their blocks into yst as [ngl groups][32 rows][112 bytes] (+ a scale dword per row at
    // 7168); in the group-major image each of those groups is 3584 consecutive bytes: whole-KiB non-temporal stores
    auto gy_flush = [&](unsigned char* cbase, size_t crow0, int cK, int g0, int ngl) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int nq = ngl * 224;
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int q = it * 64 + lane;
            if (it * 64 >= nq) break;
            if (q < nq) {
                const int gl = q >= 224 ? 1 : 0, rem = q - 224 * gl;
                const u32x4 t = *reinterpret_cast<const u32x4*>(yst + q * 16);
                __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(cbase + gy_group_off(p.Mpad, crow0, g0 + gl) + rem * 16));
            }
        }
        if (lane < 32) {
            const unsigned t = *reinterpret_cast<const unsigned*>(yst + 7168 + lane * 4);
            unsigned char* sp = cbase + gy_scale_off(p.Mpad, cK, crow0 + lane, g0);
            if (ngl == 2) *reinterpret_cast<unsigned*>(sp) = t;
            else *reinterpret_cast<unsigned short*>(sp) = (unsigned short)t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
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
            if constexpr (Z16) {         // 16 x 16 blocks: J even = 16 gate features, J odd = the matching up features, same lane and register
                const int c16 = lane & 15, qz = lane >> 4;
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const float rz = lnf ? p.a_stats[m0 + wm * 128 + c * 32 + 16 * ip + c16].y : 1.0f;
#pragma unroll
                    for (int Jp = 0; Jp < 2; ++Jp) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float gt = zacc[2 * c + ip][zj0 + 2 * Jp][e] * rz, up = zacc[2 * c + ip][zj0 + 2 * Jp + 1][e] * rz;
                            v[e] = gt * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt)) * up;
                        }
                        *reinterpret_cast<f32x4*>(stg + (16 * ip + c16) * 36 + 16 * Jp + 4 * qz) = v;
                    }
                }
            } else
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
            if constexpr (GY) {          // one 16-element block per lane: row lane >> 1, features 16 (lane & 1) ..
                const int row = lane >> 1, g2 = lane & 1;
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(stg + row * 36 + g2 * 16 + 4 * q);
                    v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
                }
                const unsigned sb = gy_store16<false, false, false>(yst, 32, 64, row, g2 * 16, v);
                yst[7168 + row * 4 + g2] = (unsigned char)sb;
                gy_flush(reinterpret_cast<unsigned char*>(p.C), (size_t)(m0 + wm * 128 + c * 32), Iw, ((n0 >> 1) + wn * 32) >> 5, 1);
            } else
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
                const int nn = Z16 ? n0 + wn * 64 + 32 * J + 8 * q + 4 * (lane >> 4) : n0 + wn * 64 + 32 * J + 8 * q + 4 * h;      // Z16: entries q = 0 / 2 = columns 16 (2 J + q / 2) + 4 qz ..
                bj[J][q] = bias ? *reinterpret_cast<const f32x4*>(bias + nn) : (f32x4){0.f, 0.f, 0.f, 0.f};
                cj[J][q] = (lnf && p.ln_c) ? *reinterpret_cast<const f32x4*>(p.ln_c + nn) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        float rg[8], rb[8];
        const bool rln = EPI == EPI_RESID && p.r_stats != nullptr;
        const bool gxout = EPI == EPI_RESID && p.ln_part != nullptr;      // raw GX rows + statistics partials out
        if constexpr (EPI == EPI_RESID && !GY) {
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
        // GY residual rows: lane = one 16-element block (row idx >> 2, columns 16 (idx & 3) ..), two per 32-row chunk
        struct YRes { gs_h8 h0, h1; u32x4 r4; u32x2 r2; unsigned b; float2 st; };
        YRes ypre[2], ycur[2];
        auto load_resid_y = [&](int c, YRes (&r)[2]) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = lane + 64 * k, row = idx >> 2, g4 = idx & 3;
                const int mrow = m0 + wm * 128 + c * 32 + row, n = n0 + wn * 64 + g4 * 16;
                r[k].st = rln ? p.r_stats[mrow] : make_float2(0.f, 1.f);
                const unsigned char* rr = reinterpret_cast<const unsigned char*>(p.resid);
                const unsigned char* rp = rr + gy_group_off(p.Mpad, mrow, n >> 5);
                const int hb = (n >> 4) & 1;
                r[k].h0 = *reinterpret_cast<const gs_h8*>(rp + 32 * hb);
                r[k].h1 = *reinterpret_cast<const gs_h8*>(rp + 32 * hb + 16);
                r[k].r4 = *reinterpret_cast<const u32x4*>(rp + 64 + 16 * hb);
                r[k].r2 = *reinterpret_cast<const u32x2*>(rp + 96 + 8 * hb);
                r[k].b = rr[gy_scale_off(p.Mpad, N, mrow, n >> 5) + hb];
            }
        };
        if constexpr (EPI == EPI_RESID) { if constexpr (GY) load_resid_y(0, ypre); else load_resid(0, rpre, rpre_lo, rst_pre); }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            gs_h8 rcur[4]; u32x2 rcur_lo[4]; float2 rst_cur[4];
            if constexpr (EPI == EPI_RESID && GY) {
                ycur[0] = ypre[0]; ycur[1] = ypre[1];
                if (c + 1 < 4) load_resid_y(c + 1, ypre);
            } else
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { rcur[k] = rpre[k]; rcur_lo[k] = rpre_lo[k]; rst_cur[k] = rst_pre[k]; }
                if (c + 1 < 4) load_resid(c + 1, rpre, rpre_lo, rst_pre);
            }
            const float2 sm = lnf ? p.a_stats[m0 + wm * 128 + c * 32 + c32] : make_float2(0.f, 1.f);
            if constexpr (Z16) {         // D[n = 16 J + 4 qz + t][m = 16 I + c16] -> patch [m][n]
                const int c16 = lane & 15, qz = lane >> 4;
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const float2 smz = lnf ? p.a_stats[m0 + wm * 128 + c * 32 + 16 * ip + c16] : make_float2(0.f, 1.f);
#pragma unroll
                    for (int J = 0; J < 4; ++J) {
                        f32x4 v = zacc[2 * c + ip][zj0 + J];
                        if constexpr (EPI != EPI_RESID) {
                            if (lnf) {
                                const f32x4 cz = cj[J >> 1][2 * (J & 1)];        // (Z16: cj / bj hold the lane's four 16-column blocks, see their loads)
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = smz.y * (v[r] - smz.x * cz[r]);
                            }
                        }
                        v += bj[J >> 1][2 * (J & 1)];
                        if (EPI == EPI_GELU) { const f32x2 g0 = glc_gelu2_f32((f32x2){v[0], v[1]}), g1 = glc_gelu2_f32((f32x2){v[2], v[3]}); v = (f32x4){g0[0], g0[1], g1[0], g1[1]}; }
                        *reinterpret_cast<f32x4*>(stg + (16 * ip + c16) * 68 + 16 * J + 4 * qz) = v;
                    }
                }
            } else
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
            } else if constexpr (GY && (EPI == EPI_BIAS || EPI == EPI_GELU || EPI == EPI_RESID)) {
                // GY rows out (and in: the residual): one 16-element block per lane — row idx >> 2, columns 16 (idx & 3) .. of the wave's 64
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int idx = lane + 64 * k, row = idx >> 2, g4 = idx & 3;
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(stg + row * 68 + g4 * 16 + 4 * q);
                        v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
                    }
                    const int m = m0 + wm * 128 + c * 32 + row;
                    const int n = n0 + wn * 64 + g4 * 16;
                    if constexpr (EPI == EPI_RESID) {
                        float r[16];
                        gy_decode16(ycur[k].h0, ycur[k].h1, ycur[k].r4, ycur[k].r2, ycur[k].b, r);
                        if (rln) {           // raw residual row: LayerNorm on the fly
                            const float2 rs = ycur[k].st;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 gq = *reinterpret_cast<const f32x4*>(p.r_gamma + n + 4 * q), bq = *reinterpret_cast<const f32x4*>(p.r_beta + n + 4 * q);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[4 * q + e] += (r[4 * q + e] - rs.x) * rs.y * gq[e] + bq[e];
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 16; ++e) v[e] += r[e];
                        }
                        if (gxout) {         // raw GY row out + this 64-column block's (sum, squared deviations from the block mean) of the row
                            float s1 = 0.f, s2 = 0.f;
#pragma unroll
                            for (int e = 0; e < 16; ++e) s1 += v[e];
                            s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
                            const float bm = s1 * (1.0f / 64.0f);
#pragma unroll
                            for (int e = 0; e < 16; ++e) { const float dv = v[e] - bm; s2 += dv * dv; }
                            s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
                            if (g4 == 0) p.ln_part[(size_t)m * (N >> 6) + ((n0 + wn * 64) >> 6)] = make_float2(s1, s2);
                            const unsigned sb = gy_store16<false, false, false>(yst, 32, 64, row, g4 * 16, v);
                            yst[7168 + row * 4 + g4] = (unsigned char)sb;
                        } else {             // plain fp32 row
                            float* cp = reinterpret_cast<float*>(p.C) + (size_t)m * N + n;
#pragma unroll
                            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(cp + 4 * q) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                        }
                    } else {
                        if (p.gs_c_plain) {
                            const int nl = n < p.perm_cols ? (n & ~127) | glc_rope_perm128(n & 127) : n;
                            float* cp = reinterpret_cast<float*>(p.C) + (size_t)m * N + nl;
#pragma unroll
                            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(cp + 4 * q) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                        } else {
                            const unsigned sb = gy_store16<false, false, false>(yst, 32, 64, row, g4 * 16, v);
                            yst[7168 + row * 4 + g4] = (unsigned char)sb;
                        }
                    }
                }
                if ((EPI == EPI_RESID && gxout) || (EPI != EPI_RESID && !p.gs_c_plain))
                    gy_flush(reinterpret_cast<unsigned char*>(p.C), (size_t)(m0 + wm * 128 + c * 32), N, (n0 + wn * 64) >> 5, 2);
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
#ifdef GLC_DEVELOPER
                    if (p.epi_abl == 0)
#endif
                    gx_store8<false, true>(reinterpret_cast<unsigned char*>(p.C) + (size_t)m * 4 * N, n, v, kHi, kLo, m < p.gx_rows ? p.gx_sat : nullptr);      // FFN1's intermediate: streams (non-temporal)
#ifdef GLC_DEVELOPER
                    else if (p.epi_abl == 2) gx_store8<false, true>(reinterpret_cast<unsigned char*>(p.C) + (size_t)(m & 255) * 4 * N, n, v, kHi, kLo, nullptr);      // (timing only: cache-resident target)
                    else if (p.epi_abl == 3) gx_store8<false, false>(reinterpret_cast<unsigned char*>(p.C) + (size_t)m * 4 * N, n, v, kHi, kLo, nullptr);      // (timing only: temporal stores)
#endif
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
            if constexpr (Z16) {         // D[m = 16 I + 4 qz + t][n = 16 J + c16] -> patch [n][m]
                const int c16 = lane & 15, qz = lane >> 4;
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                    for (int J = 0; J < 4; ++J) {
                        f32x4 v = zacc[2 * c + ip][zj0 + J];
                        const int nn = n0 + wn * 64 + 16 * J + c16;
                        if (lnf) {
                            const float cz = p.ln_c ? p.ln_c[nn] : 0.f;
                            const float2* sp = p.a_stats + m0 + wm * 128 + c * 32 + 16 * ip + 4 * qz;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const float2 sm = sp[r]; v[r] = sm.y * (v[r] - sm.x * cz); }
                        }
                        const float bz = bias ? bias[nn] : 0.f;
                        v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
                        *reinterpret_cast<f32x4*>(stg + (16 * J + c16) * 36 + 16 * ip + 4 * qz) = v;
                    }
            } else
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
    if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {
            unsigned long long* o = p.stamps + 64 * 8 * 12 + ((size_t)(blockIdx.x >> 3) * 8 + wave) * 2;
            o[0] = clk0 - t_entry;
            o[1] = __builtin_amdgcn_s_memtime() - t_loop1;
        }
    }
}

template <int EPI, bool VMODE, bool DIAG = false, int ABL = 0, bool GY = false, bool Z16 = false>
__global__ __launch_bounds__(512, 2) void gemm256x_kernel(GemmArgs p, int n_tile0, int ntn) { gemm256x_tile<EPI, VMODE, DIAG, ABL, GY, Z16>(p, n_tile0, ntn); }
constexpr int LDS_Z16 = 5 * 32768;                 // Z16: f16 ring (2 groups) + fp8 ring (3 groups)

// Decoder QKV with the RoPE / MX-tile epilogue in ONE launch: the N-tiles of the V heads (nt >= nqk) run the transposed tile (c5: 7 + 1 N-tiles x
// 128 M-tiles = 4 full rounds of the chip; as two launches the V heads' 128 workgroups would be a fifth, half-empty round)
template <bool GY>
__global__ __launch_bounds__(512, 2) void gemm256x_qkvr_kernel(GemmArgs p, int nqk, int ntn) {
    int mt, nt;
    x_tile_of_block(p, ntn, mt, nt);
    if (nt < nqk) gemm256x_tile<EPI_QKVR, false, false, 0, GY>(p, 0, ntn);
    else gemm256x_tile<EPI_QKVR, true, false, 0, GY>(p, 0, ntn);
}

template <int EPI, bool VMODE, bool DIAG = false, int ABL = 0, bool GY = false, bool Z16 = false> const char* launch_x(hipStream_t st, const GemmArgs& a, int n_tile0, int ntn) {
#ifdef GLC_DEVELOPER      // the 16 x 16 MFMA shapes (Z16): correct in every epilogue, -8 ... +5 % against the 32 x 32 loop (docs/LOG_r01-r05.md §9) — developer builds only
    if constexpr (!DIAG && ABL == 0 && !GY && !Z16) {
        if (a.z16 && !a.gy && a.K % 64 == 0 && !a.stamps && a.prio_mode < 4) return launch_x<EPI, VMODE, false, 0, false, true>(st, a, n_tile0, ntn);
    }
#endif
    if constexpr (!DIAG && ABL == 0 && !GY && !Z16) {
        if (a.gy) {
#ifdef GLC_DEVELOPER      // GY images (e2m3 parts with block scales): built and verified, not faster under the chip's power envelope (docs/LOG_r01-r05.md §6) — developer builds only
            if constexpr (EPI == EPI_BIAS && !VMODE) { if (a.stamps) return launch_x<EPI, VMODE, true, 0, true>(st, a, n_tile0, ntn); }
            return launch_x<EPI, VMODE, false, 0, true>(st, a, n_tile0, ntn);
#else
            return "gemm256x: GY images are compiled in by make DEV=1 only";
#endif
        }
    }
    if constexpr (!DIAG && ABL == 0 && EPI == EPI_BIAS && !VMODE && !GY) {
#ifdef GLC_DEVELOPER      // timing-only builds of the main loop (scripts/gemm_mx_ablate.py): 4 no fragment reads, 5 no DMA, 6 no MFMAs, 7 the traffic and MFMA format of fp6 cross terms
        if (a.prio_mode == 4) return a.stamps ? launch_x<EPI, VMODE, true, 4>(st, a, n_tile0, ntn) : launch_x<EPI, VMODE, false, 4>(st, a, n_tile0, ntn);
        if (a.prio_mode == 5) return a.stamps ? launch_x<EPI, VMODE, true, 5>(st, a, n_tile0, ntn) : launch_x<EPI, VMODE, false, 5>(st, a, n_tile0, ntn);
        if (a.prio_mode == 6) return a.stamps ? launch_x<EPI, VMODE, true, 6>(st, a, n_tile0, ntn) : launch_x<EPI, VMODE, false, 6>(st, a, n_tile0, ntn);
        if (a.prio_mode == 7) return a.stamps ? launch_x<EPI, VMODE, true, 7>(st, a, n_tile0, ntn) : launch_x<EPI, VMODE, false, 7>(st, a, n_tile0, ntn);
        if (a.prio_mode == 8) return launch_x<EPI, VMODE, false, 8>(st, a, n_tile0, ntn);
#endif
        if (a.stamps) return launch_x<EPI, VMODE, true>(st, a, n_tile0, ntn);
    }
    static std::atomic<unsigned> lds_ok{0};
    constexpr int lds_bytes = Z16 ? LDS_Z16 : GY ? LDS_GY : NSLOT * STAGE;
    if (!glc_raise_lds_limit(gemm256x_kernel<EPI, VMODE, DIAG, ABL, GY, Z16>, lds_bytes, lds_ok)) return "gemm256x: cannot raise the dynamic LDS limit";
    const int grid = (a.Mpad / TM) * ntn;
    GemmArgs b = a;
    b.n_group = 0;
    static const int prio_env = glc_dev_env("GLC_GEMM_PRIO") ? atoi(glc_dev_env("GLC_GEMM_PRIO")) : 1;      // developer A/B switch
    b.prio_mode = a.prio_mode >= 0 ? a.prio_mode : prio_env;
    if (ntn >= 8 && (a.Mpad / TM) % 8 == 0) b.n_group = ntn % 4 == 0 ? 4 : (ntn % 3 == 0 ? 3 : 0);      // wide N: as gemm256s.hip
    hipLaunchKernelGGL((gemm256x_kernel<EPI, VMODE, DIAG, ABL, GY, Z16>), dim3(grid), dim3(512), lds_bytes, st, b, n_tile0, ntn);
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

// ---- GY rows (glc_common.h): conversions for weights at load, for tests and developer tools ----
// plain fp32 rows [rows][K] -> GY rows (A order, or W order)
__global__ __launch_bounds__(256) void to_gy_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t nblocks, size_t rows, int K, int worder) {
    const size_t bi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (bi >= nblocks) return;
    const size_t row = bi / (size_t)(K >> 4);
    const int e0 = (int)(bi - row * (size_t)(K >> 4)) * 16;
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + row * K + e0 + 4 * q);
        v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
    if (worder) gy_store16<true>(dst, rows, K, row, e0, v);
    else gy_store16<false>(dst, rows, K, row, e0, v);
}
// group-split rows ([32 hi | 32 lo] f16 halves per 32 values) -> GY rows in W order: the projection weights from their split-f16 copies on the device
__global__ __launch_bounds__(256) void gs_to_gy_kernel(const f16_t* __restrict__ gs, unsigned char* __restrict__ dst, size_t nblocks, size_t rows, int K) {
    const size_t bi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (bi >= nblocks) return;
    const size_t row = bi / (size_t)(K >> 4);
    const int e0 = (int)(bi - row * (size_t)(K >> 4)) * 16;
    const f16_t* p = gs + row * 2 * K + (size_t)(e0 >> 5) * 64 + (e0 & 31);
    float v[16];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const gs_h8 hi = *reinterpret_cast<const gs_h8*>(p + 8 * i), lo = *reinterpret_cast<const gs_h8*>(p + 32 + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[8 * i + e] = (float)hi[e] + (float)lo[e];
    }
    gy_store16<true>(dst, rows, K, row, e0, v);
}
// GY rows (A order) -> plain fp32 rows: x = hi + lo
__global__ __launch_bounds__(256) void gy_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t nblocks, size_t rows, int K) {
    const size_t bi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (bi >= nblocks) return;
    const size_t row = bi / (size_t)(K >> 4);
    const int e0 = (int)(bi - row * (size_t)(K >> 4)) * 16;
    float v[16];
    gy_load16(src, rows, K, row, e0, v);
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dst + row * K + e0 + 4 * q) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}

}  // namespace

const char* glc_launch_to_gy(hipStream_t st, const float* src, void* dst, size_t rows, int K, int worder) {
    if (!src || !dst || K <= 0 || K % 64) return "to_gy: K must be a multiple of 64";
    const size_t nb = rows * (size_t)(K >> 4);
    if (nb) hipLaunchKernelGGL(to_gy_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, src, (unsigned char*)dst, nb, rows, K, worder);
    return nullptr;
}
const char* glc_launch_gs_to_gy(hipStream_t st, const void* gs, void* dst, size_t rows, int K) {
    if (!gs || !dst || K <= 0 || K % 64) return "gs_to_gy: K must be a multiple of 64";
    const size_t nb = rows * (size_t)(K >> 4);
    if (nb) hipLaunchKernelGGL(gs_to_gy_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const f16_t*)gs, (unsigned char*)dst, nb, rows, K);
    return nullptr;
}
const char* glc_launch_gy_to_f32(hipStream_t st, const void* src, float* dst, size_t rows, int K) {
    if (!src || !dst || K <= 0 || K % 64) return "gy_to_f32: K must be a multiple of 64";
    const size_t nb = rows * (size_t)(K >> 4);
    if (nb) hipLaunchKernelGGL(gy_to_f32_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, (const unsigned char*)src, dst, nb, rows, K);
    return nullptr;
}


bool glc_gemm256x_supported(const GemmArgs& a, int epi) {
    if (!(a.Mpad > 0 && a.Mpad % TM == 0 && a.N > 0 && a.N % TN == 0 && a.K > 0 && a.K % 32 == 0)) return false;
    if (a.mx_ws < -40 || a.mx_ws > 60) return false;
    if (a.gy && (a.K % 64 || a.N % 64)) return false;        // GY rows: scale bytes of two groups travel as one dword
    if (epi == EPI_QKV) return a.H % 256 == 0 && a.N == 3 * a.H && a.Sp % 64 == 0 && a.Sp >= 64 && a.nh * 64 == a.H;
    if (epi == EPI_QKVR) return a.nq > 0 && a.nkv > 0 && a.nq % 2 == 0 && a.nkv % 2 == 0 && a.N == (a.nq + 2 * a.nkv) * 128 && a.Sp % 32 == 0 && a.Sp >= 32 &&
                                a.Mvalid > 0 && a.Mvalid % a.Sp == 0 && a.Mvalid <= a.Mpad;
    return epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_SWIGLU;
}

const char* glc_launch_gemm256x(hipStream_t st, int epi, const GemmArgs& a_in) {
    GemmArgs a = a_in;
    if (!a.gx_sat) a.gx_sat = glc_gx_sat_ptr();              // fp8 range guard of the activation images this launch writes
    static const bool z16_env = glc_dev_env("GLC_GEMM_Z16") && atoi(glc_dev_env("GLC_GEMM_Z16")) != 0;      // developer A/B: the 16 x 16 MFMA shapes
    if (z16_env) a.z16 = 1;
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
#ifdef GLC_DEVELOPER
            static std::atomic<unsigned> lds_ok_y{0};
            if (a.gy && !glc_raise_lds_limit(gemm256x_qkvr_kernel<true>, LDS_GY, lds_ok_y)) return "gemm256x: cannot raise the dynamic LDS limit";
#else
            if (a.gy) return "gemm256x: GY images are compiled in by make DEV=1 only";
#endif
            if (!a.gy && !glc_raise_lds_limit(gemm256x_qkvr_kernel<false>, NSLOT * STAGE, lds_ok)) return "gemm256x: cannot raise the dynamic LDS limit";
            GemmArgs b = a;
            b.n_group = 0;
            b.prio_mode = a.prio_mode >= 0 ? a.prio_mode : 1;
            if (ntn >= 8 && (a.Mpad / TM) % 8 == 0) b.n_group = ntn % 4 == 0 ? 4 : (ntn % 3 == 0 ? 3 : 0);
#ifdef GLC_DEVELOPER
            if (a.gy) hipLaunchKernelGGL(gemm256x_qkvr_kernel<true>, dim3((a.Mpad / TM) * ntn), dim3(512), LDS_GY, st, b, nqk, ntn);
            else
#endif
            hipLaunchKernelGGL(gemm256x_qkvr_kernel<false>, dim3((a.Mpad / TM) * ntn), dim3(512), NSLOT * STAGE, st, b, nqk, ntn);
            return nullptr;
        }
    }
    return "gemm256x: bad epilogue";
}

