// 256x256 MFMA GEMM with temporally staggered wave groups.  (Its lock-step predecessor gemm256.hip — 2 x 64 KiB stages, every wave
// loading and multiplying in step — was removed in round 3; the two main loops of THIS file, half-line and full-line stages, keep the
// same accumulation order and give bit-identical outputs, which scripts/race_screen.py uses as a race detector.)
//
// Stamps in the lock-step main loop showed, per 64-deep K-tile and wave, ~2440 cycles in the load+MFMA segment (floor 2048:
// two waves share one SIMD's matrix pipe) plus ~800 cycles at s_waitcnt/s_barrier where the pipe idles: the two waves
// of a SIMD execute [loads ... MFMAs ... wait] in lockstep, so nothing covers the load and wait parts.  Here:
//   * K is consumed in 32-deep steps on a 4-slot LDS ring (32 KiB per slot), DMA issued three steps ahead, counted
//     s_waitcnt vmcnt(8|4|0), raw s_barrier;
//   * every wave runs the SAME two-phase step — phase A: issue DMA(s+3), read the 12 fragments of step s, wait;
//     phase B: 32 MFMAs from registers — with a barrier after each phase;
//   * the waves of a workgroup form two groups that sit pairwise on the same SIMDs (wave i and i+4); group 1 enters the
//     loop one barrier late (and group 0 leaves one barrier late), so on every SIMD one wave is in B (matrix pipe) while
//     its partner is in A (LDS reads + DMA issue).  The stagger is purely temporal: no per-group code.
// Hazards (time slot of A_s is 2s for group 0, 2s+1 for group 1): RAW — step s+1 is first read at slot 2s+2 and every
// wave makes its own DMA pieces of step s+1 land (vmcnt) before the barrier that ends its A_s (slot <= 2s+1).  WAR —
// DMA(s+3) reuses the slot of step s-1, last read at slot 2s-1 with lgkmcnt(0) before that slot's barrier.  The epilogue
// patches reuse the ring: group 0's extra final barrier pairs with group 1's last loop barrier, after which no wave
// reads the ring any more.
// Measured (same box, f16): 4096^3 947 -> 1247 TF; 65536x768x3072 860 -> 1069 TF; 65536x3072x768+GELU 640 -> 793 TF.
// Tried on top, no gain: DMA issue interleaved into phase B (-1..3 %), a 5-slot ring (+-1 %), persistent workgroups (+-1 %).
#include <stdlib.h>
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

constexpr int TM = 256, TN = 256;
constexpr int ROWB = 64;                   // bytes of K per row per step (32 x 16-bit)
constexpr int STAGE = (TM + TN) * ROWB;    // 32 KiB
constexpr int NSLOT = 4;                   // LDS ring slots; DMA runs 3 steps ahead (a 5-slot ring measured no faster)
constexpr int EPI_PATCH = 9216;            // bytes of wave-private fp32 epilogue staging (8 x 9 KiB < 2 stages)
extern __shared__ __attribute__((aligned(16))) unsigned char smem256s[];
#define smem256 smem256s
// chunk swizzle of the [rows][64 B] image: 4 rows fill the 64 banks; chunk ^= s(row>>2), s = {0,2,3,1} makes every
// ds_read_b128 lane group of a 16-row fragment read touch 16 distinct 16-B bank slots
__device__ __forceinline__ int swz4(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

__device__ __forceinline__ void glds16(const void* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)l, 16, 0, 0);
}

// GS (T = f16_t only): the operands of the fp32 mode.  A and W hold group-split rows (glc_kernels.h: every 32 fp32 values stored as
// [32 hi halves | 32 lo halves], 128 bytes), and a product is the three f16 MFMAs a_lo*w_hi + a_hi*w_lo + a_hi*w_hi: the same
// ring and phases over 2K/32 stages, where stage 2s fetches the (hi, lo) and stage 2s + 1 the (lo, hi) 64-byte parts of group s of
// the A and W rows (see the main loop: every part is fetched once, the a_hi fragments wait in registers).  Epilogues: GELU (exact erf form) / BIAS write C in the GS format; RESID reads the residual in the GS format and
// writes plain fp32 (the LayerNorm input); QKV writes split-f16 fragment units [8 hi | 8 lo] (glc_common.h f16x8s).
// PD (GS only): precision-budget build (glc_debug_set_precision_mask): GemmArgs::prec drops operand lo halves at run time, which is
// numerically the kernel with that operand rounded to f16 (the skipped terms are exact zeros) at unchanged cost.
// FL ("full lines", round 3): operand-major ring stages.  The loop above it fetches, per 32-deep step, 64 bytes of every A row and
// 64 bytes of every W row: each LDS-DMA wave-instruction covers 16 rows x HALF a 128-byte line, and every line is requested twice
// (by two different stages).  The DMA path, not the MFMA count, bounds that loop (round-2 ablation: 760 of the ~1250 cycles of a
// load phase are the 4 DMA pieces per wave; the vector-memory path pays per line touched).  Here a ring stage is ONE operand's rows
// for one 128-byte "group" — a GS group [32 hi | 32 lo], or 64 k-values of a 16-bit row — so a wave-instruction moves 8 rows x 128 B
// = 8 whole lines, each line exactly once: stage 2s = A rows of group s, stage 2s + 1 = W rows of group s, slot = stage & 3.
// LDS image of a slot: [256 rows][128 B], physical 16-byte chunk c of row r holds logical chunk c ^ ((r >> 1) & 7) (source-side
// swizzle: every ds_read_b128 lane group of a 16-row fragment read touches 16 distinct 16-B bank slots — two rows fill the 64 banks).
// Schedule per group s (same two-phase step and wave-group stagger as above; MFMA order unchanged, results bit-identical):
//   E: phase A = issue the DMA of group s + 1 (8 pieces per wave), read a[first half] + w[matching half]; phase B = 32 MFMAs
//   O: phase A = read the other halves, wait for my pieces of group s + 1;                                  phase B = 64 (GS) / 32 MFMAs
// Hazards (time slot of E_s phase A: 4s for group 0, 4s + 1 for the late group).  RAW: group s + 1 is first read at slot 4s + 4;
// every wave waits for its own pieces (vmcnt(0): nothing else is in flight) in phase A of O_s, slot 4s + 2 / 4s + 3, before that
// slot's barrier — at least a whole phase after requesting them.  WAR: group s + 1 lands in the slots of group s - 1, last read in
// phase A of O_(s-1), slots 4s - 2 / 4s - 1, retired (lgkmcnt(0)) before those slots' barriers; the requests go out in slots 4s / 4s + 1.
// DIAG (FL only): s_memtime stamps at the phase boundaries of the main loop, summed per wave (glc_debug_gemm_bench which = 7 prints them).
template <typename T, int EPI, bool VMODE, bool GS = false, bool PD = false, bool FL = false, bool DIAG = false>
__global__ __launch_bounds__(512, 2) void gemm256s_kernel(GemmArgs p, int n_tile0, int ntn) {
    static_assert(!(PD && FL), "the precision-budget build uses the half-line loop");
    static_assert(!DIAG || FL, "stamps exist in the full-line loop");
    const unsigned long long t_entry = DIAG ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long t_loop0 = 0, t_loop1 = 0;
    static_assert(!GS || sizeof(T) == 2, "GS operands are f16 halves");
    typedef typename Frag<T>::type frag_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15, g = lane >> 4;
    const int K = p.K, N = p.N;

    // XCD-aware tile order (bijective for any grid size)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int mt = tile / ntn, nt = tile % ntn;
    // Wide N (FFN1: 12 N-tiles = 9.4 MB of W against a 4 MB L2 per XCD): walking a whole row of N-tiles before the next M-tile re-fetches
    // every W tile for every ~3 M-tiles (FETCH_SIZE 1.25 GB per launch against 0.21 GB of operands).  Instead each XCD takes its share of
    // the M-tiles and sweeps them once per group of <= 4 N-tiles, so the group's W tiles stay in its L2: W is fetched once per XCD and
    // group, A once per group.
    if (p.n_group > 0) {
        const int mts = nwg / ntn, mpx = mts >> 3, nb = p.n_group;     // launcher: mts % 8 == 0, ntn % nb == 0
        const int i = bid >> 3, per = mpx * nb;
        const int cg = i / per, r = i - cg * per;
        mt = xcd * mpx + r / nb;
        nt = cg * nb + r % nb;
    }
    const int m0 = mt * TM, n0 = (n_tile0 + nt) * TN;   // this launch covers n-tiles [n_tile0, n_tile0 + ntn)
    if constexpr (EPI == EPI_QKV && !VMODE) {
        if (p.q_tile_flag && n0 < p.H) {                // Q third, pruned last layer: nobody reads query tiles without selected rows
            const unsigned long long f8 = *reinterpret_cast<const unsigned long long*>(p.q_tile_flag + (m0 >> 5));
            if (f8 == 0ull) return;                     // workgroup-uniform, before any barrier
        }
    }

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W);

    // DMA map (one wave instruction lands 16 rows x 64 B): wave w moves rows [32w + 16i, +16), i = 0,1, of the A tile and
    // of the W tile.  Lane L lands at (row L>>2, chunk L&3) and FETCHES chunk (L&3) ^ swz4(row).
    const int lrow = lane >> 2, lch = lane & 3;
    const T* ga[2];
    const T* gw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + lrow;
        const int ch = lch ^ swz4(row);
        ga[i] = A + (size_t)(m0 + row) * (GS ? 2 * K : K) + ch * 8;
        gw[i] = W + (size_t)(n0 + row) * (GS ? 2 * K : K) + ch * 8;
    }
    auto stage = [&](int st) {
        unsigned char* sa = smem256 + (st & (NSLOT - 1)) * STAGE + (wave * 32) * ROWB;
        unsigned char* sw = sa + TM * ROWB;
        size_t oa, ow;                          // element offsets of this step's 32 k-values in the A / W rows
        if constexpr (GS) {
            // two stages per 32-group: (a_hi, w_lo), then (a_lo, w_hi); the a_hi fragments stay in registers for the second one
            oa = (size_t)(st >> 1) * 64 + ((st & 1) ? 32 : 0);
            ow = (size_t)(st >> 1) * 64 + ((st & 1) ? 0 : 32);
        } else oa = ow = (size_t)st * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            glds16(ga[i] + oa, sa + i * 16 * ROWB);
            glds16(gw[i] + ow, sw + i * 16 * ROWB);
        }
    };

    constexpr bool vmode = VMODE;          // V third of the fused QKV projection: transposed output
    f32x4 acc[8][4];      // [mi][ni] (lane = m) or, in vmode, the same slots with lane = n
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row r (+16 i keeps (r>>2)&3), logical chunk g -> physical g ^ swz4(r)
    const int aoff = (wm * 128 + r16) * ROWB + ((g ^ swz4(r16)) * 16);
    const int boff = TM * ROWB + (wn * 64 + r16) * ROWB + ((g ^ swz4(r16)) * 16);
    const int nk = GS ? 2 * (K / 32) : K / 32;
    const bool late = wm == 1;            // group 1 runs half a step behind group 0 (one extra barrier up front)
    frag_t af[8], bf[4];

    // Every wave executes the SAME two-phase step; the stagger is purely temporal:
    //   phase A_s : issue DMA(s+3), read the fragments of step s, lgkmcnt(0), counted vmcnt  | barrier
    //   phase B_s : 32 MFMAs from registers                                                    | barrier
    // Group 1 enters the loop one barrier later, so while one wave of a SIMD is in B (matrix pipe) its partner is in A
    // (LDS reads + DMA issue).  Time slot of A_s is 2s for group 0 and 2s+1 for group 1.
    //   RAW: step s+1 is first read at slot 2s+2; every wave makes its own pieces of step s+1 land (vmcnt) before the
    //        barrier that ends its A_s (slot <= 2s+1).
    //   WAR: DMA(s+3) reuses the slot of step s-1, last read at slot 2s-1 with lgkmcnt(0) before that slot's barrier.
    if constexpr (FL) {
        constexpr int LINE = 128;                  // bytes per row and group
        const int rs = GS ? 2 * K : K;             // row stride in halves
        const int ng = GS ? K / 32 : K / 64;       // groups
        // DMA map: a wave-instruction lands 8 rows x 128 B; lane L lands at (row L >> 3, physical chunk L & 7) and fetches logical
        // chunk (L & 7) ^ ((row >> 1) & 7).  Wave w moves rows [32 w + 8 i, + 8), i = 0..3, of the A tile and of the W tile; rows
        // 16 apart share the swizzle, so two lane pointers per operand serve the four pieces.
        const int lrow8 = lane >> 3, pch = lane & 7;
        const T* fa[2];
        const T* fw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 32 + i * 8 + lrow8;
            const int lc = pch ^ ((row >> 1) & 7);
            fa[i] = A + (size_t)(m0 + row) * rs + lc * 8;
            fw[i] = W + (size_t)(n0 + row) * rs + lc * 8;
        }
        auto stage_fl = [&](int grp) {
            unsigned char* sa = smem256 + ((2 * grp) & (NSLOT - 1)) * STAGE + (wave * 32) * LINE;
            unsigned char* sw = smem256 + ((2 * grp + 1) & (NSLOT - 1)) * STAGE + (wave * 32) * LINE;
            const size_t o = (size_t)grp * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(fa[i & 1] + (size_t)(i >> 1) * 16 * rs + o, sa + i * 8 * LINE);
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(fw[i & 1] + (size_t)(i >> 1) * 16 * rs + o, sw + i * 8 * LINE);
        };
        // fragment read offsets inside a slot: row (.. + r16), logical chunk g (first half) or 4 + g (second half)
        const int hsw = (r16 >> 1) & 7;
        const int a_h0 = (wm * 128 + r16) * LINE + ((g ^ hsw) * 16), a_h1 = (wm * 128 + r16) * LINE + (((4 + g) ^ hsw) * 16);
        const int b_h0 = (wn * 64 + r16) * LINE + ((g ^ hsw) * 16), b_h1 = (wn * 64 + r16) * LINE + (((4 + g) ^ hsw) * 16);
        frag_t a0[8], a1[8], bq[4];
        stage_fl(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // group 0 is in LDS for everyone
        if (wm == 1) __builtin_amdgcn_s_barrier(); // the stagger
        unsigned long long seg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
        const unsigned long long clk0 = DIAG ? __builtin_amdgcn_s_memtime() : 0, rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0;
        t_loop0 = clk0;
        auto stamp = [&](int k) __attribute__((always_inline)) {      // time since the previous stamp goes to segment k (k < 0: start)
            if constexpr (DIAG) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (k >= 0) seg[k] += t - tlast;
                tlast = t;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto sub = [&](const int s, const int odd) __attribute__((always_inline)) {
            // ---- phase A ----
            stamp(-1);
            if (!odd && s + 1 < ng) stage_fl(s + 1);
            stamp(5 * odd + 0);                    // seg 0: DMA issue
            {
                const unsigned char* sa = smem256 + ((2 * s) & (NSLOT - 1)) * STAGE;
                const unsigned char* sw = smem256 + ((2 * s + 1) & (NSLOT - 1)) * STAGE;
                // GS: E pairs a_hi with w_lo, O a_lo with w_hi (+ a_hi w_hi); 16-bit rows: E = first 32 k of both, O = second 32 k
                const int bo = GS ? (odd ? b_h0 : b_h1) : (odd ? b_h1 : b_h0);
                const int ao = odd ? a_h1 : a_h0;
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const frag_t*>(sw + bo + j * 16 * LINE);
                if (!odd) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) a0[i] = *reinterpret_cast<const frag_t*>(sa + ao + i * 16 * LINE);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) a1[i] = *reinterpret_cast<const frag_t*>(sa + ao + i * 16 * LINE);
                }
            }
            if (odd) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            stamp(5 * odd + 1);                    // seg 1: fragment reads + waits
            __builtin_amdgcn_s_barrier();
            stamp(5 * odd + 2);                    // seg 2: barrier after phase A
            // ---- phase B ----
            __builtin_amdgcn_s_setprio(1);
            if (odd) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { if (!VMODE) mma16(bq[j], a1[i], acc[i][j]); else mma16(a1[i], bq[j], acc[i][j]); }
                }
            }
            if (!odd || GS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { if (!VMODE) mma16(bq[j], a0[i], acc[i][j]); else mma16(a0[i], bq[j], acc[i][j]); }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            stamp(5 * odd + 3);                    // seg 3: MFMA issue
            __builtin_amdgcn_s_barrier();
            stamp(5 * odd + 4);                    // seg 4: barrier after phase B
        };
        for (int s = 0; s < ng; ++s) { sub(s, 0); sub(s, 1); }
        if (wm == 0) __builtin_amdgcn_s_barrier();   // pairs with the late group's last barrier
        if constexpr (DIAG) {
            if (p.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {     // 64 workgroups of one XCD
                unsigned long long* o = p.stamps + ((size_t)(blockIdx.x >> 3) * 8 + wave) * 12;
                for (int k = 0; k < 10; ++k) o[k] = seg[k];
                const unsigned long long dc = __builtin_amdgcn_s_memtime() - clk0, dr = __builtin_amdgcn_s_memrealtime() - rt0;
                o[10] = dr ? dc * 1000 / dr : 0; o[11] = ng;
            }
            t_loop1 = __builtin_amdgcn_s_memtime();
        }
    } else {
    stage(0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // P: step 0 is in LDS for everyone
    if (late) __builtin_amdgcn_s_barrier();    // the stagger
    if constexpr (GS) {
        // Group-split rows, two ring stages per 32-group instead of one per product: stage 2s brings (a_hi, w_lo) -> 32 MFMAs a_hi*w_lo,
        // stage 2s+1 brings (a_lo, w_hi) -> 64 MFMAs a_lo*w_hi + a_hi*w_hi with the a_hi fragments kept in 32 registers: a third less
        // LDS-DMA traffic and a third fewer load phases / barriers for the same 96 MFMAs (measured -10 % per GEMM at c3).
        frag_t ah[8];
        auto substep = [&](const int st, const int odd) __attribute__((always_inline)) {
            // ---- phase A ----
            if (st + 3 < nk) stage(st + 3);
            {
                const unsigned char* sb = smem256 + (st & (NSLOT - 1)) * STAGE;
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const frag_t*>(sb + boff + j * 16 * ROWB);
                if (!odd) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) ah[i] = *reinterpret_cast<const frag_t*>(sb + aoff + i * 16 * ROWB);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const frag_t*>(sb + aoff + i * 16 * ROWB);
                }
                if constexpr (PD) {
                    if (!odd && (p.prec & 2)) {      // W rounded to f16: the a_hi * w_lo term vanishes
#pragma unroll
                        for (int j = 0; j < 4; ++j) bf[j] = (frag_t)(T)0;
                    }
                    if (odd && (p.prec & 1)) {       // A rounded to f16: the a_lo * w_hi term vanishes
#pragma unroll
                        for (int i = 0; i < 8; ++i) af[i] = (frag_t)(T)0;
                    }
                }
            }
            if (st + 3 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (st + 2 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- phase B ----
            __builtin_amdgcn_s_setprio(1);
            if (odd) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { if (!vmode) mma16(bf[j], af[i], acc[i][j]); else mma16(af[i], bf[j], acc[i][j]); }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { if (!vmode) mma16(bf[j], ah[i], acc[i][j]); else mma16(ah[i], bf[j], acc[i][j]); }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        for (int st = 0; st < nk; st += 2) { substep(st, 0); substep(st + 1, 1); }
    } else
    for (int st = 0; st < nk; ++st) {
        // ---- phase A ----
        if (st + 3 < nk) stage(st + 3);
        {
            const unsigned char* sb = smem256 + (st & (NSLOT - 1)) * STAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const frag_t*>(sb + boff + j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const frag_t*>(sb + aoff + i * 16 * ROWB);
        }
        // my pieces of step st+1 have landed once only the (up to two) younger steps are outstanding
        if (st + 3 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (st + 2 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase B ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!vmode) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(bf[j], af[i], acc[i][j]);   // D[n = 4g+r][m = r16]
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(af[i], bf[j], acc[i][j]);   // D[m = 4g+r][n = r16]
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (!late) __builtin_amdgcn_s_barrier();   // pairs with group 1's last barrier
    }

    // ---------------- epilogue ----------------
    // The accumulator layout gives a lane 4 consecutive columns of one row (8-byte pieces).  Storing that
    // directly costs 32 narrow, scattered stores per lane — as much time as a K=768 main loop.  Instead each
    // wave stages its 128x64 sub-tile through a PRIVATE fp32 LDS patch, 32 rows at a time (the stage ring
    // is dead after the last barrier), and writes it back as 16-byte pieces: one wave instruction = 8 rows
    // x 128 contiguous bytes (row-major outputs) or whole 16-byte fragment units (Q / K / V^T).  Bias and
    // GELU are applied before staging, the residual is added in fp32 at the store, so the value is rounded
    // once, exactly as before.  Wave-local LDS ordering only; no workgroup barrier.
    typedef __attribute__((ext_vector_type(8))) T vec8T;
    const float* __restrict__ bias = p.bias;
    float* stg = reinterpret_cast<float*>(smem256 + wave * EPI_PATCH);
    const int qkv_b0 = (EPI == EPI_QKV) ? m0 / p.Sp : 0;
    if constexpr (EPI == EPI_SWIGLU) {
        // W rows alternate 16 gate features / 16 up features (engine.hip interleaves them at load), so the accumulator
        // blocks j = 0,2 hold gate and j = 1,3 the matching up columns of the SAME 16 features: the product needs no
        // exchange.  The wave's 128x64 sub-tile becomes 128x32 outputs (Q2:47 silu(gate(x)) * up(x)); patch [32 rows][32
        // features], row stride 36 floats; 16-byte stores of 8 features.
        const int I = N >> 1;
        // RMSNorm folded into this GEMM (GemmArgs::a_stats: W holds W diag(gamma), the rows are raw): gate and up scale by the row's rstd
        const bool lnf = GS && p.a_stats != nullptr;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const float rs = lnf ? p.a_stats[m0 + wm * 128 + (2 * c + ii) * 16 + r16].y : 1.0f;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    f32x4 gt = acc[2 * c + ii][2 * jj], up = acc[2 * c + ii][2 * jj + 1];
                    if (lnf) { gt *= rs; up *= rs; }
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = gt[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * gt[r])) * up[r];
                    *reinterpret_cast<f32x4*>(stg + (ii * 16 + r16) * 36 + jj * 16 + 4 * g) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = lane + 64 * k, row = idx >> 2, g4 = idx & 3;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * 36 + g4 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * 36 + g4 * 8 + 4);
                vec8T o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (T)lo[e]; o[4 + e] = (T)hi[e]; }
                const int m = m0 + wm * 128 + c * 32 + row;
                if constexpr (GS) {         // group-split output row of I features
                    vec8T ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ol[e] = (T)(lo[e] - (float)o[e]); ol[4 + e] = (T)(hi[e] - (float)o[4 + e]); }
                    const int f = (n0 >> 1) + wn * 32 + g4 * 8;
                    T* cp = reinterpret_cast<T*>(p.C) + (size_t)m * 2 * I + (f >> 5) * 64 + (f & 31);
                    *reinterpret_cast<vec8T*>(cp) = o;
                    *reinterpret_cast<vec8T*>(cp + 32) = ol;
                } else
                *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.C) + (size_t)m * I + (n0 >> 1) + wn * 32 + g4 * 8) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    } else if (!vmode) {
        // D[n = 16j + 4g + r][m = 16i + r16]; patch [32 rows m][64 cols n], row stride 68 floats
        const int which = (EPI == EPI_QKV) ? n0 / p.H : 0;
        float bj[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (bias) { const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0 + wn * 64 + j * 16 + 4 * g); bj[j][0] = bv[0]; bj[j][1] = bv[1]; bj[j][2] = bv[2]; bj[j][3] = bv[3]; }
            else { bj[j][0] = bj[j][1] = bj[j][2] = bj[j][3] = 0.f; }
        }
        // LayerNorm folded into this GEMM (GemmArgs::a_stats): the accumulator is raw_row . (W diag(gamma))^T; the lane's 8 accumulator
        // rows are m0 + 128 wm + 16 i + r16
        const bool lnf = EPI != EPI_RESID && p.a_stats != nullptr;
        float cj[4][4];
        float2 st_i[8];
        if constexpr (EPI != EPI_RESID) {
            if (lnf) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {       // (RMSNorm: no mean, no ln_c)
                    const f32x4 cv = p.ln_c ? *reinterpret_cast<const f32x4*>(p.ln_c + n0 + wn * 64 + j * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    cj[j][0] = cv[0]; cj[j][1] = cv[1]; cj[j][2] = cv[2]; cj[j][3] = cv[3];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) st_i[i] = p.a_stats[m0 + wm * 128 + i * 16 + r16];
            }
        }
        // EPI_RESID with GemmArgs::r_stats: the residual rows are raw, LayerNorm is applied on the fly; this lane's 8 columns
        float rg[8], rb[8];
        const bool rln = EPI == EPI_RESID && p.r_stats != nullptr;
        const bool gsout = EPI == EPI_RESID && p.ln_part != nullptr;     // raw rows (group-split, or T in the 16-bit modes) + statistics partials out; lane map: 8 consecutive columns
        if constexpr (EPI == EPI_RESID) {
            if (rln) {
                const int nb = n0 + wn * 64 + ((gsout || !GS) ? (lane & 7) * 8 : (lane & 7) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rg[e] = p.r_gamma[nb + e]; rb[e] = p.r_beta[nb + e];
                    rg[4 + e] = p.r_gamma[nb + ((gsout || !GS) ? 4 : 32) + e]; rb[4 + e] = p.r_beta[nb + ((gsout || !GS) ? 4 : 32) + e];
                }
            }
        }
        // residual rows are fetched one 32-row chunk AHEAD of their use (16-byte coalesced loads): without this each
        // chunk exposed a full HBM round trip between its LDS read-back and its store (+4.7 us per tile measured)
        vec8T rpre[4], rpre_lo[GS ? 4 : 1];
        float2 rst_pre[4];
        auto load_resid = [&](int c, vec8T (&r)[4], vec8T (&rl)[GS ? 4 : 1], float2 (&rst)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, row = idx >> 3, g8 = idx & 7;
                if (rln) rst[k] = p.r_stats[m0 + wm * 128 + c * 32 + row];
                if constexpr (GS) {
                    if (gsout) {                 // 8 consecutive columns per lane (GS residual rows only: the fused pipeline never mixes in plain ones)
                        const int n = n0 + wn * 64 + g8 * 8;
                        const T* rp = reinterpret_cast<const T*>(p.resid) + (size_t)(m0 + wm * 128 + c * 32 + row) * 2 * N + (n >> 5) * 64 + (n & 31);
                        r[k] = *reinterpret_cast<const vec8T*>(rp);
                        rl[k] = *reinterpret_cast<const vec8T*>(rp + 32);
                        if constexpr (PD) { if (p.prec & 4) rl[k] = (vec8T)(T)0; }      // residual stream rounded to f16
                        continue;
                    }
                    // "wide" lane map of the fp32-row epilogue: a lane owns columns [4 g8, +4) and [32 + 4 g8, +4) of the wave's 64, so that
                    // each store / load instruction covers whole 128-byte row segments (8 floats per lane left 16-byte holes: +8 % per GEMM)
                    const int na = n0 + wn * 64 + g8 * 4;
                    typedef __attribute__((ext_vector_type(4))) T vec4T;
                    if (p.gs_resid_plain) {      // plain fp32 residual row: its 2 x 4 floats ride in the two 16-byte registers
                        const float* rp = reinterpret_cast<const float*>(p.resid) + (size_t)(m0 + wm * 128 + c * 32 + row) * N + na;
                        r[k] = __builtin_bit_cast(vec8T, *reinterpret_cast<const f32x4*>(rp));
                        rl[k] = __builtin_bit_cast(vec8T, *reinterpret_cast<const f32x4*>(rp + 32));
                    } else {                     // group-split residual row: r = [4 hi of a | 4 hi of b], rl = the lo halves
                        const T* rp = reinterpret_cast<const T*>(p.resid) + (size_t)(m0 + wm * 128 + c * 32 + row) * 2 * N + (na >> 5) * 64 + (na & 31);
                        const vec4T ha = *reinterpret_cast<const vec4T*>(rp), la = *reinterpret_cast<const vec4T*>(rp + 32);
                        const vec4T hb = *reinterpret_cast<const vec4T*>(rp + 64), lb = *reinterpret_cast<const vec4T*>(rp + 96);
                        r[k] = (vec8T){ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]};
                        rl[k] = (vec8T){la[0], la[1], la[2], la[3], lb[0], lb[1], lb[2], lb[3]};
                        if constexpr (PD) { if (p.prec & 4) rl[k] = (vec8T)(T)0; }
                    }
                } else {
                    r[k] = *reinterpret_cast<const vec8T*>(reinterpret_cast<const T*>(p.resid) +
                                                           (size_t)(m0 + wm * 128 + c * 32 + row) * N + n0 + wn * 64 + g8 * 8);
                }
            }
        };
        if (EPI == EPI_RESID) load_resid(0, rpre, rpre_lo, rst_pre);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            vec8T rcur[4], rcur_lo[GS ? 4 : 1];
            float2 rst_cur[4];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { rcur[k] = rpre[k]; rst_cur[k] = rst_pre[k]; if constexpr (GS) rcur_lo[k] = rpre_lo[k]; }
                if (c + 1 < 4) load_resid(c + 1, rpre, rpre_lo, rst_pre);
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[2 * c + ii][j];
                    if constexpr (EPI != EPI_RESID) {
                        if (lnf) {
                            const float2 sm = st_i[2 * c + ii];
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = sm.y * (v[r] - sm.x * cj[j][r]);
                        }
                    }
                    v[0] += bj[j][0]; v[1] += bj[j][1]; v[2] += bj[j][2]; v[3] += bj[j][3];
                    if (EPI == EPI_GELU) {
                        if constexpr (GS) {     // fp32 mode: the erf form at fp32 resolution (the 16-bit epilogue's logistic fit is only f16-exact)
                            const f32x2 g0 = glc_gelu2_f32((f32x2){v[0], v[1]}), g1 = glc_gelu2_f32((f32x2){v[2], v[3]}); v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
                        } else {
                            const f32x2 g0 = glc_gelu2((f32x2){v[0], v[1]}), g1 = glc_gelu2((f32x2){v[2], v[3]});
                            v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
                        }
                    }
                    *reinterpret_cast<f32x4*>(stg + (ii * 16 + r16) * 68 + j * 16 + 4 * g) = v;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, row = idx >> 3, g8 = idx & 7;
                const bool wide = GS && EPI == EPI_RESID && !gsout;      // fp32 row output: see load_resid
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + row * 68 + (wide ? g8 * 4 : g8 * 8));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + row * 68 + (wide ? 32 + g8 * 4 : g8 * 8 + 4));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const int m = m0 + wm * 128 + c * 32 + row;
                const int n = n0 + wn * 64 + (wide ? g8 * 4 : g8 * 8);
                if (EPI == EPI_RESID) {
                    if (GS && p.gs_resid_plain) {
                        const f32x4 ra = __builtin_bit_cast(f32x4, rcur[k]), rb4 = __builtin_bit_cast(f32x4, rcur_lo[GS ? k : 0]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += ra[e]; v[4 + e] += rb4[e]; }
                    } else if (rln) {           // raw residual row: LayerNorm on the fly
                        const float2 sm = rst_cur[k];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float r = GS ? (float)rcur[k][e] + (float)rcur_lo[GS ? k : 0][e] : (float)rcur[k][e];
                            v[e] += (r - sm.x) * sm.y * rg[e] + rb[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += GS ? (float)rcur[k][e] + (float)rcur_lo[GS ? k : 0][e] : (float)rcur[k][e];
                    }
                }
                if constexpr (EPI == EPI_RESID) {
                    if (gsout) {
                        // raw GS row out + this 64-column block's (sum, sum of squares) of the row: the 8 lanes of a row are consecutive
                        // (sum, M2 = sum of squared deviations from THIS block's mean): merged across blocks by Chan's formula in
                        // ln_stats_kernel — no E[x^2] - mean^2 cancellation whatever the row's mean
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
                        vec8T oh, ol2;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { oh[e] = (T)v[e]; ol2[e] = (T)(v[e] - (float)oh[e]); }
                        if constexpr (GS) {
                            T* cp = reinterpret_cast<T*>(p.C) + (size_t)m * 2 * N + (n >> 5) * 64 + (n & 31);
                            *reinterpret_cast<vec8T*>(cp) = oh;
                            *reinterpret_cast<vec8T*>(cp + 32) = ol2;
                        } else *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.C) + (size_t)m * N + n) = oh;      // 16-bit modes: the raw row in T
                        continue;
                    }
                }
                vec8T o, ol;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = (T)v[e]; if constexpr (GS) ol[e] = (T)(v[e] - (float)o[e]); }
                if (EPI == EPI_QKV) {
                    if (m < p.Mvalid) {
                        int b = qkv_b0, sq = m - qkv_b0 * p.Sp;
                        while (sq >= p.Sp) { sq -= p.Sp; ++b; }          // a 256-row tile spans <= 5 sequences (Sp >= 64)
                        const int nn = n - which * p.H, hh = nn >> 6, dd = nn & 63;     // dd is a multiple of 8: one 16-B unit
                        const int bh = b * p.nh + hh;
                        const size_t off = which == 0 ? glc_qoff(p.Sp, bh, sq, dd) : glc_koff(p.Sp, bh, sq, dd);
                        T* base = reinterpret_cast<T*>(which == 0 ? p.Qh : p.Kh);
                        if constexpr (GS) {         // split-f16 unit: the 8 elements' 32 bytes are [8 hi | 8 lo]
                            *reinterpret_cast<vec8T*>(base + 2 * off) = o;
                            *reinterpret_cast<vec8T*>(base + 2 * off + 8) = ol;
                        } else *reinterpret_cast<vec8T*>(base + off) = o;
                    }
                } else if constexpr (GS) {
                    if (EPI == EPI_RESID || p.gs_c_plain) {         // plain fp32 row (LayerNorm input; decoder: QKV / gate|up rows)
                        float* cp = reinterpret_cast<float*>(p.C) + (size_t)m * N + n;
                        *reinterpret_cast<f32x4*>(cp) = (f32x4){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(cp + (wide ? 32 : 4)) = (f32x4){v[4], v[5], v[6], v[7]};
                    } else {                        // GS row
                        T* cp = reinterpret_cast<T*>(p.C) + (size_t)m * 2 * N + (n >> 5) * 64 + (n & 31);
                        *reinterpret_cast<vec8T*>(cp) = o;
                        *reinterpret_cast<vec8T*>(cp + 32) = ol;
                    }
                } else {
                    *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.C) + (size_t)m * N + n) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // V third: D[m = 16i + 4g + r][n = 16j + r16]; patch [64 rows dd][32 cols key], row stride 36 floats
        float bn[4], cn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) bn[j] = bias ? bias[n0 + wn * 64 + j * 16 + r16] : 0.f;
        const bool lnf = p.a_stats != nullptr;             // LayerNorm folded into this GEMM (GemmArgs::a_stats)
        if (lnf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cn[j] = p.ln_c ? p.ln_c[n0 + wn * 64 + j * 16 + r16] : 0.f;
        }
        const int hh = (n0 + wn * 64 - 2 * p.H) >> 6;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[2 * c + ii][j];
                    if (lnf) {      // accumulator rows m0 + 128 wm + 16 (2c + ii) + 4g + r
                        const float2* sp = p.a_stats + m0 + wm * 128 + (2 * c + ii) * 16 + 4 * g;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float2 sm = sp[r]; v[r] = sm.y * (v[r] - sm.x * cn[j]); }
                    }
                    v[0] += bn[j]; v[1] += bn[j]; v[2] += bn[j]; v[3] += bn[j];
                    *reinterpret_cast<f32x4*>(stg + (j * 16 + r16) * 36 + ii * 16 + 4 * g) = v;
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
                    if constexpr (GS) { ol[e] = (T)(lo[e] - (float)o[e]); ol[4 + e] = (T)(hi[e] - (float)o[4 + e]); }
                }
                const int m = m0 + wm * 128 + c * 32 + kg * 8;           // first of 8 consecutive keys
                if (m < p.Mvalid) {
                    int b = qkv_b0, sq = m - qkv_b0 * p.Sp;
                    while (sq >= p.Sp) { sq -= p.Sp; ++b; }
                    const size_t off = glc_voff(p.Sp, b * p.nh + hh, dd, sq);
                    if constexpr (GS) {
                        *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.Vt) + 2 * off) = o;
                        *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.Vt) + 2 * off + 8) = ol;
                    } else *reinterpret_cast<vec8T*>(reinterpret_cast<T*>(p.Vt) + off) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tile's stores have left
        if (p.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {
            unsigned long long* o = p.stamps + 64 * 8 * 12 + ((size_t)(blockIdx.x >> 3) * 8 + wave) * 2;
            o[0] = t_loop0 - t_entry;
            o[1] = __builtin_amdgcn_s_memtime() - t_loop1;
        }
    }
}

// full-line stages (kernel header): default on; GLC_GEMM_FL=0 selects the half-line loop (developer A/B switch; same results bit for bit)
static std::atomic<int> g_full_lines{-1};
static bool use_full_lines() {
    int v = g_full_lines.load(std::memory_order_relaxed);
    if (v < 0) { v = (glc_dev_env("GLC_GEMM_FL") == nullptr || atoi(glc_dev_env("GLC_GEMM_FL")) != 0) ? 1 : 0; g_full_lines.store(v, std::memory_order_relaxed); }
    return v != 0;
}
template <typename T, int EPI, bool VMODE, bool GS = false, bool PD = false, bool FL = false, bool DIAG = false> const char* launch_e(hipStream_t st, const GemmArgs& a, int n_tile0, int ntn) {
    if constexpr (GS && !PD && !FL) { if (a.prec) return launch_e<T, EPI, VMODE, GS, true, false>(st, a, n_tile0, ntn); }
    if constexpr (!PD && !FL) { if (use_full_lines() && (GS || a.K % 64 == 0)) return launch_e<T, EPI, VMODE, GS, false, true>(st, a, n_tile0, ntn); }
#ifdef GLC_DEVELOPER      // the stamped build: developer libraries only
    if constexpr (FL && !DIAG && EPI == EPI_BIAS && !VMODE) { if (a.stamps) return launch_e<T, EPI, VMODE, GS, false, true, true>(st, a, n_tile0, ntn); }
#else
    if (a.stamps) return "gemm256s: the stamped build exists in developer builds only (make DEV=1)";
#endif
    static std::atomic<unsigned> lds_ok{0};        // per device: several engines of one process may sit on different GPUs
    if (!glc_raise_lds_limit(gemm256s_kernel<T, EPI, VMODE, GS, PD, FL, DIAG>, NSLOT * STAGE, lds_ok)) return "gemm256s: cannot raise the dynamic LDS limit";
    const int grid = (a.Mpad / TM) * ntn;
    GemmArgs b = a;
    static const int ng_env = glc_dev_env("GLC_GEMM_NGROUP") ? atoi(glc_dev_env("GLC_GEMM_NGROUP")) : -1;      // developer A/B switch: 0 = row-major tile order
    b.n_group = 0;
    if (ng_env != 0 && ntn >= 8 && (a.Mpad / TM) % 8 == 0) b.n_group = ntn % 4 == 0 ? 4 : (ntn % 3 == 0 ? 3 : 0);     // (6 N-tiles in 2 groups measured MORE fetch: A twice, W fitted anyway)
    hipLaunchKernelGGL((gemm256s_kernel<T, EPI, VMODE, GS, PD, FL, DIAG>), dim3(grid), dim3(512), NSLOT * STAGE, st, b, n_tile0, ntn);
    return nullptr;
}
template <typename T> const char* launch_t(hipStream_t st, int epi, const GemmArgs& a) {
    const int ntn = a.N / TN;
    switch (epi) {
        case EPI_BIAS: return launch_e<T, EPI_BIAS, false>(st, a, 0, ntn);
        case EPI_GELU: return launch_e<T, EPI_GELU, false>(st, a, 0, ntn);
        case EPI_RESID: return launch_e<T, EPI_RESID, false>(st, a, 0, ntn);
        case EPI_SWIGLU: return launch_e<T, EPI_SWIGLU, false>(st, a, 0, ntn);
        case EPI_QKV: {   // Q|K columns in row orientation, V columns transposed: two grids, one stream
            const int nqk = 2 * a.H / TN, nq = a.qkv_skip_q ? a.H / TN : 0;     // skip the Q columns when asked
            const char* m = launch_e<T, EPI_QKV, false>(st, a, nq, nqk - nq);
            return m ? m : launch_e<T, EPI_QKV, true>(st, a, nqk, ntn - nqk);
        }
    }
    return "gemm256s: bad epilogue";
}

}  // namespace

static bool gemm256s_supported(int dtype, const GemmArgs& a) {
    return (dtype == GLC_DT_BF16 || dtype == GLC_DT_F16) && a.Mpad > 0 && a.Mpad % TM == 0 && a.N > 0 && a.N % TN == 0 &&
           a.K > 0 && a.K % 32 == 0;
}

bool glc_gemm256s_gs_supported(const GemmArgs& a, int epi) {
    if (!(a.Mpad > 0 && a.Mpad % TM == 0 && a.N > 0 && a.N % TN == 0 && a.K > 0 && a.K % 32 == 0)) return false;
    if (epi == EPI_QKV) return a.H % 256 == 0 && a.N == 3 * a.H && a.Sp % 64 == 0 && a.Sp >= 64 && a.nh * 64 == a.H;
    return epi == EPI_BIAS || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_SWIGLU;
}

// fp32 mode on group-split operands (see the kernel header): T = f16 halves, 2K/32 ring stages for 3K/32 MFMA steps.
const char* glc_launch_gemm256s_gs(hipStream_t st, int epi, const GemmArgs& a) {
    if (!glc_gemm256s_gs_supported(a, epi)) return "gemm256s(gs): unsupported shape";
    if (!a.A || !a.W) return "gemm256s(gs): null operand";
    if (epi == EPI_QKV) { if (!a.Qh || !a.Kh || !a.Vt) return "gemm256s(gs): null QKV output"; }
    else if (!a.C) return "gemm256s(gs): null output";
    if (epi == EPI_RESID && !a.resid) return "gemm256s(gs): null residual";
    const int ntn = a.N / TN;
    switch (epi) {
        case EPI_BIAS: return launch_e<f16_t, EPI_BIAS, false, true>(st, a, 0, ntn);
        case EPI_GELU: return launch_e<f16_t, EPI_GELU, false, true>(st, a, 0, ntn);
        case EPI_RESID: return launch_e<f16_t, EPI_RESID, false, true>(st, a, 0, ntn);
        case EPI_SWIGLU: return a.bias ? "gemm256s(gs): the SwiGLU epilogue takes no bias" : launch_e<f16_t, EPI_SWIGLU, false, true>(st, a, 0, ntn);
        case EPI_QKV: {
            const int nqk = 2 * a.H / TN, nq = a.qkv_skip_q ? a.H / TN : 0;
            const char* m = launch_e<f16_t, EPI_QKV, false, true>(st, a, nq, nqk - nq);
            return m ? m : launch_e<f16_t, EPI_QKV, true, true>(st, a, nqk, ntn - nqk);
        }
    }
    return "gemm256s(gs): bad epilogue";
}

// Host-side shape contract: 16-bit T; Mpad % 256 == 0 (buffers allocated with Mpad rows), N % 256 == 0,
// K % 64 == 0; EPI_QKV: H % 256 == 0 (a tile never straddles Q|K|V), Sp % 64 == 0.
const char* glc_launch_gemm256s(hipStream_t st, int dtype, int epi, const GemmArgs& a_in) {
    const GemmArgs& a = a_in;
    if (!gemm256s_supported(dtype, a)) return "gemm256s: unsupported shape";
    if (!a.A || !a.W) return "gemm256s: null operand";
    if (epi == EPI_QKV) {
        if (a.H % 256 || a.N != 3 * a.H || a.Sp % 64 || a.Sp < 64 || !a.Qh || !a.Kh || !a.Vt || a.nh * 64 != a.H) return "gemm256s: bad QKV epilogue shape";
    } else if (!a.C) return "gemm256s: null output";
    if (epi == EPI_RESID && !a.resid) return "gemm256s: null residual";
    if (epi == EPI_SWIGLU && a.bias) return "gemm256s: the SwiGLU epilogue takes no bias";
    return dtype == GLC_DT_BF16 ? launch_t<bf16_t>(st, epi, a) : launch_t<f16_t>(st, epi, a);
}

void glc_gemm_set_full_lines(int on) { g_full_lines.store(on ? 1 : 0, std::memory_order_relaxed); }

bool glc_gemm_small_m(const GemmArgs& a) {
    static const int mode = glc_dev_env("GLC_GEMM_SMALL_M") ? atoi(glc_dev_env("GLC_GEMM_SMALL_M")) : 1;     // developer A/B switch (0 = off)
    if (!mode) return false;
    const int ncu = glc_device_cus();
    return (long long)(a.Mpad / TM) * (a.N / TN) * 2 < ncu;
}

bool glc_gemm256_supported(int dtype, const GemmArgs& a) {
    return (dtype == GLC_DT_BF16 || dtype == GLC_DT_F16) && a.Mpad > 0 && a.Mpad % TM == 0 && a.N > 0 && a.N % TN == 0 && a.K > 0 && a.K % 64 == 0;
}
