// DeBERTa-v2/v3 disentangled self-attention (modeling_deberta_v2.py:229-345), re-derived for gfx950.
//
// Algebra.  With share_att_key, pos_att_type = c2p|p2c and rel(q,k) = bucket(q-k) (an odd function,
// :57-69), both relative-position terms use ONE index that depends on q-k only:
//     delta(q-k) = clamp(bucket(q-k) + span, 0, 2 span - 1)
//     score[q,k] = ( Q_q.K_k  +  Q_q.PK[delta(q-k)]  +  K_k.PQ[delta(q-k)] ) / sqrt(3 d)
// (c2p: :315-324, p2c: :327-343 — the p2c gather index clamp(-bucket(k-q)+span) equals delta(q-k)).
// log2(e)/sqrt(3d) is folded into Wq at load time, so Q and PQ = query_proj(rel) arrive pre-scaled and
// the scores are in log2 units (softmax = exp2).
// Key padding (masked_fill(finfo.min), :256-257) is an additive per-key bias of -1e30; rows whose
// query is padding are don't-care (they never feed a valid row).
//
// attn_band_kernel (16-bit operands): one wave owns a 32-query tile and walks 32-key tiles with an
// online softmax, all on 32x32x16 MFMA:
//   S^T  = K_tile Q^T                    (keys on rows -> each lane holds 16 keys of ONE query)
//   c2p  : B^T[rr][q]  = PK[delta(rmin+rr)] . Q_q     rr = 0..63  ("Toeplitz band": rr = q-k-rmin)
//   p2c  : B [rr][k]   = PQ[delta(rmin+rr)] . K_k
//   O^T += V^T_tile P^T                  (P^T taken straight from the S^T accumulators)
// The position operand rows are gathered by delta at fragment-load time, so the band is indexed by
// the relative distance itself and the per-element lookup is pure arithmetic (rr = c - kk + 31).
// Both bands go through a wave-private LDS scratch (2 x 32 x 68 floats) — written as b128 rows,
// read back as conflict-free b32 gathers; no workgroup barrier anywhere in the kernel.
// The band SLIDES: stepping one key tile lowers rmin by 32, so the upper 32-row block of tile kt+1 is
// the lower block of tile kt.  Per key tile only one new block of PK/PQ rows is loaded, the c2p band
// (a function of (rel, q) only) is computed for that block alone and kept in a 2-slot LDS ring,
// and the next tile's K / PK / PQ fragments are loaded under the current tile's work (in the default build straight
// into the registers of the operands that have just issued their last MFMA: one K set, two PQ sets).
// Tried and rejected (same-box A/B, see git history "attn_band2"): one wave owning two query tiles at one
// wave per SIMD (shared K/V/PQ loads, 36 instead of 40 MFMA per tile pair) — bit-identical results but 33 %
// slower: hipcc's schedule does not overlap the two tiles well enough to replace two-wave TLP.
// Key rows are loaded in the order pi(r) = swap(bit2,bit3) so that the accumulator-as-operand k
// permutation of the P*V MFMA lines up with 8 CONTIGUOUS keys of V^T (one 16-B load per lane).
#include <stdlib.h>
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

constexpr float RESCALE_THR = 8.0f;   // log2 units
constexpr int LROW = 68;  // floats per LDS band row (64 + 4 pad: 16-B aligned rows, odd multiple of 4 banks)

__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Diagnostic build (DIAG): s_memtime stamps at the segment boundaries of a band tile, summed per wave.
template <bool ON> struct SegClock {
    unsigned prev = 0, seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tiles = 0;
    bool live = false;          // only band tiles are accounted (softmax_pv is shared with the saturated tiles)
    __device__ __forceinline__ static unsigned now() {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return (unsigned)t;
    }
    __device__ __forceinline__ void start() { if constexpr (ON) { prev = now(); live = true; ++tiles; } }
    __device__ __forceinline__ void stop() { if constexpr (ON) live = false; }
    __device__ __forceinline__ void mark(int k) { if constexpr (ON) { if (live) { const unsigned t = now(); seg[k] += t - prev; prev = t; } } }
};

// fp32 (T = float, the parity-grade mode) runs the same code on 32x32x2 fp32 MFMAs; its fragments are twice as large, so it is
// built for one wave per SIMD (512 VGPRs) with the rolled band loop.
// SPLIT (T = float): the operand units hold split-f16 pairs (glc_common.h f16x8s) — same addresses, same code, every product is
// three f16 32x32x16 MFMAs (3 x 32 cycles) instead of eight fp32 32x32x2 MFMAs (8 x 64 cycles); the probabilities are split on
// the fly before P*V.  Error class of the fp32 kernel (tests: same 1e-4 bound), operands must be inside the f16 range.
template <bool SPLIT, typename T> struct BandFrag { typedef typename AFrag<T>::type type; };
template <typename T> struct BandFrag<true, T> { typedef f16x8s type; };
template <typename T, bool UNROLL6 /* the lean two-step band loop (default for 16-bit T); false = rolled loop */, bool DIAG = false, bool SPLIT = false>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 1 : 2) void attn_band_kernel(AttnArgs a) {
    static_assert(!SPLIT || sizeof(T) == 4, "split operands live in the fp32 layouts");
    typedef typename BandFrag<SPLIT, T>::type frag_t;
    constexpr int UNITB = 512 * (int)sizeof(T);      // bytes of one fragment unit (64 lanes x 8 elements)
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * 32 * LROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int Sp = a.Sp;
    // XCD-aware decode of the 1-D grid: workgroups are dealt round-robin to the 8 XCDs, so give all
    // query blocks of one (batch, head) the same id % 8 — they then share one L2 for that head's K / V^T
    // (speed only; any placement is correct).
    const int nqb = (Sp + 127) >> 7;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int bh = xcd + 8 * (jj / nqb);
    int q0 = ((jj % nqb) * 4 + wave) * 32;
    if (bh >= a.B * a.nh) return;              // whole workgroup leaves
    const int b = bh / a.nh, hh = bh - b * a.nh;
    // Pruned last layer, cooperative form (a.ksplit): when exactly ONE of this workgroup's four query tiles holds selected rows — the usual
    // case: [CLS] and the class tokens sit at the head of the sequence — the four waves share that tile and split its KEYS four ways
    // (contiguous quarters of the key tiles), then merge their (m, l, O) partials through LDS.  A wave walking all key tiles of its tile
    // alone, one wave per SIMD and nothing to overlap with, was half of the pruned layer's time.
    int kpart = -1;                            // >= 0: this wave's quarter
    if (a.tile_flag && a.ksplit) {
        const unsigned char* tf = a.tile_flag + (size_t)b * (Sp >> 5) + (jj % nqb) * 4;
        int cnt = 0, sel = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) if (((jj % nqb) * 4 + w) * 32 < Sp && tf[w]) { ++cnt; sel = w; }
        if (cnt == 0) return;                  // workgroup-uniform
        if (cnt == 1) { q0 = ((jj % nqb) * 4 + sel) * 32; kpart = wave; }
    }
    if (q0 >= Sp) return;                      // whole wave leaves; no workgroup barriers below (cooperative form: q0 < Sp for all four)
    if (kpart < 0 && a.tile_flag && !a.tile_flag[(size_t)b * (Sp >> 5) + (q0 >> 5)]) return;   // pruned last layer: no selected row in this query tile
    if (q0 >= a.klen[b] && q0 > 0) {
        // every query of this tile lies past the row's last attended token: a padding-only tile of a ragged batch.  Its
        // output never reaches an attended row (those keys are masked), so skip the work and store zeros (finite).
        if (kpart > 0) return;                 // cooperative form: one wave stores
        T* outz = reinterpret_cast<T*>(a.CTX) + ((size_t)b * Sp + q0 + c) * a.H + hh * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            store4<T>(outz + 8 * g + 4 * h, 0.f, 0.f, 0.f, 0.f);
            store4<T>(outz + 32 + 8 * g + 4 * h, 0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

    // fragment-major operands (glc_layout.h): one tile = 4 (or 2x2) units of 64 lanes x 16 B
    const T* __restrict__ Qp = reinterpret_cast<const T*>(a.Qh) + ((size_t)bh * Sp + q0) * 64 + lane * 8;
    const T* __restrict__ Kp = reinterpret_cast<const T*>(a.Kh) + (size_t)bh * Sp * 64 + lane * 8;
    const T* __restrict__ Vp = reinterpret_cast<const T*>(a.Vt) + (size_t)bh * 64 * Sp + lane * 8;
    const T* __restrict__ PKp = reinterpret_cast<const T*>(a.PK) + (size_t)hh * a.P * 64 + 32 * 8 * h;
    const T* __restrict__ PQp = reinterpret_cast<const T*>(a.PQ) + (size_t)hh * a.P * 64 + 32 * 8 * h;
    const float* __restrict__ kb = a.kbias + (size_t)b * Sp;
    float* c2p_l = lds + (size_t)wave * 2 * 32 * LROW;
    float* p2c_l = c2p_l + 32 * LROW;

    int nkt = (a.klen[b] + 31) >> 5;                               // key tiles beyond the last valid key add exactly 0
    nkt = nkt < 1 ? 1 : (nkt > (Sp >> 5) ? (Sp >> 5) : nkt);
    const int kfirst = a.kfirst[b];                                // first masked key: tiles below it need no key bias
    const int foff = 8 * h;                                        // fragment k-offset of this lane half

    // delta of row c of relative-distance block L(t): rel = q0 - 32 t - 31 + c  (t = -1 is the block above tile 0)
    // Position-row addressing.  otab (host-built per Sp, engine.hip) maps a relative distance q - k, entry (q - k) + Sp - 1 + 64,
    // to the BYTE offsets of row delta(q - k) in the PQ (x) and PK (y) fragment layouts; the 64 entries of padding on each side
    // repeat the clamped ends, so a block of 32 consecutive distances needs one 8-byte load per lane and no clamp / shift / pi math.
    // Row c of relative-distance block L(t) is rel = q0 - 32 t - 31 + c  (t = -1 is the block above tile 0).
    const int2* __restrict__ otab = a.otab + (q0 - 31 + c + Sp - 1 + 64);
    auto block_delta = [&](int t) -> int2 { return otab[-32 * t]; };
    auto load_tile = [&](const T* base, int tile, frag_t (&f)[4]) {      // K tile / Q tile: 4 contiguous 1-KiB units
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const frag_t*>(base + (size_t)tile * 2048 + s * 512);
    };
    auto load_pq = [&](int2 d, frag_t (&f)[4]) {                          // row of query_proj(rel): Q layout
        const char* p = reinterpret_cast<const char*>(PQp) + d.x;
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const frag_t*>(p + s * UNITB);
    };
    auto load_pk = [&](int2 d, frag_t (&f)[4]) {                          // row of key_proj(rel): K layout (pi on the row)
        const char* p = reinterpret_cast<const char*>(PKp) + d.y;
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const frag_t*>(p + s * UNITB);
    };
    auto band_store = [&](float* dst, const f32x16& v) {           // 4 consecutive rr per register group
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(dst + 8 * g + 4 * h) = (f32x4){v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    };

    frag_t qf[4];
    load_tile(Qp, 0, qf);

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    const int rr_base = c - 8 * h + 31;
    SegClock<DIAG> clk;

    // Shared tail of every key tile: key bias, online softmax (log2 units, deferred rescale), P*V.
    auto softmax_pv = [&](float (&sv)[16], int kt) {
        const int k0 = kt * 32;
        frag_t vt[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            vt[0][t] = *reinterpret_cast<const frag_t*>(Vp + (size_t)kt * 2048 + t * 512);
            vt[1][t] = *reinterpret_cast<const frag_t*>(Vp + (size_t)kt * 2048 + 1024 + t * 512);
        }
        if (k0 + 32 > kfirst) {                                             // wave-uniform: tile holds masked keys
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(kb + k0 + foff);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(kb + k0 + foff + 4);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff);
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sv[i] += b0[i]; sv[4 + i] += b1[i]; sv[8 + i] += b2[i]; sv[12 + i] += b3[i]; }
        }
        float mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);          // v_max3 chain
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, sv[i]), sv[i + 1]);
        mx = fmaxf(mx, sv[15]);
        clk.mark(3);
        // Deferred rescale: scores are in log2 units; the reference exponent m only moves when some row's
        // maximum outgrows it by more than RESCALE_THR (P then stays <= 2^THR, exact in fp32 sums and
        // at unchanged relative precision in the 16-bit P operand).  Wave-uniform branch.  m is identical in
        // the two lane halves of a query (it only ever takes exchanged values), so the test needs no
        // cross-half exchange — that LDS round trip sits in the rare branch.
        if (__builtin_amdgcn_ballot_w64(mx - m > RESCALE_THR) != 0ull) {
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
        clk.mark(4);
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
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) pfr[j] = (T)sv[8 * t + j];
            }
            mma32(vt[0][t], pfr, o0);     // O^T[dd][query c], dd = (i&3) + 8*(i>>2) + 4h
            mma32(vt[1][t], pfr, o1);     //                   dd + 32
        }
        clk.mark(5);
    };

    // Saturated key tiles: every q-k of the tile lies beyond the clamp of the bucket table, so delta is ONE
    // value d*: c2p = Q_q.PK[d*] is a per-query constant and p2c = K_k.PQ[d*] is a second product on the
    // same K fragments.  8 + 4 MFMA, no band, no LDS.  (same operands, same fp32 accumulation as the band path)
    auto sat_tiles = [&](int kt_lo, int kt_hi, int dstar) {
        if (kt_lo >= kt_hi) return;
        frag_t pqb[4], pkb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // broadcast fragments: every row / column is table row d*
            pqb[s] = *reinterpret_cast<const frag_t*>(PQp + (size_t)(dstar >> 5) * 2048 + (dstar & 31) * 8 + s * 512);
            pkb[s] = *reinterpret_cast<const frag_t*>(PKp + (size_t)(dstar >> 5) * 2048 + glc_pi32(dstar & 31) * 8 + s * 512);
        }
        float cq;
        {
            f32x16 t;
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mma32(pkb[s], qf[s], t);            // every row = PK[d*] . Q_c
            cq = t[0];
        }
        frag_t kf[4];
        load_tile(Kp, kt_lo, kf);
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            frag_t n_kf[4];
            load_tile(Kp, kt + 1 < kt_hi ? kt + 1 : kt, n_kf);
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = cq;
#pragma unroll
            for (int s = 0; s < 4; ++s) mma32(kf[s], qf[s], sacc);
#pragma unroll
            for (int s = 0; s < 4; ++s) mma32(kf[s], pqb[s], sacc);        // + K_k . PQ[d*] (same for every query column)
            float sv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = sacc[i];
            softmax_pv(sv, kt);
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = n_kf[s];
        }
    };

    // key-tile ranges: [0, kt_a) saturated high (q-k >= rsat_pos), [kt_a, kt_b) banded, [kt_b, nkt) saturated low
    int kt_a = q0 - 31 - a.rsat_pos >= 0 ? (q0 - 31 - a.rsat_pos) / 32 + 1 : 0;
    kt_a = kt_a > nkt ? nkt : kt_a;
    int kt_b = (q0 + 31 - a.rsat_neg + 31) / 32;
    kt_b = kt_b < kt_a ? kt_a : (kt_b > nkt ? nkt : kt_b);
    // cooperative form: this wave's quarter [k_lo, k_hi) cuts the three ranges; an empty quarter leaves m = -3e38, l = 0, O = 0
    const int k_lo = kpart < 0 ? 0 : (nkt * kpart) >> 2, k_hi = kpart < 0 ? nkt : (nkt * (kpart + 1)) >> 2;
    const int s0_hi = kt_a < k_hi ? kt_a : k_hi, s1_lo = kt_b > k_lo ? kt_b : k_lo;
    kt_a = kt_a > k_lo ? kt_a : k_lo;
    kt_b = kt_b < k_hi ? kt_b : k_hi;

    sat_tiles(k_lo, s0_hi, a.P - 1);

    if (kt_a < kt_b) {
    // Fragment sets are addressed STATICALLY (runtime-indexed register arrays would go to scratch).  The default build
    // (UNROLL6, band_tile2 below) needs KF[0], PQ[0], PQ[2]; the rolled form (diagnostics, fp32) slides three PQ sets by copying.
    frag_t KF[2][4], PQ[3][4];
    // ---- band prologue: c2p blocks L(kt_a-1), L(kt_a) -> their ring halves; PQ fragments of both; K of tile kt_a ----
    {
        frag_t pk[4];
        f32x16 bacc;
        int2 d = block_delta(kt_a - 1);
        load_pk(d, pk);
        load_pq(d, PQ[2]);                   // step 0: low = PQ[0], high = PQ[2]
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) mma32(pk[s], qf[s], bacc);
        band_store(c2p_l + c * LROW + 32, bacc);            // ring half 1: the high block of the first band tile (step 0)
        d = block_delta(kt_a);
        load_pk(d, pk);
        load_pq(d, PQ[0]);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) mma32(pk[s], qf[s], bacc);
        band_store(c2p_l + c * LROW, bacc);                 // ring half 0: its low block
        load_tile(Kp, kt_a, KF[0]);
    }
    int2 d_next = block_delta(kt_a + 1 < kt_b ? kt_a + 1 : kt_a);

    // xr = 32 * (step parity): the ring half that holds this tile's low block.  It follows the STEP (a compile-time constant in
    // the unrolled loop), not the tile index, so on even steps the c2p gather addresses are lane base + immediate.
    auto band_tile = [&](int kt, const int xr, frag_t (&kf)[4], frag_t (&n_kf)[4], frag_t (&pq_lo)[4], frag_t (&pq_hi)[4], frag_t (&n_pq)[4]) {
        // ---- prefetch the next tile's operands (clamped re-load on the last tile) ----
        const int ktn = kt + 1 < kt_b ? kt + 1 : kt;
        clk.start();
        load_tile(Kp, ktn, n_kf);
        load_pq(d_next, n_pq);
        const int2 d_pk = d_next;
        d_next = block_delta(kt + 2 < kt_b ? kt + 2 : ktn);

        // ---- S^T = K Q^T ; reg i <-> key k0 + 16*(i>>3) + 8h + (i&7) ----
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) mma32(kf[s], qf[s], sacc);

        // ---- p2c band for both blocks of this key tile: [rr][key lane] ----
        {
            f32x16 bacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mma32(pq_lo[s], kf[s], bacc);
            band_store(p2c_l + c * LROW, bacc);          // row = lane (conflict-free); the gather applies pi
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mma32(pq_hi[s], kf[s], bacc);
            band_store(p2c_l + c * LROW + 32, bacc);
        }
        clk.mark(0);
        wave_lds_sync();
        clk.mark(1);

        float sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kc = 16 * (i >> 3) + (i & 7);                         // key offset minus 8h
            const int prow = 16 * (i >> 3) + 8 * ((i >> 2) & 1) + (i & 3);  // pi(key offset) minus 4h: p2c rows are in lane order
            const int rr = rr_base - kc;
            sv[i] = sacc[i] + c2p_l[c * LROW + (rr ^ xr)] + p2c_l[(prow + 4 * h) * LROW + rr];
        }
        clk.mark(2);
        frag_t pk[4];
        load_pk(d_pk, pk);                      // needed only after the softmax: its latency hides under it
        softmax_pv(sv, kt);

        // ---- c2p band of the NEXT tile's low block L(kt+1) -> ring half (kt+1)&1 (held L(kt-1), now dead) ----
        {
            f32x16 bacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mma32(pk[s], qf[s], bacc);          // [rr][query c]
            wave_lds_sync();              // this tile's gathers retire before the ring slot is overwritten
            band_store(c2p_l + c * LROW + (xr ^ 32), bacc);
        }
        clk.mark(6);
        clk.stop();
    };
    // Lean form of the same tile (the default build): ONE K set and TWO PQ sets, re-loaded IN PLACE as soon as their last
    // MFMA of this tile has issued (the next K tile into the K set; the next low block into the set that was this tile's high
    // block), and the c2p MFMAs of the next block issued BEFORE the softmax so the matrix pipe works under its VALU stream.
    // The roles of the two PQ sets swap every tile, so the loop is unrolled by two.  Same operands, same arithmetic.
    auto band_tile2 = [&](int kt, const int xr, frag_t (&kf)[4], frag_t (&pq_lo)[4], frag_t (&pq_hi)[4]) {
        const int ktn = kt + 1 < kt_b ? kt + 1 : kt;
        frag_t pk[4];
        load_pk(d_next, pk);                    // PK rows of L(kt+1): consumed right after the gather
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) mma32(kf[s], qf[s], sacc);
        {
            f32x16 bacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mma32(pq_lo[s], kf[s], bacc);
            band_store(p2c_l + c * LROW, bacc);
#pragma unroll
            for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) mma32(pq_hi[s], kf[s], bacc);
            band_store(p2c_l + c * LROW + 32, bacc);
        }
        load_tile(Kp, ktn, kf);                 // in place: K(kt+1)
        load_pq(d_next, pq_hi);                 // in place: L(kt+1), the next tile's low block
        d_next = block_delta(kt + 2 < kt_b ? kt + 2 : ktn);
        wave_lds_sync();
        float sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kc = 16 * (i >> 3) + (i & 7);
            const int prow = 16 * (i >> 3) + 8 * ((i >> 2) & 1) + (i & 3);
            const int rr = rr_base - kc;
            sv[i] = sacc[i] + c2p_l[c * LROW + (rr ^ xr)] + p2c_l[(prow + 4 * h) * LROW + rr];
        }
        f32x16 bacc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc2[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) mma32(pk[s], qf[s], bacc2);             // c2p of L(kt+1)  [rr][query c]
        softmax_pv(sv, kt);
        wave_lds_sync();                        // this tile's gathers retire before the ring slot is overwritten
        band_store(c2p_l + c * LROW + (xr ^ 32), bacc2);
    };
    if constexpr (UNROLL6) {
        // prologue left: KF[0] = K(kt_a), PQ[0] = L(kt_a) (low), PQ[2] = L(kt_a - 1) (high)
        for (int kt = kt_a;;) {
            band_tile2(kt, 0, KF[0], PQ[0], PQ[2]);
            if (++kt >= kt_b) break;
            band_tile2(kt, 32, KF[0], PQ[2], PQ[0]);
            if (++kt >= kt_b) break;
        }
    } else {                                   // rolled form: slide by copying the fragment sets
        for (int kt = kt_a; kt < kt_b; ++kt) {
            band_tile(kt, ((kt - kt_a) & 1) << 5, KF[0], KF[1], PQ[0], PQ[2], PQ[1]);
#pragma unroll
            for (int s = 0; s < 4; ++s) { PQ[2][s] = PQ[0][s]; PQ[0][s] = PQ[1][s]; KF[0][s] = KF[1][s]; }
        }
    }
    }

    sat_tiles(s1_lo, k_hi, 0);

    if constexpr (DIAG) {
        if (a.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {     // 64 blocks of XCD 0
            unsigned long long* o = a.stamps + ((size_t)(blockIdx.x >> 3) * 4 + wave) * 8;
            for (int k = 0; k < 7; ++k) o[k] = clk.seg[k];
            o[7] = clk.tiles;
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (kpart >= 0) {
        // merge the four key quarters: M = max m_j, L = sum l_j 2^(m_j - M), O = sum O_j 2^(m_j - M).  The band scratch is dead; every
        // wave parks (m, l, O) in its own 8.5 KiB of it, wave 0 folds the other three in and stores.
        float* park = lds + (size_t)wave * 2 * 32 * LROW;          // 34 floats per lane, lane-major (conflict-free)
        __syncthreads();                                            // (all four waves are here: same q0, same branches above)
        if (wave > 0) {
            park[lane] = m; park[64 + lane] = l;
#pragma unroll
            for (int i = 0; i < 16; ++i) { park[(2 + i) * 64 + lane] = o0[i]; park[(18 + i) * 64 + lane] = o1[i]; }
        }
        __syncthreads();
        if (wave > 0) return;
        float mj[3], M = m;
#pragma unroll
        for (int j = 0; j < 3; ++j) { mj[j] = lds[(size_t)(j + 1) * 2 * 32 * LROW + lane]; M = fmaxf(M, mj[j]); }
        const float f0 = __builtin_amdgcn_exp2f(m - M);
        l *= f0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= f0; o1[i] *= f0; }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float* pj = lds + (size_t)(j + 1) * 2 * 32 * LROW;
            const float fj = __builtin_amdgcn_exp2f(mj[j] - M);
            l += pj[64 + lane] * fj;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] += pj[(2 + i) * 64 + lane] * fj; o1[i] += pj[(18 + i) * 64 + lane] * fj; }
        }
    }
    const float inv = 1.0f / l;
    T* out = reinterpret_cast<T*>(a.CTX) + ((size_t)b * Sp + q0 + c) * a.H + hh * 64;
    if constexpr (sizeof(T) == 2) {
        store_acc32_wide<T>(o0, inv, out, h);          // 16-byte stores via v_permlane32_swap (glc_common.h)
        store_acc32_wide<T>(o1, inv, out + 32, h);
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            store4<T>(out + 8 * g + 4 * h, o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            store4<T>(out + 32 + 8 * g + 4 * h, o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

// Straightforward kernel (any T, no MFMA): one block per (query row, batch*head).  Serves the fp32 mode, the
// on-device cross-check of the band kernel, and — with a row selection — the pruned last layer, where only
// the 1+C rows per sequence that the head reads need attention output (their positions are arbitrary, so
// the 32-consecutive-query band kernel does not apply).  16-byte loads in the fragment-major layouts.
template <typename T>
__global__ __launch_bounds__(256) void attn_simple_kernel(AttnArgs a) {
    constexpr int VEC = 16 / (int)sizeof(T);          // elements per 16-byte chunk (8, or 4 for f32)
    typedef __attribute__((ext_vector_type(VEC))) T vecT;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qv = sm;            // [64]
    float* red = sm + 64;      // [256]
    float* sc = sm + 64 + 256; // [Sp]
    const int Sp = a.Sp, hh = blockIdx.y, t = threadIdx.x;
    int b, q, orow;
    if (a.sel_b) { b = a.sel_b[blockIdx.x]; q = a.sel_q[blockIdx.x]; orow = blockIdx.x; }
    else { b = blockIdx.z; q = blockIdx.x; orow = b * Sp + q; }
    const int bh = b * a.nh + hh;
    const T* Kg = reinterpret_cast<const T*>(a.Kh);
    const T* Vg = reinterpret_cast<const T*>(a.Vt);
    const T* PKg = reinterpret_cast<const T*>(a.PK);
    const T* PQg = reinterpret_cast<const T*>(a.PQ);
    const float* kb = a.kbias + (size_t)b * Sp;
    if (t < 64)
        qv[t] = a.Qrow ? (float)reinterpret_cast<const T*>(a.Qrow)[(size_t)orow * a.H + hh * 64 + t]
                       : (float)reinterpret_cast<const T*>(a.Qh)[glc_qoff(Sp, bh, q, t)];
    __syncthreads();
    int kend = (a.klen[b] + 7) & ~7;                  // keys past the last valid one contribute exactly 0
    kend = kend < 8 ? 8 : (kend > Sp ? Sp : kend);
    float mx = -3.0e38f;
    for (int k = t; k < kend; k += 256) {
        const int dl = a.dtab[q - k + Sp - 1];
        float s = 0.f;
#pragma unroll
        for (int e0 = 0; e0 < 64; e0 += VEC) {
            const vecT kv = *reinterpret_cast<const vecT*>(Kg + glc_koff(Sp, bh, k, e0));
            const vecT pk = *reinterpret_cast<const vecT*>(PKg + glc_koff(a.P, hh, dl, e0));
            const vecT pq = *reinterpret_cast<const vecT*>(PQg + glc_qoff(a.P, hh, dl, e0));
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float kf = (float)kv[e];
                s += qv[e0 + e] * (kf + (float)pk[e]) + kf * (float)pq[e];
            }
        }
        s += kb[k];
        sc[k] = s;
        mx = fmaxf(mx, s);
    }
    red[t] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmaxf(red[t], red[t + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int k = t; k < kend; k += 256) { const float p = __builtin_amdgcn_exp2f(sc[k] - mx); sc[k] = p; sum += p; }
    red[t] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const float inv = 1.0f / red[0];
    __syncthreads();
    const int dd = t & 63, part = t >> 6;
    float acc = 0.f;
    for (int k0 = part * VEC; k0 < kend; k0 += 4 * VEC) {            // 16-byte units of V^T: VEC consecutive keys
        const vecT vv = *reinterpret_cast<const vecT*>(Vg + glc_voff(Sp, bh, dd, k0));
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += sc[k0 + e] * (float)vv[e];
    }
    red[t] = acc;
    __syncthreads();
    if (t < 64) {
        const float v = (red[t] + red[t + 64] + red[t + 128] + red[t + 192]) * inv;
        reinterpret_cast<T*>(a.CTX)[(size_t)orow * a.H + hh * 64 + t] = (T)v;
    }
}

}  // namespace

// Shape contract: head_dim == 64, Sp % 64 == 0, P % 32 == 0, H == nh*64, dtab has 2*Sp-1 entries in [0, P);
// Q/K/V^T/PQ/PK in the fragment-major layouts of glc_layout.h.
const char* glc_launch_attention(hipStream_t st, int dtype, int impl, const AttnArgs& a) {
    if ((!a.Qh && !a.Qrow) || !a.Kh || !a.Vt || !a.PK || !a.PQ || !a.dtab || !a.kbias || !a.klen || !a.kfirst || !a.CTX) return "attention: null pointer";
    if (impl == 2 && !a.otab) return "attention: the band kernel needs the offset table";
    if (a.B <= 0 || a.nh <= 0 || a.Sp <= 0 || a.Sp % 64 || a.H != a.nh * 64 || a.P <= 0 || a.P % 32) return "attention: bad shape";
    if (impl == 2 && (a.sel_b || !a.Qh)) return "attention: the band kernel takes no row selection";
    if (a.split && (impl != 2 || dtype != GLC_DT_F32)) return "attention: split operands exist for the fp32 band kernel only";
    if (impl == 2) {
        const int nqb = (a.Sp + 127) / 128, bh8 = (a.B * a.nh + 7) / 8 * 8;
        dim3 grid(nqb * bh8), block(256);
        const size_t dyn = (a.variant & 2) ? 60 * 1024 : 0;     // diagnostic: pad LDS so that one block fits per CU (one wave per SIMD)
        static const bool unroll6 = glc_dev_env("GLC_ATTN_ROLLED") == nullptr;       // default: lean two-step band loop; the env picks the rolled one (A/B)
        if (dtype == GLC_DT_F32) {
            if (a.stamps) return "attention: the stamped build exists for f16 only";
            // The lean (in-place reload) loop with split fragments: round 1 measured it 5 % faster but WRONG (err 5e-2).  Root cause (round 2,
            // docs/LOG_r01-r05.md section 2): a wait count that hipcc leaves out in THIS instantiation once its SLP vectoriser has formed packed-fp32
            // (v_pk_*_f32) instructions — the same object is correct with -mllvm -amdgpu-waitcnt-forcezero, with packed fp32 disabled, or
            // with -fno-slp-vectorize, which is how this translation unit is built (Makefile); in that build the instantiation passes the
            // whole fp32 parity suite (GLC_ATTN_WG=0 GLC_ATTN_SPLIT_LEAN=1).  It stays a diagnostic: the split mode's attention is
            // attention_wg.hip, and this kernel serves it only for the pruned last layer, in the rolled form.
            static const bool lean32 = glc_dev_env("GLC_ATTN_F32_LEAN") != nullptr;      // diagnostic only: plain fp32 fragments, lean loop
            static const bool split_lean = glc_dev_env("GLC_ATTN_SPLIT_LEAN") != nullptr;     // diagnostic only: split fragments, lean loop
            if (a.split && split_lean) hipLaunchKernelGGL((attn_band_kernel<float, true, false, true>), grid, block, dyn, st, a);
            else if (a.split) hipLaunchKernelGGL((attn_band_kernel<float, false, false, true>), grid, block, dyn, st, a);
            else if (lean32) hipLaunchKernelGGL((attn_band_kernel<float, true>), grid, block, dyn, st, a);
            else hipLaunchKernelGGL((attn_band_kernel<float, false>), grid, block, dyn, st, a);
        } else if (a.stamps) {
#ifdef GLC_DEVELOPER
            if (dtype != GLC_DT_F16) return "attention: the stamped build exists for f16 only";
            hipLaunchKernelGGL((attn_band_kernel<f16_t, false, true>), grid, block, dyn, st, a);   // rolled loop: room for the stamp registers
#else
            return "attention: the stamped build exists in developer builds only (make DEV=1)";
#endif
        } else if (dtype == GLC_DT_BF16) {
            if (unroll6) hipLaunchKernelGGL((attn_band_kernel<bf16_t, true>), grid, block, dyn, st, a);
            else hipLaunchKernelGGL((attn_band_kernel<bf16_t, false>), grid, block, dyn, st, a);
        } else {
            if (unroll6) hipLaunchKernelGGL((attn_band_kernel<f16_t, true>), grid, block, dyn, st, a);
            else hipLaunchKernelGGL((attn_band_kernel<f16_t, false>), grid, block, dyn, st, a);
        }
        return nullptr;
    }
    const size_t shm = (size_t)(64 + 256 + a.Sp) * sizeof(float);
    if (shm > 64 * 1024) return "attention(simple): sequence too long for the reference kernel";
    if ((a.sel_b != nullptr) != (a.sel_q != nullptr) || (a.sel_b && (!a.Qrow || a.nsel <= 0))) return "attention(simple): bad row selection";
    dim3 grid(a.sel_b ? a.nsel : a.Sp, a.nh, a.sel_b ? 1 : a.B), block(256);
    switch (dtype) {
        case GLC_DT_F32: hipLaunchKernelGGL(attn_simple_kernel<float>, grid, block, shm, st, a); break;
        case GLC_DT_BF16: hipLaunchKernelGGL(attn_simple_kernel<bf16_t>, grid, block, shm, st, a); break;
        case GLC_DT_F16: hipLaunchKernelGGL(attn_simple_kernel<f16_t>, grid, block, shm, st, a); break;
        default: return "attention: bad dtype";
    }
    return nullptr;
}
