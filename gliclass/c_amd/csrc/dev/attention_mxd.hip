// DeBERTa-v2/v3 disentangled self-attention on MX tiles, TWO query tiles per wave, ONE wave per SIMD, software-pipelined over the key tiles
// (round 5, developer build: correct, 1.27 ms per launch at c3 where the band kernel of attention_mx.hip takes 1.09 — not the default).
//
// The band kernel (attention_mx.hip: read its header first — algebra, operand formats, the c2p rings, the shared p2c image, saturated tiles)
// runs two independent waves per SIMD with 256 registers each; every wave reads the whole K / V^T tile and requests a whole PK block per key
// tile for its ONE query tile: ≈37 KB of LDS traffic and 10 KB of row requests per (query tile, key tile) — the LDS array and the matrix pipe
// are equally loaded (docs/LOG_r01-r05.md §3g, "the budgets").  Here a workgroup is four waves, one per SIMD with the SIMD's whole register file (512),
// and wave w owns the query tiles A = 2 w and B = 2 w + 1 of 256 consecutive queries:
//   * K fragments are read from LDS once per key tile for both query tiles, the key bias likewise (V^T: per 32-row half and query tile — both
//     halves resident cost 16 registers more than the fused step has);
//   * the PK block that is NEW for tile B at key tile t is the block that was new for tile A at t - 1: one PK request per wave and key tile;
//     tile B's c2p block is computed from it together with tile A's and waits a step in 16 registers (`cbn`);
//   * the p2c image has nine slots; a wave keeps TWO PQ blocks resident (slots (w + t) & 7 and (w + t + 4) & 7; the block that leaves is replaced
//     in place every fourth key tile by an ordinary conditional load — at 512 registers hipcc coalesces it, no inline asm), the ninth block is
//     computed by the wave (2 - t) & 3 from rows the DMA staged in LDS two steps ahead (no registers held while they travel);
//   * a step is S(t) fused with M(t + 1): the back of S(t) (exponentials, splits, P.V, c2p blocks: ≈2000 cycles of VALU with 1024 of MFMA) and
//     M(t + 1) (K fragments, p2c blocks, S^T of both tiles: 1088 cycles of MFMA, no VALU) sit in branch-free segments between scheduling
//     fences; everything conditional (block replacement, staging, the ninth block) sits between the segments;
//   * two workgroup barriers per step for eight query tiles (the band kernel: three per tile for four): X' (every wave has gathered image(t))
//     and Y' (image(t + 1) complete, K(t + 2) / V^T(t + 1) landed).  K(t + 2) is requested at the step's start, the PK rows in the first
//     segment, V^T(t + 1) and the staging in the second: 14 KB per wave in one burst blocked the issue for 700 cycles.
// Built with -mllvm -amdgpu-mfma-vgpr-form (Makefile): by default every MFMA result of a kernel that may use accumulation registers is
// allocated there, the accumulators alone fill the 256 AGPRs and the operands spill (37–190 spilled registers in every form tried); with the
// flag the allocator treats the 512 registers as one pool: 0 spills at 256 + 190.  (A spilling variant under the flag crashes this LLVM in
// AMDGPURewriteAGPRCopyMFMA — keep the pressure below the limit.)
// What the measurements say (profiles/r05/attn_mxd_log.txt): untracked inline-asm loads are NOT usable here — under register pressure the
// allocator copies their destination registers before the wait (stale rows; two builds wrong for that reason); the scheduler does interleave
// the two streams, but a fused segment takes MFMA + VALU, not max(MFMA, VALU) (512 + 850 -> 1385 ticks) — whether hipcc places the instructions
// (with or without sched_group_barrier pipelines) or the source does (the band -> band step below: one softmax slice between two MFMAs of different
// accumulator chains, fenced; -disable-machine-sink keeps the slices where they are): the time moves between the segments, their sum stays at
// ≈4500 of the step's ≈6100 cycles.  A clean loop does overlap (scripts/probes/mfma_valu_overlap_probe.hip: 8 MFMAs + 64 v_exp in one wave's stream take
// 604 ticks where they take 256 and 560 alone; only v_pk_add_f32 refuses: 672 against 256 / 344), so the cause is in this kernel's streams, not in the
// SIMD — 16-register accumulators and 8-register operands in the same register file as the softmax's operands are the suspect; not found this round.
// Products and their order are those of the band kernel except: S^T starts from zero and the c2p band is added with the p2c band (one rounding
// apart), the saturated tiles' K.PQ[d*] is computed once for both query tiles — results agree to one unit of the GX output format.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "../glc_common.h"
#include "../glc_kernels.h"
#include "../glc_layout.h"
#include "../glc_pfrag.h"

namespace {

constexpr float RESCALE_THR = 8.0f;
constexpr int LROW = 68;                    // floats per c2p ring row (2 blocks of 32 + 4 pad)
constexpr int TILEB = GLC_MXT_BYTES;
constexpr int NQT = 8;                      // query tiles per workgroup
constexpr int LROWP = 32 * (NQT + 1) + 4;   // floats per p2c image row (nine slots)
constexpr size_t OFF_IMG = (size_t)NQT * 32 * LROW * 4;
constexpr size_t OFF_K = OFF_IMG + (size_t)32 * LROWP * 4;
constexpr size_t OFF_V = OFF_K + 2 * TILEB;
constexpr size_t OFF_PX = OFF_V + 2 * TILEB;     // rows of the ninth p2c block, staged a tile ahead by the wave that will use them (two buffers)
constexpr size_t LDS_BYTES = OFF_PX + 2 * TILEB;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct MxFrag { f16x8 f[4]; i32x8 x[2]; };

__device__ __forceinline__ void glds16_sv(const unsigned char* ubase, unsigned lane_off, void* l) {
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
    // (the builtin, not inline asm: the compiler's own wait insertion then knows that nothing is outstanding behind this point — with the asm form it
    //  keeps counted waits for last tile's table offsets in the next tile, and a counted wait behind fresh DMA requests stalls for their latency)
    __builtin_amdgcn_s_waitcnt(0x0070);         // vmcnt(0) lgkmcnt(0)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ i32x8 cat8(const i32x4& a, const i32x4& b) {
    i32x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem_mxd[];

struct QState { MxFrag qf; f32x16 o0, o1; float m, l; };

template <bool DIAG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_mxd_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int Sp = a.Sp;
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
    auto mm_p_lh = [&](const PFrag& hl, const MxFrag& lh, f32x16& acc) __attribute__((always_inline)) {      // a resident PQ block (hi8 | lo8) x K
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl.f[s], lh.f[s], acc, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat8(hl.xa[m], hl.xb[m]), lh.x[m], acc, 0, 0, 1, SC, 0, SC);
    };

    float* ring_a = reinterpret_cast<float*>(smem_mxd) + (size_t)(2 * wave) * 32 * LROW;      // this wave's two c2p rings [32 q][64 + 4]
    float* ring_b = ring_a + 32 * LROW;
    float* img = reinterpret_cast<float*>(smem_mxd + OFF_IMG);                                  // shared [32 keys][LROWP]
    unsigned char* k_lds = smem_mxd + OFF_K;
    unsigned char* v_lds = smem_mxd + OFF_V;

    const int nqb = (Sp + 32 * NQT - 1) / (32 * NQT);
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int bh = xcd + 8 * (jj / nqb);
    const int Q0 = (jj % nqb) * 32 * NQT;
    const int QX = Q0 + 32 * NQT;
    if (bh >= a.B * a.nh) return;
    const int b = bh / a.nh, hh = bh - b * a.nh;
    const int q0 = Q0 + 64 * wave;                      // tile A: q0 .. q0 + 31, tile B: q0 + 32 .. q0 + 63 (Sp is a multiple of 64)
    const bool active = q0 < Sp;
    const int q0m = active ? q0 : Sp - 64;
    const int klen = a.klen[b];
    if (Q0 >= klen && Q0 > 0) {
        if (active) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                unsigned char* row = reinterpret_cast<unsigned char*>(a.CTX) + ((size_t)b * Sp + q0 + 32 * t + c) * 4 * a.H + (size_t)(2 * hh) * 128 + h * 128;
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(row + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
            }
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

    const int otab_max = 2 * Sp - 2 + 128;
    auto block_x = [&](int qb, int t) -> int {
        int idx = qb - 32 * t - 31 + c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return reinterpret_cast<const int*>(a.otab)[2 * idx];
    };
    auto block_y = [&](int qb, int t) -> int {
        int idx = qb - 32 * t - 31 + c + Sp - 1 + 64;
        idx = idx < 0 ? 0 : (idx > otab_max ? otab_max : idx);
        return reinterpret_cast<const int*>(a.otab)[2 * idx + 1];
    };
    auto rows_vf = [&](int off) -> unsigned { return (unsigned)((off & ~8191) + ((off & 8191) >> 1) + h * 512); };
    auto rows_vx = [&](int off) -> unsigned { return (unsigned)(off + 4096 + h * 1024); };
    auto load_rows = [&](const unsigned char* base, int off, MxFrag& f) __attribute__((always_inline)) {
        const unsigned vf = rows_vf(off), vx = rows_vx(off);
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(base + (size_t)vf + s * 1024);
#pragma unroll
        for (int m = 0; m < 2; ++m)
            f.x[m] = cat8(*reinterpret_cast<const i32x4*>(base + (size_t)vx + m * 2048), *reinterpret_cast<const i32x4*>(base + (size_t)vx + (m * 2048 + 16)));
    };
    auto load_prows = [&](const unsigned char* base, int off, PFrag& f) __attribute__((always_inline)) {
        const unsigned vf = rows_vf(off), vx = rows_vx(off);
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(base + (size_t)vf + s * 1024);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f.xa[m] = *reinterpret_cast<const glc_i32x4*>(base + (size_t)vx + m * 2048);
            f.xb[m] = *reinterpret_cast<const glc_i32x4*>(base + (size_t)vx + (m * 2048 + 16));
        }
    };
    auto lds_tile = [&](const unsigned char* tile, MxFrag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(tile + s * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) f.x[m] = cat8(*reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + lane * 16), *reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + 1024 + lane * 16));
    };
    auto k_tile = [&](int t, MxFrag& f) __attribute__((always_inline)) {
        const unsigned char* tile = k_lds + (size_t)(t & 1) * TILEB;
#pragma unroll
        for (int s = 0; s < 4; ++s) f.f[s] = *reinterpret_cast<const f16x8*>(tile + s * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) f.x[m] = cat8(*reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + lane * 16), *reinterpret_cast<const i32x4*>(tile + 4096 + m * 2048 + 1024 + lane * 16));
    };
    auto band_store = [&](float* dst, const f32x16& v) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(dst + 8 * g + 4 * h) = (f32x4){v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    };
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
    // key tile t: K and V^T into their slots t & 1 (requested behind the barrier that ends tile t - 1 for everyone); wave w moves the piece pairs of attention_mx.hip's 4-wave form
    auto dma_k = [&](int t) __attribute__((always_inline)) { dma_pair(Kg + (size_t)t * TILEB + piece_src, k_lds + (size_t)(t & 1) * TILEB + piece_src, wave >= 2); };
    auto dma_v = [&](int t) __attribute__((always_inline)) { dma_pair(Vg + (size_t)t * TILEB + piece_src, v_lds + (size_t)(t & 1) * TILEB + piece_src, (wave & 1) != 0); };

    // the rows of a PQ block (table offset `off`, load_rows' addressing) gathered by the DMA into an LDS image of lds_tile's layout
    unsigned char* px_lds = smem_mxd + OFF_PX;
    auto stage_rows = [&](int off, unsigned char* buf) __attribute__((always_inline)) {
        const unsigned vf = rows_vf(off), vx = rows_vx(off);
#pragma unroll
        for (int s = 0; s < 4; ++s) glds16_sv(PQg + s * 1024, vf, buf + s * 1024);
#pragma unroll
        for (int m = 0; m < 2; ++m) { glds16_sv(PQg + m * 2048, vx, buf + 4096 + m * 2048); glds16_sv(PQg + m * 2048 + 16, vx, buf + 4096 + m * 2048 + 1024); }
    };
    QState A, B;
    auto load_q = [&](const unsigned char* g, MxFrag& qf) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) qf.f[s] = *reinterpret_cast<const f16x8*>(g + s * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) qf.x[m] = cat8(*reinterpret_cast<const i32x4*>(g + 4096 + m * 2048 + lane * 32), *reinterpret_cast<const i32x4*>(g + 4096 + m * 2048 + lane * 32 + 16));
    };
    load_q(Qg, A.qf);
    load_q(Qg + TILEB, B.qf);
    dma_k(0); dma_v(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) { A.o0[i] = 0.f; A.o1[i] = 0.f; B.o0[i] = 0.f; B.o1[i] = 0.f; }
    A.m = -3.0e38f; A.l = 0.f; B.m = -3.0e38f; B.l = 0.f;
    const int rr_base = c - 8 * h + 31;
    float one_f = 1.0f;
    asm volatile("" : "+s"(one_f));
    MxFrag kf;

    // V^T fragments of key tile kt (both 32-row halves) and the key bias: shared by the two query tiles
    // V^T fragments of key tile kt, one 32-row half at a time (both halves of both query tiles resident: 32 registers the fused step does not have)
    struct VHalf { f16x8 f[2]; i32x8 x; };
    auto load_vh = [&](int kt, int d, VHalf& v) __attribute__((always_inline)) {
        const unsigned char* vtile = v_lds + (size_t)(kt & 1) * TILEB + d * 4096;
        v.f[0] = *reinterpret_cast<const f16x8*>(vtile + lane * 16);
        v.f[1] = *reinterpret_cast<const f16x8*>(vtile + 1024 + lane * 16);
        v.x = cat8(*reinterpret_cast<const i32x4*>(vtile + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(vtile + 3072 + lane * 16));
    };
    // online softmax of one query tile's scores (log2 units, deferred rescale) and the split of P (attention_mx.hip softmax_pv)
    auto row_max = [&](const float (&sv)[16]) __attribute__((always_inline)) -> float {
        float mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, sv[i]), sv[i + 1]);
        return fmaxf(mx, sv[15]);
    };
    auto rescale = [&](float mx, QState& q) __attribute__((always_inline)) {      // the deferred rescale (rare)
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(q.m, mx);
        const float alpha = __builtin_amdgcn_exp2f(q.m - mnew);
        q.m = mnew;
        q.l *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { q.o0[i] *= alpha; q.o1[i] *= alpha; }
    };
    auto softmax_split = [&](float (&sv)[16], QState& q, f16x8 (&pf)[2], i32x8& px) __attribute__((always_inline)) {
        const f32x2 m2 = {q.m, q.m};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 d = (f32x2){sv[i], sv[i + 1]} - m2;
            sv[i] = __builtin_amdgcn_exp2f(d[0]); sv[i + 1] = __builtin_amdgcn_exp2f(d[1]);
            ps2 += (f32x2){sv[i], sv[i + 1]};
        }
        q.l += ps2[0] + ps2[1];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[t][j] = (f16_t)sv[8 * t + j];
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            int wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * qq], sv[4 * qq + 1], 0, false);
            wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * qq + 2], sv[4 * qq + 3], wh, true);
            px[qq] = wh;
            float r[4];
            const i32x4 pfw = __builtin_bit_cast(i32x4, pf[qq >> 1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pw = pfw[2 * (qq & 1) + (e >> 1)];
                if (e & 1) asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r[e]) : "v"(sv[4 * qq + e]), "s"(one_f), "v"(pw));
                else asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r[e]) : "v"(sv[4 * qq + e]), "s"(one_f), "v"(pw));
            }
            typedef short v2i16 __attribute__((ext_vector_type(2)));
            v2i16 wl2 = {0, 0};
            wl2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl2, r[0], r[1], 1.0f / (float)(1 << GLC_GX_SHIFT), false);
            wl2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl2, r[2], r[3], 1.0f / (float)(1 << GLC_GX_SHIFT), true);
            px[4 + qq] = __builtin_bit_cast(int, wl2);
        }
    };
    // the same softmax + split in twelve slices (quad k / 3 of the 16 scores, part k % 3): the fused band step issues one slice between two MFMAs
    auto sm_step = [&](const int k, float (&sv)[16], QState& q, f16x8 (&pf)[2], i32x8& px, f32x2& ps2) __attribute__((always_inline)) {
        const int qq = k / 3, part = k % 3;
        if (part == 0) {
            // (scalar adds: v_pk_add_f32 does not overlap with MFMAs at all — scripts/probes/mfma_valu_overlap_probe.hip: 672 ticks for 8 MFMAs + 64 packed adds
            //  where the two take 256 and 344 alone; v_add / v_fma / v_exp / the fp8 conversions hide under the matrix pipe)
#pragma unroll
            for (int i = 4 * qq; i < 4 * qq + 4; ++i) {
                float d = sv[i] - q.m;
                asm volatile("" : "+v"(d));
                sv[i] = __builtin_amdgcn_exp2f(d);
                ps2[i & 1] += sv[i];
            }
        } else if (part == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pf[qq >> 1][4 * (qq & 1) + e] = (f16_t)sv[4 * qq + e];
            int wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * qq], sv[4 * qq + 1], 0, false);
            wh = __builtin_amdgcn_cvt_pk_fp8_f32(sv[4 * qq + 2], sv[4 * qq + 3], wh, true);
            px[qq] = wh;
        } else {
            float r[4];
            const i32x4 pfw = __builtin_bit_cast(i32x4, pf[qq >> 1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pw = pfw[2 * (qq & 1) + (e >> 1)];
                if (e & 1) asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r[e]) : "v"(sv[4 * qq + e]), "s"(one_f), "v"(pw));
                else asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r[e]) : "v"(sv[4 * qq + e]), "s"(one_f), "v"(pw));
            }
            typedef short v2i16 __attribute__((ext_vector_type(2)));
            v2i16 wl2 = {0, 0};
            wl2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl2, r[0], r[1], 1.0f / (float)(1 << GLC_GX_SHIFT), false);
            wl2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wl2, r[2], r[3], 1.0f / (float)(1 << GLC_GX_SHIFT), true);
            px[4 + qq] = __builtin_bit_cast(int, wl2);
            if (k == 11) q.l += ps2[0] + ps2[1];
        }
    };
    // single MFMAs of the four product kinds (s: f16 unit 0 .. 3, m: MX step 0 / 1)
    auto mf = [&](f32x16& acc, const f16x8& x, const f16x8& y) __attribute__((always_inline)) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0); };
    auto ms_lh = [&](f32x16& acc, const i32x8& lh, const i32x8& hl) __attribute__((always_inline)) { acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(lh, hl, acc, 0, 0, 0, SC, 1, SC); };
    auto ms_hl = [&](f32x16& acc, const i32x8& hl, const i32x8& lh) __attribute__((always_inline)) { acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(hl, lh, acc, 0, 0, 1, SC, 0, SC); };
    auto pv = [&](int kt, QState& q, const f16x8 (&pf)[2], const i32x8& px) __attribute__((always_inline)) {
        VHalf v;
        load_vh(kt, 0, v);
#pragma unroll
        for (int t = 0; t < 2; ++t) q.o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v.f[t], pf[t], q.o0, 0, 0, 0);
        q.o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v.x, px, q.o0, 0, 0, 0, SC, 1, SC);
        load_vh(kt, 1, v);
#pragma unroll
        for (int t = 0; t < 2; ++t) q.o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v.f[t], pf[t], q.o1, 0, 0, 0);
        q.o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v.x, px, q.o1, 0, 0, 0, SC, 1, SC);
    };
    // ===================== the software pipeline over key tiles =====================
    // Step kt = S(kt) fused with M(kt + 1):
    //   M(t): K(t) fragments, S^T(t) = K Q^T of both query tiles from zero (saturated tiles: from the per-query c2p constant, and K.PQ[d*] once),
    //         band tiles: the p2c blocks of this wave into the image (the image must be free: barrier X' of the step).
    //   S(t): [front] scores = S^T + c2p band (ring gather) + p2c band (image gather), V^T fragments, key bias, row maxima, the rare rescale;
    //         [back]  exponentials, splits, P.V of both query tiles; band tiles: the c2p blocks of tile t + 1 into the rings.
    // The back of S(kt) — ≈2000 cycles of VALU with 1024 of MFMA — and M(kt + 1) — 1088 cycles of MFMA, almost no VALU — sit in ONE basic block
    // behind the rescale branch: two independent instruction streams for the scheduler.  Barrier Y' ends the step (image(kt + 1) complete, K(kt + 2)
    // and V^T(kt + 1) — requested at the step's start — landed).
    int kt_a = Q0 - 31 - a.rsat_pos >= 0 ? (Q0 - 31 - a.rsat_pos) / 32 + 1 : 0;
    kt_a = kt_a > nkt ? nkt : kt_a;
    int kt_b = (Q0 + 32 * (NQT - 1) + 31 - a.rsat_neg + 31) / 32;
    kt_b = kt_b < kt_a ? kt_a : (kt_b > nkt ? nkt : kt_b);

    f32x16 sa, sb, sk;                  // M's results: raw S^T of the two query tiles; saturated tiles: K.PQ[d*] (shared)
    MxFrag pqb;                         // saturated tiles: every row = PQ[d*]
    float cqa = 0.f, cqb = 0.f;         // ... and the per-query c2p constants Q_q.PK[d*]
    MxFrag pkN;                         // band: the PK block that is new for tile A at the next key tile — and for tile B one tile later:
    f32x16 cbn;                         // ... tile B's c2p block from it is computed with tile A's and kept in registers for a step (16 instead of the rows' 32)
    PFrag P0, P1;                       // band: resident PQ blocks, image slots (wave + t) & 7 and (wave + t + 4) & 7 at key tile t
    int ody_n = 0, odx_n = 0, eq_n = 0; // table offsets fetched a step ahead: PK rows of L_A(kt + 2), PQ rows of the ninth block of tile kt + 3, of the block that enters at tile kt + 3
    auto extra_wave = [&](int t) -> int { return (2 - t) & 3; };

    auto setup_sat = [&](int dstar) __attribute__((always_inline)) {
        MxFrag pkb;
        load_rows(PQg, (dstar >> 5) * 8192 + (dstar & 31) * 32, pqb);
        load_rows(PKg, (dstar >> 5) * 8192 + glc_pi32(dstar & 31) * 32, pkb);
        f32x16 t;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = 0.f;
        mm_lh_hl(pkb, A.qf, t);
        cqa = t[0];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = 0.f;
        mm_lh_hl(pkb, B.qf, t);
        cqb = t[0];
    };
    auto setup_band = [&]() __attribute__((always_inline)) {      // rings L(kt_a - 1), L(kt_a) of both query tiles (L_B(t) = L_A(t - 1)), resident blocks, the ninth block's rows
        f32x16 bacc;
        load_rows(PKg, block_y(q0, kt_a - 2), pkN);              // L_A(kt_a - 2) = L_B(kt_a - 1)
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
        mm_lh_hl(pkN, B.qf, bacc);
        band_store(ring_b + c * LROW + 32, bacc);
        load_rows(PKg, block_y(q0, kt_a - 1), pkN);              // L_A(kt_a - 1) = L_B(kt_a)
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
        mm_lh_hl(pkN, A.qf, bacc);
        band_store(ring_a + c * LROW + 32, bacc);
#pragma unroll
        for (int i = 0; i < 16; ++i) bacc[i] = 0.f;
        mm_lh_hl(pkN, B.qf, bacc);
        band_store(ring_b + c * LROW, bacc);
        load_rows(PKg, block_y(q0, kt_a), pkN);                  // L_A(kt_a) = L_B(kt_a + 1)
#pragma unroll
        for (int i = 0; i < 16; ++i) { bacc[i] = 0.f; cbn[i] = 0.f; }
        mm_lh_hl(pkN, A.qf, bacc);
        band_store(ring_a + c * LROW, bacc);
        mm_lh_hl(pkN, B.qf, cbn);                                // tile B's new block of the first band step: stored by that step
        load_prows(PQg, block_x(Q0 + 32 * ((wave + kt_a) & 7), kt_a), P0);
        load_prows(PQg, block_x(Q0 + 32 * ((wave + kt_a + 4) & 7), kt_a), P1);
        if (extra_wave(kt_a) == wave) stage_rows(block_x(QX, kt_a), px_lds + (size_t)(kt_a & 1) * TILEB);
        if (kt_a + 1 < kt_b && extra_wave(kt_a + 1) == wave) stage_rows(block_x(QX, kt_a + 1), px_lds + (size_t)((kt_a + 1) & 1) * TILEB);
        ody_n = block_y(q0, kt_a + 1);
        odx_n = block_x(QX, kt_a + 2);
        eq_n = block_x(Q0, kt_a + 1);
        __builtin_amdgcn_s_waitcnt(0x0070);
        wave_lds_sync();
    };

    // ---- M(t) ----
    auto M_sat = [&](int t) __attribute__((always_inline)) {
        k_tile(t, kf);
#pragma unroll
        for (int i = 0; i < 16; ++i) { sk[i] = 0.f; sa[i] = cqa; sb[i] = cqb; }
        mm_lh_hl(kf, pqb, sk);
        mm_lh_hl(kf, A.qf, sa);
        mm_lh_hl(kf, B.qf, sb);
    };
    // M(t) of a band tile in branch-free pieces (the step places them inside its segments; everything conditional sits between segments)
    auto M_band_a = [&](int t) __attribute__((always_inline)) {      // K fragments, the wave's first resident block, S^T of tile A
        k_tile(t, kf);
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa[i] = 0.f; sb[i] = 0.f; }
        f32x16 b0;
#pragma unroll
        for (int i = 0; i < 16; ++i) b0[i] = 0.f;
        mm_p_lh(P0, kf, b0);
        mm_lh_hl(kf, A.qf, sa);
        band_store(img + c * LROWP + 32 * ((wave + t) & 7), b0);
    };
    auto M_band_b = [&](int t) __attribute__((always_inline)) {      // second resident block, S^T of tile B
        f32x16 b1;
#pragma unroll
        for (int i = 0; i < 16; ++i) b1[i] = 0.f;
        mm_p_lh(P1, kf, b1);
        mm_lh_hl(kf, B.qf, sb);
        band_store(img + c * LROWP + 32 * ((wave + t + 4) & 7), b1);
    };
    auto M_band_r0 = [&](int t, int eq) __attribute__((always_inline)) {      // behind the first block's product: the block that leaves — the entering one takes its registers
        if (((wave + t) & 7) == 7 && t + 1 < kt_b) load_prows(PQg, eq, P0);
    };
    auto M_band_r1 = [&](int t, int eq) __attribute__((always_inline)) {
        if (((wave + t + 4) & 7) == 7 && t + 1 < kt_b) load_prows(PQg, eq, P1);
    };
    auto M_band_x = [&](int t) __attribute__((always_inline)) {      // the ninth block, on one wave
        if (extra_wave(t) == wave) {
            MxFrag pqx;
            lds_tile(px_lds + (size_t)(t & 1) * TILEB, pqx);
            f32x16 b2;
#pragma unroll
            for (int i = 0; i < 16; ++i) b2[i] = 0.f;
            mm_hl_lh(pqx, kf, b2);
            band_store(img + c * LROWP + 32 * NQT, b2);
        }
    };
    auto M_band = [&](int t, int eq) __attribute__((always_inline)) { M_band_a(t); M_band_r0(t, eq); M_band_b(t); M_band_r1(t, eq); M_band_x(t); };

    // ---- one step: S(kt) of type TS fused with M(kt + 1) of type TM ----
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tiles = 0, tlast = 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {      // DIAG: time since the previous stamp goes to segment k (k < 0: start)
        if constexpr (DIAG) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) seg[k] += t - tlast;
            tlast = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr int T_SAT = 0, T_BAND = 1, T_NONE = 3;      // (2: the band's last tile — no c2p block for a next one)
    auto step = [&](auto TSc, auto TMc, const int xr, const int kt) __attribute__((always_inline)) {
        constexpr int TS = decltype(TSc)::value, TM = decltype(TMc)::value;
        constexpr bool ST = TS == T_BAND && TM == T_BAND;       // DIAG stamps: the band's steady state
        if constexpr (ST) stamp(-1);
        // requests of the step: K(kt + 2), V^T(kt + 1); band: the PK rows of L_A(kt + 1) (for this step's c2p), the ninth block's rows of tile kt + 2
        if (kt + 2 < nkt) dma_k(kt + 2);
        int eq = 0;
        if constexpr (TM == T_BAND) { eq = eq_n; eq_n = block_x(Q0, kt + 3); }
        int ody = 0, odx = 0;
        if constexpr (TS == T_BAND) {
            ody = ody_n; odx = odx_n;
            ody_n = block_y(q0, kt + 2);
            odx_n = block_x(QX, kt + 3);
        }
        if constexpr (ST) stamp(0);                              // seg 0: DMA + row requests
        // ---- S front ----
        float sva[16], svb[16];
        if constexpr (TS == T_SAT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { sva[i] = sa[i] + sk[i]; svb[i] = sb[i] + sk[i]; }
        } else {
            // (all 64 LDS reads first, the adds behind a scheduling fence: with the adds in reach the scheduler — out of registers — waits for every
            //  pair of reads before it issues the next, 1700 cycles of exposed LDS latency per step)
            int rbo = rr_base;
            asm volatile("" : "+v"(rbo));
            float ga[16], gb[16], ha[16], hb[16];
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const int kc = 16 * (i >> 3) + (i & 7);
                const int prow = 16 * (i >> 3) + 8 * ((i >> 2) & 1) + (i & 3);
                const float* ia = img + (prow + 4 * h) * LROWP + 64 * wave + rr_base - kc;
                const int ra = c * LROW + ((rbo - kc) ^ xr), rb = c * LROW + ((rbo - kc - 1) ^ xr);
                ga[i] = ring_a[ra]; ga[i + 1] = ring_a[rb];
                gb[i] = ring_b[ra]; gb[i + 1] = ring_b[rb];
                ha[i] = ia[0]; ha[i + 1] = ia[LROWP - 1];
                hb[i] = ia[32]; hb[i + 1] = ia[LROWP + 31];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                sva[i] = (sa[i] + ga[i]) + ha[i]; sva[i + 1] = (sa[i + 1] + ga[i + 1]) + ha[i + 1];
                svb[i] = (sb[i] + gb[i]) + hb[i]; svb[i + 1] = (sb[i + 1] + gb[i + 1]) + hb[i + 1];
            }
        }
        const int k0 = kt * 32;
        if (k0 + 32 > kfirst) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(kb + k0 + foff);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(kb + k0 + foff + 4);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff);
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sva[i] += b0[i]; sva[4 + i] += b1[i]; sva[8 + i] += b2[i]; sva[12 + i] += b3[i];
                svb[i] += b0[i]; svb[4 + i] += b1[i]; svb[8 + i] += b2[i]; svb[12 + i] += b3[i];
            }
        }
        const float mxa = row_max(sva), mxb = row_max(svb);
        if constexpr (ST) stamp(1);                              // seg 1: gathers, bias, maxima
        if constexpr (TS != T_SAT && TM == T_BAND) wg_barrier_lds();      // X': every wave has gathered image(kt): M(kt + 1) may store image(kt + 1)
        if (__builtin_amdgcn_ballot_w64(mxa - A.m > RESCALE_THR || mxb - B.m > RESCALE_THR) != 0ull) {
            if (__builtin_amdgcn_ballot_w64(mxa - A.m > RESCALE_THR) != 0ull) rescale(mxa, A);
            if (__builtin_amdgcn_ballot_w64(mxb - B.m > RESCALE_THR) != 0ull) rescale(mxb, B);
        }
        if constexpr (ST) stamp(2);                              // seg 2: barrier X' (+ rescale)
        // ---- the fused block ----
        f16x8 pfa[2], pfb[2];
        i32x8 pxa, pxb;
        if constexpr (TS == T_BAND && TM == T_BAND) {
            // Hand-placed order (profiles/r05/mfma_valu_overlap_probe.txt: one wave DOES run VALU under its own MFMAs — if the instructions alternate in
            // program order and consecutive MFMAs belong to different accumulators; a dependent MFMA at the head of the stream stalls everything behind it).
            // Every line below is one scheduling region: a slice of the softmax, then one MFMA of each of the chains that are open.
#define GLC_FENCE() __builtin_amdgcn_sched_barrier(0)
            const int t1 = kt + 1;
            f32x2 psa = {0.f, 0.f}, psb = {0.f, 0.f};
            load_rows(PKg, ody, pkN);                            // the rows of L_A(kt + 1): used by the c2p blocks at the step's end
            k_tile(t1, kf);
#pragma unroll
            for (int i = 0; i < 16; ++i) { sa[i] = 0.f; sb[i] = 0.f; }
            f32x16 b0, b1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { b0[i] = 0.f; b1[i] = 0.f; }
            sm_step(0, sva, A, pfa, pxa, psa); GLC_FENCE();
            sm_step(1, sva, A, pfa, pxa, psa); GLC_FENCE();
            // segment A: p2c block of the first resident set (X) and S^T of tile A (Y) under the softmax of tile A
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mf(b0, P0.f[u], kf.f[u]); sm_step(2 + 2 * u, sva, A, pfa, pxa, psa); GLC_FENCE();
                mf(sa, kf.f[u], A.qf.f[u]); sm_step(3 + 2 * u, sva, A, pfa, pxa, psa); GLC_FENCE();
            }
            ms_hl(b0, cat8(P0.xa[0], P0.xb[0]), kf.x[0]); sm_step(10, sva, A, pfa, pxa, psa); GLC_FENCE();
            ms_lh(sa, kf.x[0], A.qf.x[0]); sm_step(11, sva, A, pfa, pxa, psa); GLC_FENCE();
            ms_hl(b0, cat8(P0.xa[1], P0.xb[1]), kf.x[1]);
            ms_lh(sa, kf.x[1], A.qf.x[1]); GLC_FENCE();
            M_band_r0(t1, eq);
            if (kt + 1 < nkt) dma_v(kt + 1);
            if (kt + 2 < kt_b && extra_wave(kt + 2) == wave) stage_rows(odx, px_lds + (size_t)(kt & 1) * TILEB);
            if constexpr (ST) stamp(3);
            GLC_FENCE();
            // segment B: second resident set (X'), S^T of tile B (Y'), P.V of tile A (Z: o0 then o1) under the softmax of tile B
            VHalf v;
            load_vh(kt, 0, v);
            band_store(img + c * LROWP + 32 * ((wave + t1) & 7), b0);
            sm_step(0, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(b1, P1.f[0], kf.f[0]); sm_step(1, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(A.o0, v.f[0], pfa[0]); sm_step(2, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(sb, kf.f[0], B.qf.f[0]); sm_step(3, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(A.o0, v.f[1], pfa[1]); sm_step(4, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(b1, P1.f[1], kf.f[1]); sm_step(5, svb, B, pfb, pxb, psb); GLC_FENCE();
            ms_lh(A.o0, v.x, pxa); sm_step(6, svb, B, pfb, pxb, psb); GLC_FENCE();
            load_vh(kt, 1, v);
            mf(sb, kf.f[1], B.qf.f[1]); sm_step(7, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(b1, P1.f[2], kf.f[2]); sm_step(8, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(A.o1, v.f[0], pfa[0]); sm_step(9, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(sb, kf.f[2], B.qf.f[2]); sm_step(10, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(A.o1, v.f[1], pfa[1]); sm_step(11, svb, B, pfb, pxb, psb); GLC_FENCE();
            mf(b1, P1.f[3], kf.f[3]);
            ms_lh(A.o1, v.x, pxa);
            mf(sb, kf.f[3], B.qf.f[3]); GLC_FENCE();
            ms_hl(b1, cat8(P1.xa[0], P1.xb[0]), kf.x[0]);
            ms_lh(sb, kf.x[0], B.qf.x[0]); GLC_FENCE();
            ms_hl(b1, cat8(P1.xa[1], P1.xb[1]), kf.x[1]);
            ms_lh(sb, kf.x[1], B.qf.x[1]); GLC_FENCE();
            M_band_r1(t1, eq);
            if constexpr (ST) stamp(4);
            GLC_FENCE();
            // segment C: P.V of tile B (o0, o1), c2p blocks of both tiles (ca, cbn), the ninth block: four chains, no VALU
            band_store(img + c * LROWP + 32 * ((wave + t1 + 4) & 7), b1);
            band_store(ring_b + c * LROW + (xr ^ 32), cbn);      // L_B(kt + 1), computed a step ago
            f32x16 ca;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ca[i] = 0.f; cbn[i] = 0.f; }
            load_vh(kt, 0, v);
            mf(ca, pkN.f[0], A.qf.f[0]); mf(cbn, pkN.f[0], B.qf.f[0]); mf(B.o0, v.f[0], pfb[0]); GLC_FENCE();
            mf(ca, pkN.f[1], A.qf.f[1]); mf(cbn, pkN.f[1], B.qf.f[1]); mf(B.o0, v.f[1], pfb[1]); GLC_FENCE();
            mf(ca, pkN.f[2], A.qf.f[2]); mf(cbn, pkN.f[2], B.qf.f[2]); ms_lh(B.o0, v.x, pxb); GLC_FENCE();
            load_vh(kt, 1, v);
            mf(ca, pkN.f[3], A.qf.f[3]); mf(cbn, pkN.f[3], B.qf.f[3]); mf(B.o1, v.f[0], pfb[0]); GLC_FENCE();
            ms_lh(ca, pkN.x[0], A.qf.x[0]); ms_lh(cbn, pkN.x[0], B.qf.x[0]); mf(B.o1, v.f[1], pfb[1]); GLC_FENCE();
            ms_lh(ca, pkN.x[1], A.qf.x[1]); ms_lh(cbn, pkN.x[1], B.qf.x[1]); ms_lh(B.o1, v.x, pxb); GLC_FENCE();
            M_band_x(t1);
            band_store(ring_a + c * LROW + (xr ^ 32), ca);
#undef GLC_FENCE
        } else {
        if constexpr (TS == T_BAND) load_rows(PKg, ody, pkN);   // (the rows of L_A(kt + 1): used by the c2p blocks at the step's end)
        if constexpr (TM == T_SAT) M_sat(kt + 1);
        if constexpr (TM == T_BAND) M_band_a(kt + 1);
        softmax_split(sva, A, pfa, pxa);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TM == T_BAND) M_band_r0(kt + 1, eq);
        if (kt + 1 < nkt) dma_v(kt + 1);
        if constexpr (TS == T_BAND) { if (kt + 2 < kt_b && extra_wave(kt + 2) == wave) stage_rows(odx, px_lds + (size_t)(kt & 1) * TILEB); }
        if constexpr (ST) stamp(3);                              // seg 3: M first half || exponentials + split of tile A
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TM == T_BAND) M_band_b(kt + 1);
        pv(kt, A, pfa, pxa);
        softmax_split(svb, B, pfb, pxb);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TM == T_BAND) { M_band_r1(kt + 1, eq); M_band_x(kt + 1); }
        if constexpr (ST) stamp(4);                              // seg 4: M second half || P.V of A || exponentials + split of B (+ the ninth block)
        __builtin_amdgcn_sched_barrier(0);
        pv(kt, B, pfb, pxb);
        if constexpr (TS == T_BAND) {
            band_store(ring_b + c * LROW + (xr ^ 32), cbn);      // L_B(kt + 1), computed a step ago
            f32x16 ca;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ca[i] = 0.f; cbn[i] = 0.f; }
            mm_lh_hl(pkN, A.qf, ca);                             // c2p of L_A(kt + 1)
            mm_lh_hl(pkN, B.qf, cbn);                            // c2p of L_B(kt + 2) = L_A(kt + 1)
            band_store(ring_a + c * LROW + (xr ^ 32), ca);
        }
        }
        if constexpr (ST) stamp(5);                              // seg 5: P.V of B, c2p blocks, ring stores
        if constexpr (TM != T_NONE) wg_barrier_all();            // Y'
        if constexpr (ST) { stamp(6); if constexpr (DIAG) ++tiles; }      // seg 6: requests landed + barrier Y'
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // ---- prologue: K(0), K(1), V^T(0) in LDS; then the three phases as loops of their own (a phase's operands are dead outside it) ----
    if (nkt > 1) dma_k(1);
    wg_barrier_all();
    const bool has_band = kt_a < kt_b;
    int kt = 0;
    if (kt_a > 0 || !has_band) {        // saturated tiles first (no band at all: tiles below kt_a = kt_b use d* = P - 1, the others d* = 0)
        setup_sat(kt_a > 0 ? a.P - 1 : 0);
        M_sat(0);
        wg_barrier_all();
        const int end_lo = has_band ? kt_a - 1 : nkt - 1;
        for (; kt < end_lo; ++kt) {
            if (!has_band && kt + 1 == kt_b) setup_sat(0);
            step(I0{}, I0{}, 0, kt);
        }
        if (has_band) { setup_band(); step(I0{}, I1{}, 0, kt); ++kt; }
        else { step(I0{}, I3{}, 0, kt); ++kt; }
    } else {
        setup_band();
        M_band(0, eq_n);
        eq_n = block_x(Q0, 2);
        wg_barrier_all();
    }
    if (has_band) {
        for (; kt + 1 < kt_b; ++kt) step(I1{}, I1{}, ((kt - kt_a) & 1) * 32, kt);
        const bool odd = ((kt - kt_a) & 1) != 0;
        if (kt_b < nkt) {
            setup_sat(0);
            step(I2{}, I0{}, odd ? 32 : 0, kt);
            ++kt;
            for (; kt + 1 < nkt; ++kt) step(I0{}, I0{}, 0, kt);
            step(I0{}, I3{}, 0, kt);
        } else step(I2{}, I3{}, odd ? 32 : 0, kt);
    }

    if constexpr (DIAG) {
        if (a.stamps && blockIdx.x < 64 * 8 && (blockIdx.x & 7) == 0 && lane == 0) {
            unsigned long long* o = a.stamps + ((size_t)(blockIdx.x >> 3) * 8 + wave) * 10;
            for (int k = 0; k < 8; ++k) o[k] = seg[k];
            o[8] = 0; o[9] = tiles;
        }
    }
    if (!active) return;
    unsigned char* rowp = reinterpret_cast<unsigned char*>(a.CTX) + ((size_t)b * Sp + q0 + c) * 4 * a.H;
    auto store_q = [&](QState& q, unsigned char* row) __attribute__((always_inline)) {
        q.l += __shfl_xor(q.l, 32, 64);
        const float inv = 1.0f / q.l;
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
        store_gx(q.o0, 64 * hh);
        store_gx(q.o1, 64 * hh + 32);
    };
    store_q(A, rowp);
    store_q(B, rowp + (size_t)32 * 4 * a.H);
}

}  // namespace

const char* glc_launch_attention_mxd(hipStream_t st, const AttnArgs& a_in) {
    AttnArgs a = a_in;
    if (!a.gx_sat) a.gx_sat = glc_gx_sat_ptr();
    if (!a.act_sc) a.act_sc = glc_gx_act_sc();
    if (!a.Qh || !a.Kh || !a.Vt || !a.PK || !a.PQ || !a.kbias || !a.klen || !a.kfirst || !a.CTX || !a.otab) return "attention(mxd): null pointer";
    if (a.B <= 0 || a.nh <= 0 || a.Sp <= 0 || a.Sp % 64 || a.H != a.nh * 64 || a.P <= 0 || a.P % 32) return "attention(mxd): bad shape";
    if (a.sel_b || a.tile_flag) return "attention(mxd): no row selection in this kernel";
    static std::atomic<unsigned> r0{0}, r1{0};
    const int nqb = (a.Sp + 32 * NQT - 1) / (32 * NQT), bh8 = (a.B * a.nh + 7) / 8 * 8;
    if (a.stamps) {
        if (!glc_raise_lds_limit(attn_mxd_kernel<true>, (int)LDS_BYTES, r1)) return "attention(mxd): cannot raise the dynamic LDS limit";
        hipLaunchKernelGGL(attn_mxd_kernel<true>, dim3(nqb * bh8), dim3(256), LDS_BYTES, st, a);
        return nullptr;
    }
    if (!glc_raise_lds_limit(attn_mxd_kernel<false>, (int)LDS_BYTES, r0)) return "attention(mxd): cannot raise the dynamic LDS limit";
    hipLaunchKernelGGL(attn_mxd_kernel<false>, dim3(nqb * bh8), dim3(256), LDS_BYTES, st, a);
    return nullptr;
}
