// Decoder-style backbone (Qwen2 arithmetic, decoder.hip), the attention of the MX pipeline (round 4): grouped-query causal attention on
// "MX tiles" — every product a*b = a_hi*b_hi in f16 MFMAs + (a_hi*b_lo + a_lo*b_hi) in ONE block-scaled fp8 MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4), two instead of the three f16-MFMA times of the split-unit kernel (attn_gqa_mfma_kernel<float, D, true>):
// the arithmetic of attention_mx.hip, in the work split of decoder.hip (one independent wave = 32 queries of one query head, four waves
// per block, no LDS, no barrier; K set re-loaded in place; causal tiles above the diagonal never visited).
//
// Operand layout per (batch, head), head_dim D in {64, 128}, NS = D / 16 f16 units and NM = D / 32 MX steps per 32-row tile:
//   Q, K : tile of 32 rows = [NS x 1 KiB f16 units | NM x 2 KiB MX steps]                      (32 D 4 bytes: what the split units took)
//          f16 unit s, lane 32 hh + r -> hi halves of columns 16 s + 8 hh + j; MX step m, lane 32 hh + r -> 32 bytes [first | second] =
//          the fp8 parts of columns 32 m + 16 hh + y; Q travels as (hi8 | lo8), K as (lo8 | hi8), K rows at slot pi(r)      (glc_layout.h)
//   V^T  : per 32-key tile D / 32 sub-tiles of 4 KiB (32 rows dd each): [f16 unit t = 0 | t = 1 | one MX step], (lo8 | hi8)   (glc_layout.h)
// written by qkv_layout_mx_kernel from the fused QKV rows (plain fp32): RoPE on the rotate-half pairs and the softmax scale on Q in fp32,
// then hi = f16(x), hi8 = e4m3(x), lo8 = e4m3((x - hi) 2^GLC_GX_SHIFT) (saturating; the fp8 range guard of glc_common.h counts |x| > 448).
#include "glc_common.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

constexpr float DEC_RESCALE_THR = 8.0f;   // log2 units (attention.hip)
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ i32x8 cat8(const i32x4& a, const i32x4& b) {
    i32x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

template <int D>
__global__ __launch_bounds__(256) void qkv_layout_mx_kernel(const float* __restrict__ QKV, const float* __restrict__ cs, unsigned char* __restrict__ Qm,
                                                            unsigned char* __restrict__ Km, unsigned char* __restrict__ Vm, int Sp, int nq, int nkv, float qscale,
                                                            unsigned* sat) {
    constexpr int HD2 = D / 2, NS = D / 16, TILE = 32 * D * 4;
    __shared__ float vs[32][D + 4];
    const int tile = blockIdx.x, head = blockIdx.y, t = threadIdx.x;           // tile over all B*Sp/32 row tiles
    const int ld = (nq + 2 * nkv) * D;
    const int m0 = tile * 32, b = m0 / Sp, st = (m0 - b * Sp) >> 5, nt = Sp >> 5;
    if (head < nq + nkv) {
        // ---- Q or K head: RoPE on the (i, i + D/2) pairs, 8 consecutive i per thread (decoder.hip qkv_layout_kernel) ----
        const bool isq = head < nq;
        for (int idx = t; idx < 32 * (HD2 / 8); idx += 256) {
            const int r = idx / (HD2 / 8), c8 = idx - r * (HD2 / 8);
            const int s = (m0 - b * Sp) + r;
            const float* src = QKV + (size_t)(m0 + r) * ld + (size_t)head * D + c8 * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(src + HD2), b1 = *reinterpret_cast<const f32x4*>(src + HD2 + 4);
            const float x1[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, x2[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            const float* c = cs + ((size_t)s * HD2 + c8 * 8) * 2;
            const float sc = isq ? qscale : 1.f;
            float o1[8], o2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float co = c[2 * j], sn = c[2 * j + 1];
                o1[j] = (x1[j] * co - x2[j] * sn) * sc;
                o2[j] = (x2[j] * co + x1[j] * sn) * sc;
            }
            const int slot = isq ? r : glc_pi32(r);                            // K rows sit at slot pi(r)
            unsigned char* base = (isq ? Qm + ((size_t)(b * nq + head) * nt + st) * TILE : Km + ((size_t)(b * nkv + (head - nq)) * nt + st) * TILE);
            const int d1 = c8 * 8, d2 = d1 + HD2;
            auto f16_at = [&](int e0) { return base + (e0 >> 4) * 1024 + (32 * ((e0 >> 3) & 1) + slot) * 16; };
            auto mx_at = [&](int e0) { return base + NS * 1024 + (e0 >> 5) * 2048 + (32 * ((e0 >> 4) & 1) + slot) * 32 + 8 * ((e0 >> 3) & 1); };
            store_mx8(f16_at(d1), mx_at(d1), o1, isq, sat);
            store_mx8(f16_at(d2), mx_at(d2), o2, isq, sat);
        }
    } else {
        // ---- V head: transpose the 32 x D tile through LDS; 8 consecutive keys of one row dd per thread ----
        const int g = head - nq - nkv;
        for (int idx = t; idx < 32 * (D / 4); idx += 256) {
            const int r = idx / (D / 4), c4 = idx - r * (D / 4);
            *reinterpret_cast<f32x4*>(&vs[r][c4 * 4]) = *reinterpret_cast<const f32x4*>(QKV + (size_t)(m0 + r) * ld + (size_t)(nq + nkv + g) * D + c4 * 4);
        }
        __syncthreads();
        unsigned char* base = Vm + ((size_t)(b * nkv + g) * nt + st) * TILE;
        for (int u = t; u < D * 4; u += 256) {                                 // (dd, kg): row dd of V^T, keys 8 kg .. 8 kg + 7 of the tile
            const int dd = u >> 2, kg = u & 3;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = vs[8 * kg + j][dd];
            unsigned char* sub = base + (dd >> 5) * 4096;
            store_mx8(sub + (kg >> 1) * 1024 + (32 * (kg & 1) + (dd & 31)) * 16, sub + 2048 + (32 * (kg & 1) + (dd & 31)) * 32 + 8 * (kg >> 1), o, false, sat);
        }
    }
}

// Two waves per SIMD: the operand sets of head_dim 128 (Q 64, K 64, O^T 64 registers, V^T 16 per sub-tile, read where it is used) fit 256
// registers — the split-unit kernel, with three fragment sets of twice the size, runs one.
template <int D>
__global__ __launch_bounds__(256, 2) void attn_gqa_mx_kernel(const unsigned char* __restrict__ Qm, const unsigned char* __restrict__ Km, const unsigned char* __restrict__ Vm,
                                                             const float* __restrict__ kbias, const int* __restrict__ klen, const int* __restrict__ kfirst_,
                                                             unsigned char* __restrict__ CTX, int B, int Sp, int nq, int nkv, int causal, unsigned* gx_sat, int act_sc) {
    constexpr int NS = D / 16, NM = D / 32, ND = D / 32, TILE = 32 * D * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int nt = Sp >> 5, nqb = (nt + 3) >> 2, grp = nq / nkv;
    const int SC = h ? (127 | ((127 - GLC_GX_SHIFT) << 8)) : ((127 - GLC_GX_SHIFT) | (127 << 8));      // attention_mx.hip: one scale register, picked by op_sel
    // XCD-aware decode (decoder.hip): every block of one (batch, kv group) on one XCD
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int per = grp * nqb;
    const int bg = xcd + 8 * (jj / per), rem = jj % per;
    if (bg >= B * nkv) return;
    const int b = bg / nkv, g = bg - b * nkv;
    const int hq = g * grp + rem / nqb;
    const int qt = nt - 1 - ((rem % nqb) * 4 + wave);       // longest tiles first within a head
    if (qt < 0) return;
    const int q0 = qt * 32;

    const unsigned char* __restrict__ Qp = Qm + ((size_t)(b * nq + hq) * nt + qt) * TILE;
    const unsigned char* __restrict__ Kp = Km + (size_t)(b * nkv + g) * nt * TILE;
    const unsigned char* __restrict__ Vp = Vm + (size_t)(b * nkv + g) * nt * TILE;
    const float* __restrict__ kb = kbias + (size_t)b * Sp;
    unsigned char* row = CTX + ((size_t)b * Sp + q0 + c) * 4 * ((size_t)nq * D);

    if (q0 >= klen[b] && q0 > 0) {           // padding-only query tile of a ragged batch: no attended row reads it; store zeros (GX rows)
        unsigned char* z = row + (size_t)hq * D * 4 + h * (D * 2);
#pragma unroll
        for (int i = 0; i < D / 8; ++i) *reinterpret_cast<u32x4*>(z + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
        return;
    }
    int nkt = (klen[b] + 31) >> 5;
    nkt = nkt < 1 ? 1 : (nkt > nt ? nt : nkt);
    if (causal && nkt > qt + 1) nkt = qt + 1;
    const int kfirst = kfirst_[b];
    const int foff = 8 * h;

    f16x8 qf[NS], kf[NS];
    i32x8 qx[NM], kx[NM];
    auto load_tile = [&](const unsigned char* tile, f16x8 (&f)[NS], i32x8 (&x)[NM]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) f[s] = *reinterpret_cast<const f16x8*>(tile + s * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < NM; ++m) x[m] = cat8(*reinterpret_cast<const i32x4*>(tile + NS * 1024 + m * 2048 + lane * 32), *reinterpret_cast<const i32x4*>(tile + NS * 1024 + m * 2048 + lane * 32 + 16));
    };
    load_tile(Qp, qf, qx);
    load_tile(Kp, kf, kx);
    f32x16 o[ND];
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[a][i] = 0.f;
    float m = -3.0e38f, l = 0.f;
    float one_f = 1.0f;
    asm volatile("" : "+s"(one_f));          // opaque to the optimiser: fma(p, 1, -half) stays a v_fma_mix_f32 (attention_mx.hip)

    for (int kt = 0; kt < nkt; ++kt) {
        const int ktn = kt + 1 < nkt ? kt + 1 : kt;
        // S^T = K Q^T ; reg i <-> key k0 + 16*(i>>3) + 8h + (i&7), column = query c
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qf[s], sacc, 0, 0, 0);
#pragma unroll
        for (int mm = 0; mm < NM; ++mm) sacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kx[mm], qx[mm], sacc, 0, 0, 0, SC, 1, SC);
        load_tile(Kp + (size_t)ktn * TILE, kf, kx);          // the next K set, in place, right behind its last MFMA
        float sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = sacc[i];
        const int k0 = kt * 32;
        if (k0 + 32 > kfirst) {                                             // wave-uniform: tile holds masked keys
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(kb + k0 + foff);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(kb + k0 + foff + 4);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff);
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sv[i] += b0[i]; sv[4 + i] += b1[i]; sv[8 + i] += b2[i]; sv[12 + i] += b3[i]; }
        }
        if (causal && kt == qt) {                                           // diagonal tile: key offset > query offset is masked
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ko = 16 * (i >> 3) + foff + (i & 7);
                if (ko > c) sv[i] = GLC_NEG_BIG;
            }
        }
        float mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, sv[i]), sv[i + 1]);
        mx = fmaxf(mx, sv[15]);
        if (__builtin_amdgcn_ballot_w64(mx - m > DEC_RESCALE_THR) != 0ull) {   // deferred rescale (attention.hip)
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            l *= alpha;
#pragma unroll
            for (int a = 0; a < ND; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[a][i] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(sv[i] - m); psum += sv[i]; }
        l += psum;
        // P travels as (hi8 | lo8): f16(p) for the f16 MFMAs, the fp8 parts of the 16 keys for the scaled one (attention_mx.hip softmax_pv)
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
        const unsigned char* vtile = Vp + (size_t)kt * TILE;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            const f16x8 v0 = *reinterpret_cast<const f16x8*>(vtile + a * 4096 + lane * 16);
            const f16x8 v1 = *reinterpret_cast<const f16x8*>(vtile + a * 4096 + 1024 + lane * 16);
            const i32x8 vx = cat8(*reinterpret_cast<const i32x4*>(vtile + a * 4096 + 2048 + lane * 32), *reinterpret_cast<const i32x4*>(vtile + a * 4096 + 2048 + lane * 32 + 16));
            o[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf[0], o[a], 0, 0, 0);      // O^T[dd = 32a + (i&3) + 8(i>>2) + 4h][query c]
            o[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf[1], o[a], 0, 0, 0);
            o[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o[a], 0, 0, 0, SC, 1, SC);
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // GX context rows (glc_common.h; the A operand of the MX cross-term GEMM): one cross-half exchange per register (decoder.hip)
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float own_a = o[a][8 * p + e] * inv, own_b = o[a][8 * p + 4 + e] * inv;
                const float got = __shfl_xor(h ? own_a : own_b, 32, 64);
                v[e] = h ? got : own_a;
                v[4 + e] = h ? own_b : got;
            }
            gx_store8(row, hq * D + 32 * a + 16 * p + 8 * h, v, gx_act_khi(act_sc), gx_act_klo(act_sc), gx_sat);
        }
}


// ---- the same attention with K and V^T through LDS (the default) ----
// attn_gqa_mx_kernel above loads every K and V^T tile per wave: 32 KiB per wave and key tile at head_dim 128 through the CU's 64 B/clk
// vector-memory path — 4096 cycles per tile round of 8 waves against 2048 of matrix pipe: the path, not the pipe, bounds it.  The four waves
// of a workgroup are four consecutive query tiles of ONE query head, so they walk the same K / V^T tiles: here each tile is fetched once
// per workgroup by LDS-DMA (8 one-KiB pieces per wave and tile instead of 32 loads) into two-slot rings (K 2 x 16 KiB, V^T 2 x 16 KiB at
// head_dim 128; two workgroups per CU) and read as fragments by all four, with the protocol of attention_mx2.hip: ONE workgroup barrier per
// key tile — after barrier(t) every wave holds its K(t) fragments and has left V^T(t - 1), so V^T(t + 1) and K(t + 2) are requested there, each
// wave waits for its own pieces right before barrier(t + 1), which publishes them.  Causal: a wave's query tile ends the walk at its own
// diagonal; waves that are done (or have no tile, or a padding-only one) keep moving their pieces and meeting the barriers until the
// workgroup's longest walk ends.  MX steps land re-arranged as [64 lanes x first 16 B | 64 lanes x second] (per-lane DMA source address).
extern __shared__ __attribute__((aligned(16))) unsigned char smem_dmx[];

__device__ __forceinline__ void glds16_sv(const unsigned char* ubase, unsigned lane_off, void* l) {      // attention_wg.hip
    const unsigned la = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)l;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la), "v"(lane_off), "s"(ubase) : "memory");
}

template <int D>
__global__ __launch_bounds__(256, 2) void attn_gqa_mx_ring_kernel(const unsigned char* __restrict__ Qm, const unsigned char* __restrict__ Km, const unsigned char* __restrict__ Vm,
                                                                  const float* __restrict__ kbias, const int* __restrict__ klen, const int* __restrict__ kfirst_,
                                                                  unsigned char* __restrict__ CTX, int B, int Sp, int nq, int nkv, int causal, unsigned* gx_sat, int act_sc) {
    constexpr int NS = D / 16, NM = D / 32, ND = D / 32, TILE = 32 * D * 4, P4 = D / 32;      // P4: one-KiB pieces per wave and tile of K (and of V^T)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int nt = Sp >> 5, nqb = (nt + 3) >> 2, grp = nq / nkv;
    const int SC = h ? (127 | ((127 - GLC_GX_SHIFT) << 8)) : ((127 - GLC_GX_SHIFT) | (127 << 8));
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int per = grp * nqb;
    const int bg = xcd + 8 * (jj / per), rem = jj % per;
    if (bg >= B * nkv) return;                               // (workgroup-uniform)
    const int b = bg / nkv, g = bg - b * nkv;
    const int hq = g * grp + rem / nqb;
    const int qt = nt - 1 - ((rem % nqb) * 4 + wave);       // longest tiles first within a head; wave 0 holds the workgroup's last query tile
    const int q0 = qt * 32;
    const int kl = klen[b];
    int nkt_len = (kl + 31) >> 5;
    nkt_len = nkt_len < 1 ? 1 : (nkt_len > nt ? nt : nkt_len);
    const bool has_tile = qt >= 0;
    const bool pad_only = has_tile && q0 >= kl && q0 > 0;   // padding-only query tile of a ragged batch: zeros out, no walk
    const bool active = has_tile && !pad_only;
    const int nkt = !active ? 0 : (causal && nkt_len > qt + 1 ? qt + 1 : nkt_len);       // this wave's walk
    // the workgroup's walk = the longest of its waves' (walks grow with the query tile: the first active wave has it)
    __shared__ int s_nkt[4];
    if (lane == 0) s_nkt[wave] = nkt;
    __syncthreads();
    const int nkt_wg = max(max(s_nkt[0], s_nkt[1]), max(s_nkt[2], s_nkt[3]));

    unsigned char* row = CTX + ((size_t)b * Sp + (has_tile ? q0 : 0) + c) * 4 * ((size_t)nq * D);
    if (pad_only) {
        unsigned char* z = row + (size_t)hq * D * 4 + h * (D * 2);
#pragma unroll
        for (int i = 0; i < D / 8; ++i) *reinterpret_cast<u32x4*>(z + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
    }
    if (nkt_wg == 0) return;                                 // (workgroup-uniform: nobody walks)

    const unsigned char* __restrict__ Qp = Qm + ((size_t)(b * nq + hq) * nt + (has_tile ? qt : 0)) * TILE;
    const unsigned char* __restrict__ Kp = Km + (size_t)(b * nkv + g) * nt * TILE;
    const unsigned char* __restrict__ Vp = Vm + (size_t)(b * nkv + g) * nt * TILE;
    const float* __restrict__ kb = kbias + (size_t)b * Sp;
    const int kfirst = kfirst_[b];
    const int foff = 8 * h;
    unsigned char* const kring = smem_dmx;                   // 2 x TILE
    unsigned char* const vring = smem_dmx + 2 * TILE;        // 2 x TILE

    // LDS-DMA: wave w moves pieces P4 w .. P4 w + P4 - 1 of the K tile and of the V^T tile
    const unsigned off16 = lane * 16, off32 = lane * 32;
    auto uniform_ptr = [](const unsigned char* q) -> const unsigned char* {
        const unsigned long long v = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
    };
    auto dma_k = [&](int t) __attribute__((always_inline)) {
        const unsigned char* src = Kp + (size_t)t * TILE;
        unsigned char* dst = kring + (size_t)(t & 1) * TILE;
#pragma unroll
        for (int i = 0; i < P4; ++i) {
            const int p = P4 * wave + i;                     // K tile: pieces 0 .. NS - 1 the f16 units, then two per MX step
            if (p < NS) glds16_sv(uniform_ptr(src + p * 1024), off16, dst + p * 1024);
            else { const int q = p - NS; glds16_sv(uniform_ptr(src + NS * 1024 + (q >> 1) * 2048 + 16 * (q & 1)), off32, dst + NS * 1024 + (q >> 1) * 2048 + (q & 1) * 1024); }
        }
    };
    auto dma_v = [&](int t) __attribute__((always_inline)) {
        const unsigned char* src = Vp + (size_t)t * TILE;
        unsigned char* dst = vring + (size_t)(t & 1) * TILE;
#pragma unroll
        for (int i = 0; i < P4; ++i) {
            const int p = P4 * wave + i, a = p >> 2, r = p & 3;      // V^T tile: per 4-KiB sub-tile [f16 unit | f16 unit | MX step = two pieces]
            if (r < 2) glds16_sv(uniform_ptr(src + a * 4096 + r * 1024), off16, dst + a * 4096 + r * 1024);
            else glds16_sv(uniform_ptr(src + a * 4096 + 2048 + 16 * (r - 2)), off32, dst + a * 4096 + 2048 + (r - 2) * 1024);
        }
    };

    f16x8 qf[NS];
    i32x8 qx[NM];
    if (active) {
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const f16x8*>(Qp + s * 1024 + lane * 16);
#pragma unroll
        for (int mm = 0; mm < NM; ++mm) qx[mm] = cat8(*reinterpret_cast<const i32x4*>(Qp + NS * 1024 + mm * 2048 + lane * 32), *reinterpret_cast<const i32x4*>(Qp + NS * 1024 + mm * 2048 + lane * 32 + 16));
    }
    dma_k(0);
    dma_v(0);
    if (nkt_wg > 1) dma_k(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // K(0) is in the ring for everyone (K(1), V^T(0): published by barrier(0))

    f32x16 o[ND];
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[a][i] = 0.f;
    float m = -3.0e38f, l = 0.f;
    float one_f = 1.0f;
    asm volatile("" : "+s"(one_f));

    for (int kt = 0; kt < nkt_wg; ++kt) {
        const bool work = kt < nkt;                          // wave-uniform: this wave's walk still covers key tile kt
        float sv[16];
        if (work) {
            const unsigned char* ktile = kring + (size_t)(kt & 1) * TILE;
            f32x16 sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(ktile + s * 1024 + lane * 16), qf[s], sacc, 0, 0, 0);
#pragma unroll
            for (int mm = 0; mm < NM; ++mm) {
                const i32x8 kx = cat8(*reinterpret_cast<const i32x4*>(ktile + NS * 1024 + mm * 2048 + lane * 16), *reinterpret_cast<const i32x4*>(ktile + NS * 1024 + mm * 2048 + 1024 + lane * 16));
                sacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kx, qx[mm], sacc, 0, 0, 0, SC, 1, SC);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = sacc[i];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // my pieces of V^T(kt) and K(kt + 1), requested a tile ago; my K(kt) fragment reads
        __builtin_amdgcn_s_barrier();                                    // barrier(kt)
        if (kt + 1 < nkt_wg) dma_v(kt + 1);
        if (kt + 2 < nkt_wg) dma_k(kt + 2);
        if (!work) continue;
        const int k0 = kt * 32;
        if (k0 + 32 > kfirst) {                                             // wave-uniform: tile holds masked keys
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(kb + k0 + foff);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(kb + k0 + foff + 4);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff);
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(kb + k0 + 16 + foff + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { sv[i] += b0[i]; sv[4 + i] += b1[i]; sv[8 + i] += b2[i]; sv[12 + i] += b3[i]; }
        }
        if (causal && kt == qt) {                                           // diagonal tile: key offset > query offset is masked
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ko = 16 * (i >> 3) + foff + (i & 7);
                if (ko > c) sv[i] = GLC_NEG_BIG;
            }
        }
        float mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, sv[i]), sv[i + 1]);
        mx = fmaxf(mx, sv[15]);
        if (__builtin_amdgcn_ballot_w64(mx - m > DEC_RESCALE_THR) != 0ull) {   // deferred rescale (attention.hip)
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            l *= alpha;
#pragma unroll
            for (int a = 0; a < ND; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[a][i] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(sv[i] - m); psum += sv[i]; }
        l += psum;
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
        const unsigned char* vtile = vring + (size_t)(kt & 1) * TILE;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            const f16x8 v0 = *reinterpret_cast<const f16x8*>(vtile + a * 4096 + lane * 16);
            const f16x8 v1 = *reinterpret_cast<const f16x8*>(vtile + a * 4096 + 1024 + lane * 16);
            const i32x8 vx = cat8(*reinterpret_cast<const i32x4*>(vtile + a * 4096 + 2048 + lane * 16), *reinterpret_cast<const i32x4*>(vtile + a * 4096 + 3072 + lane * 16));
            o[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf[0], o[a], 0, 0, 0);
            o[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf[1], o[a], 0, 0, 0);
            o[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vx, px, o[a], 0, 0, 0, SC, 1, SC);
        }
    }
    if (!active) return;
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float own_a = o[a][8 * p + e] * inv, own_b = o[a][8 * p + 4 + e] * inv;
                const float got = __shfl_xor(h ? own_a : own_b, 32, 64);
                v[e] = h ? got : own_a;
                v[4 + e] = h ? own_b : got;
            }
            gx_store8(row, hq * D + 32 * a + 16 * p + 8 * h, v, gx_act_khi(act_sc), gx_act_klo(act_sc), gx_sat);
        }
}

}  // namespace

// RoPE + softmax scale + MX tiles of the fused fp32 projection.  Sp % 64 == 0, d in {64, 128}; Qm / Km / Vm: 4 bytes per element.
const char* glc_launch_qkv_layout_mx(hipStream_t st, const void* QKV, const float* cs, void* Qm, void* Km, void* Vm, int B, int Sp, int nq, int nkv, int d, float qscale) {
    if (!QKV || !cs || !Qm || !Km || !Vm || B <= 0 || Sp <= 0 || Sp % 64 || nq <= 0 || nkv <= 0 || (d != 64 && d != 128)) return "qkv_layout_mx: bad args";
    const dim3 grid(B * Sp / 32, nq + 2 * nkv), block(256);
    unsigned* sat = glc_gx_sat_ptr();
    if (d == 128) hipLaunchKernelGGL(qkv_layout_mx_kernel<128>, grid, block, 0, st, (const float*)QKV, cs, (unsigned char*)Qm, (unsigned char*)Km, (unsigned char*)Vm, Sp, nq, nkv, qscale, sat);
    else hipLaunchKernelGGL(qkv_layout_mx_kernel<64>, grid, block, 0, st, (const float*)QKV, cs, (unsigned char*)Qm, (unsigned char*)Km, (unsigned char*)Vm, Sp, nq, nkv, qscale, sat);
    return nullptr;
}

// Grouped-query attention on those MX tiles; CTX [B*Sp, nq*d] as GX rows.
const char* glc_launch_attention_gqa_mx(hipStream_t st, const void* Qm, const void* Km, const void* Vm, const float* kbias, const int* klen, const int* kfirst, void* CTX,
                                        int B, int Sp, int nq, int nkv, int d, int causal) {
    if (!Qm || !Km || !Vm || !kbias || !klen || !kfirst || !CTX || B <= 0 || Sp <= 0 || Sp % 64 || nq <= 0 || nkv <= 0 || nq % nkv || (d != 64 && d != 128))
        return "attention_gqa_mx: bad args";
    const int nt = Sp / 32, nqb = (nt + 3) / 4, per = (nq / nkv) * nqb, bg8 = (B * nkv + 7) / 8 * 8;
    const dim3 grid(per * bg8), block(256);
    unsigned* sat = glc_gx_sat_ptr();
    static const bool direct = glc_dev_env("GLC_DEC_ATTN_DIRECT") && atoi(glc_dev_env("GLC_DEC_ATTN_DIRECT")) != 0;      // developer A/B: K / V^T per wave from memory
    if (!direct) {
        const size_t lds = (size_t)4 * 32 * d * 4;            // K ring + V^T ring, two slots each
        static std::atomic<unsigned> ok128{0}, ok64{0};
        if (d == 128) {
            if (!glc_raise_lds_limit(attn_gqa_mx_ring_kernel<128>, (int)lds, ok128)) return "attention_gqa_mx: cannot raise the dynamic LDS limit";
            hipLaunchKernelGGL(attn_gqa_mx_ring_kernel<128>, grid, block, lds, st, (const unsigned char*)Qm, (const unsigned char*)Km, (const unsigned char*)Vm, kbias, klen, kfirst, (unsigned char*)CTX, B, Sp, nq, nkv, causal, sat, glc_gx_act_sc());
        } else {
            if (!glc_raise_lds_limit(attn_gqa_mx_ring_kernel<64>, (int)lds, ok64)) return "attention_gqa_mx: cannot raise the dynamic LDS limit";
            hipLaunchKernelGGL(attn_gqa_mx_ring_kernel<64>, grid, block, lds, st, (const unsigned char*)Qm, (const unsigned char*)Km, (const unsigned char*)Vm, kbias, klen, kfirst, (unsigned char*)CTX, B, Sp, nq, nkv, causal, sat, glc_gx_act_sc());
        }
        return nullptr;
    }
    if (d == 128) hipLaunchKernelGGL(attn_gqa_mx_kernel<128>, grid, block, 0, st, (const unsigned char*)Qm, (const unsigned char*)Km, (const unsigned char*)Vm, kbias, klen, kfirst, (unsigned char*)CTX, B, Sp, nq, nkv, causal, sat, glc_gx_act_sc());
    else hipLaunchKernelGGL(attn_gqa_mx_kernel<64>, grid, block, 0, st, (const unsigned char*)Qm, (const unsigned char*)Km, (const unsigned char*)Vm, kbias, klen, kfirst, (unsigned char*)CTX, B, Sp, nq, nkv, causal, sat, glc_gx_act_sc());
    return nullptr;
}
