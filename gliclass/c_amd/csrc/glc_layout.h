// Fragment-major HBM layouts of the attention operands.
//
// The QKV GEMM epilogue writes Q, K and V^T directly in the order the attention kernel's 32x32x16
// MFMA fragments consume them, so that every wave-level 16-byte-per-lane load is ONE contiguous
// 1 KiB (a lane-strided [row][64] layout makes each such load touch 32 half-used cache lines and
// the kernel L1/TA-bound).  Unit = 8 elements (16 B).  For a (batch*head) index bh and NT = Sp/32:
//
//   Q : [bh][qt ][s][lane = 32h + r][8]   holds Q [32qt + r    ][16s + 8h + j]
//   K : [bh][kt ][s][lane = 32h + r][8]   holds K [32kt + pi(r)][16s + 8h + j]     pi = swap(bit2,bit3)
//   Vt: [bh][kt ][dt][t][lane = 32h + r][8] holds V^T[32dt + r][32kt + 16t + 8h + j]
//
// PQ / PK (query_proj / key_proj of the relative-position table, per head) use the Q / K layouts
// with "sequence" = table row; rows gathered by delta stay coalesced because consecutive deltas sit
// in consecutive 16-byte units.
#pragma once

__host__ __device__ __forceinline__ int glc_pi32(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

// element offset of Q[row][e] (row within this bh's sequence)
__host__ __device__ __forceinline__ size_t glc_qoff(int Sp, int bh, int row, int e) {
    const int nt = Sp >> 5;
    return ((((size_t)bh * nt + (row >> 5)) * 4 + (e >> 4)) * 64 + 32 * ((e >> 3) & 1) + (row & 31)) * 8 + (e & 7);
}
__host__ __device__ __forceinline__ size_t glc_koff(int Sp, int bh, int row, int e) {
    const int nt = Sp >> 5;
    return ((((size_t)bh * nt + (row >> 5)) * 4 + (e >> 4)) * 64 + 32 * ((e >> 3) & 1) + glc_pi32(row & 31)) * 8 + (e & 7);
}
// element offset of V^T[dd][key]
__host__ __device__ __forceinline__ size_t glc_voff(int Sp, int bh, int dd, int key) {
    const int nt = Sp >> 5, ko = key & 31;
    return (((((size_t)bh * nt + (key >> 5)) * 2 + (dd >> 5)) * 2 + (ko >> 4)) * 64 + 32 * ((ko >> 3) & 1) + (dd & 31)) * 8 + (ko & 7);
}

// ---- "MX tiles" (round 3): the attention operands of the MX pipeline (attention_mx.hip) ----
// One 32-row x 64-column tile = 8 KiB, the size of its four split-f16 units, holding what a product a*b = a_hi*b_hi (f16 MFMAs) +
// (a_hi*b_lo + a_lo*b_hi) (ONE block-scaled fp8 MFMA per 32 columns) reads:
//   bytes [0, 4096)     four f16 units (the 16-bit layouts above): unit s, lane 32h + r -> hi halves of columns 16 s + 8 h + j
//   bytes [4096, 8192)  two MX steps of 2 KiB: step m, lane 32h + r -> 32 bytes = [first 16 | second 16] fp8 parts of columns
//                       32 m + 16 h + y, y = 0..15, (hi8 | lo8) or (lo8 | hi8) by tensor: Q, PQ and P travel as (hi8 | lo8), K, PK and
//                       V^T as (lo8 | hi8) — every product pairs a (lo8 | hi8) operand with a (hi8 | lo8) one, block by block.
// Q / K / PQ / PK: tile = 32 consecutive rows (K, PK: row slots permuted by pi) x the head's 64 columns.  V^T: per 32-key tile two
// sub-tiles of 4 KiB (dd 0-31, dd 32-63): [f16 unit t = 0 | t = 1 | one MX step over the 32 keys], lane 32h + dd; MX byte y of lane
// (dd, h) <-> key 16 (y >> 3) + 8 h + (y & 7), the order in which a lane of the S^T accumulator holds its 16 keys.
// lo8 = e4m3((x - f16(x)) * 2^GLC_GX_SHIFT), hi8 = e4m3(x), exponent 0, saturating (glc_common.h).
constexpr int GLC_MXT_BYTES = 8192;
// byte offsets inside this (batch, head)'s tensor of the 8 consecutive columns e0 .. e0 + 7 (e0 % 8 == 0) of Q-layout row `row`
// (slot = row & 31, or pi of it for the K layout): the f16 unit piece (16 B) and the MX piece (8 B at +0: first part, +16: second part)
__host__ __device__ __forceinline__ size_t glc_mxt_f16(int tile, int slot, int e0) {
    return (size_t)tile * GLC_MXT_BYTES + (e0 >> 4) * 1024 + (32 * ((e0 >> 3) & 1) + slot) * 16;
}
__host__ __device__ __forceinline__ size_t glc_mxt_mx(int tile, int slot, int e0) {
    return (size_t)tile * GLC_MXT_BYTES + 4096 + (e0 >> 5) * 2048 + (32 * ((e0 >> 4) & 1) + slot) * 32 + 8 * ((e0 >> 3) & 1);
}
// planar form of a position table's MX steps (round 6; attention_mx.hip DIET): step m as two 1-KiB planes [64 lanes x 16 B first | 64 lanes x 16 B
// second] — the piece of lane 32 h + slot at + lane * 16, so the f16 unit piece (glc_mxt_f16) and all four MX pieces of a row share one per-lane offset
__host__ __device__ __forceinline__ size_t glc_mxt_mx_planar(int tile, int slot, int e0) {
    return (size_t)tile * GLC_MXT_BYTES + 4096 + (e0 >> 5) * 2048 + (32 * ((e0 >> 4) & 1) + slot) * 16 + 8 * ((e0 >> 3) & 1);
}
// V^T: the 8 consecutive keys k0 .. k0 + 7 (k0 % 8 == 0, inside key tile `tile`) of row dd
__host__ __device__ __forceinline__ size_t glc_mxt_v_f16(int tile, int dd, int k0) {
    const int kg = (k0 & 31) >> 3;
    return (size_t)tile * GLC_MXT_BYTES + (dd >> 5) * 4096 + (kg >> 1) * 1024 + (32 * (kg & 1) + (dd & 31)) * 16;
}
__host__ __device__ __forceinline__ size_t glc_mxt_v_mx(int tile, int dd, int k0) {
    const int kg = (k0 & 31) >> 3;
    return (size_t)tile * GLC_MXT_BYTES + (dd >> 5) * 4096 + 2048 + (32 * (kg & 1) + (dd & 31)) * 32 + 8 * (kg >> 1);
}
