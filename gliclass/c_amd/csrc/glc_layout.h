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
