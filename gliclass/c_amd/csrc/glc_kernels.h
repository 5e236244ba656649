// Internal launcher interface between engine.hip and the kernel translation units.
// Every launcher validates the shapes its kernel assumes and returns an error string
// (nullptr = launched) instead of launching on a bad shape.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { GLC_DT_F32 = 0, GLC_DT_BF16 = 1, GLC_DT_F16 = 2 };  // == GLC_F32/BF16/F16 of gliclass_hip.h
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_QKV = 3,
       EPI_SWIGLU = 4,     // gemm256s / gemm256x: W rows interleave 16 gate / 16 up features; C [Mpad, N/2] = silu(gate) * up
       EPI_QKVR = 5 };     // gemm256x only, decoder backbone: RoPE + softmax scale + the MX tiles of decoder_mx.hip written by the epilogue (rope_cs ..)

// Developer A/B switches (GLC_* environment variables: kernel variants kept for same-box comparisons, docs/LOG_r01-r05.md §7) are read only by a
// library built with -DGLC_DEVELOPER (make DEV=1); the product library reads the documented GLICLASS_* knobs and nothing else.
#include <stdlib.h>
inline const char* glc_dev_env(const char* name) {
#ifdef GLC_DEVELOPER
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// fp8 range guard (glc_common.h gx_range_note): the device counter the launchers of this host thread hand to every kernel that writes
// activation operand images (GX rows, MX tiles).  Set by the engine around its launch sequence (under its lock); null = no counting.
inline unsigned*& glc_gx_sat_ptr() { static thread_local unsigned* p = nullptr; return p; }
// ... and the exponent of the activation GX rows (hi8 = e4m3(x 2^sc), glc_common.h) those launchers write and read: 0 unless the engine has lowered
// it after a forward left the e4m3 range (engine.hip act_sc).  Both are set for the duration of one engine forward and reset behind it.
inline int& glc_gx_act_sc() { static thread_local int sc = 0; return sc; }

struct GemmArgs {
    const void* A = nullptr;      // [Mpad, K]  T
    const void* W = nullptr;      // [N, K]     T
    const float* bias = nullptr;  // [N] or null
    void* C = nullptr;            // [Mpad, N]  T   (BIAS / GELU / RESID)
    const void* resid = nullptr;  // [Mpad, N]  T   (RESID)
    void* Qh = nullptr;           // B*nh*Sp*64 T, fragment-major (glc_layout.h)  (QKV)
    void* Kh = nullptr;           // idem, rows permuted by pi inside each 32-key tile
    void* Vt = nullptr;           // V transposed, fragment-major
    int Mpad = 0, N = 0, K = 0;
    int Mvalid = 0, Sp = 0, nh = 0, H = 0;  // QKV only
    const void* W2 = nullptr; const float* bias2 = nullptr; int m_split = 0;   // gemm_nt (128-tile): rows >= m_split use W2 / bias2 (two-group GEMM)
    unsigned long long* stamps = nullptr;   // gemm256s / gemm256x diagnostic builds only: per (block < 64, wave) cycle sums of the main-loop phases
    int qkv_skip_q = 0;                     // QKV: produce only K and V^T (pruned last layer)
    int w_presplit = 0;                     // gemm_nt split-f16 path (T = float): W already holds [32 hi halves | 32 lo halves] per 32-k group (glc_launch_presplit)
    int qkv_split = 0;                      // QKV, T = float: write Q / K / V^T units as [8 hi halves | 8 lo halves] (split-f16 attention, glc_common.h f16x8s)
    // gemm_nt (128-tile) split-K for small M: the K loop is cut into `ksplit` parts (grid.z), each writes its fp32 partial tile to
    // ws[z][Mpad][N]; a second pass sums the parts in a fixed order and applies the epilogue.  ws_bytes = capacity of ws.
    float* ws = nullptr; size_t ws_bytes = 0;
    // gemm256s on group-split operands (glc_launch_gemm256s_gs): EPI_BIAS / EPI_GELU write C as plain fp32 rows instead of GS rows;
    // EPI_RESID reads its residual as plain fp32 rows instead of GS rows (decoder backbone: the residual stream itself)
    int gs_c_plain = 0, gs_resid_plain = 0;
    int n_group = 0;                        // gemm256s, set by its launcher: tile order sweeps the M-tiles once per group of n_group N-tiles (0: row-major)
    // QKV (gemm256s), pruned last layer: one byte per 32-row tile of the [Mpad] rows; a workgroup of the Q third whose 256 rows hold no
    // flagged tile returns at once (only the query tiles with selected rows are ever read)
    const unsigned char* q_tile_flag = nullptr;
    // LayerNorm folded into the group-split GEMMs around it (glc_launch_gemm256s_gs; docs/LOG_r01-r05.md "LayerNorm folded away").  The producer
    // (EPI_RESID) writes the RAW sum (GS rows, C) plus per-row partial (sum, squared deviations from the block mean) of each 64-column block to ln_part
    // [Mpad][N / 64]; glc_launch_ln_stats turns them into (mean, rstd) per row.  A consumer whose A rows are such raw rows gets
    // a_stats [Mpad] + ln_c [N]: W then holds W . diag(gamma), ln_c[n] = sum_k W'[n][k], bias[n] = sum_k beta[k] W[n][k] + b[n], and the
    // epilogue forms rstd_m (acc - mean_m ln_c[n]) + bias[n] = (LN(x) W^T + b)[m][n].  EPI_RESID with r_stats normalises its raw
    // residual row on the fly: (r - mean_m) rstd_m r_gamma[n] + r_beta[n].
    float2* ln_part = nullptr;
    const float2* a_stats = nullptr; const float* ln_c = nullptr;
    const float2* r_stats = nullptr; const float* r_gamma = nullptr; const float* r_beta = nullptr;
    // precision-budget switches (glc_debug_set_precision_mask; GS kernels): bit 0 = A rounded to f16 (lo halves dropped), bit 1 = W,
    // bit 2 = the GS residual rows
    int prec = 0;
    int prio_mode = -1;                     // gemm256x: wave priority policy of the main loop (-1: default / GLC_GEMM_PRIO; developer A/B)
    int qkv_mxt = 0;                        // gemm256x, EPI_QKV: write Q / K / V^T as MX tiles (glc_layout.h) for attention_mx.hip instead of split-f16 units
    int mx_ws = 0;                          // gemm256x: exponent of W's fp8 parts (GX rows written with glc_launch_to_gx(.., mx_ws))
    unsigned* gx_sat = nullptr;             // gemm256x: fp8 range guard counter (filled by the launcher from glc_gx_sat_ptr())
    int z16 = 0;                            // gemm256x: the main loop on v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4 (same GX images; K % 64 == 0)
    int gy = 0;                             // gemm256x: A, W, resid and C (where they are operand images) are GY rows — e2m3 parts with block scales (glc_common.h) — not GX rows
    int act_sc = 0;                         // gemm256x: exponent of the ACTIVATION GX rows it reads (A, resid) and writes (C); 0 unless the engine lowered it (engine.hip act_sc)
    int gx_rows = 0;                        // ... counted over rows [0, gx_rows) only: the slack rows up to Mpad hold leftovers of other forwards (0: Mvalid, else Mpad)
    // gemm256x, EPI_QKVR (decoder backbone, head_dim 128, nq and nkv even): N = (nq + 2 nkv) 128 fused projection columns; W rows (and bias)
    // of every Q / K head in the order glc_rope_perm128 gives (the two members of a rotate-half pair in one wave's accumulators); the
    // epilogue applies RoPE (rope_cs [Sp][64] (cos, sin)) and qscale (Q) in fp32 and writes Qh / Kh / Vt as the MX tiles of decoder_mx.hip.
    const float* rope_cs = nullptr; float qscale = 1.f; int nq = 0, nkv = 0;
    int epi_abl = 0;                        // developer timing ablations of the GX-row epilogue (glc_debug_gemm_bench): 1 = no stores, 2 = every tile stores into rows [0, 256)
    int perm_cols = 0;                      // gemm256x, EPI_BIAS with gs_c_plain: columns [0, perm_cols) arrive in that order and are stored at their logical place
};
// EPI_QKVR: physical row p (0 .. 127) of a Q / K head of the fused projection weight holds logical feature glc_rope_perm128(p): the 32-blocks
// 1 and 2 trade places, so that a wave's 64 columns are features [32 h, 32 h + 32) and their rotate-half partners [64 + 32 h, 64 + 32 h + 32)
__host__ __device__ inline int glc_rope_perm128(int p) { const int b = (p >> 5) & 3; return (p & 31) | ((((b & 1) << 1) | (b >> 1)) << 5); }
// Precision-budget mask of an engine (developer, gliclass_hip.h glc_debug_set_precision_mask): a set bit rounds that operand of the
// default mode's group-split pipeline to f16 by dropping its lo halves (numerically identical to the cheaper kernel that never
// fetches them).  GEMM classes x (A, W); attention operand tensors; the GS residual reads of the two residual GEMMs.
enum { PM_QKV_A = 1 << 0, PM_QKV_W = 1 << 1, PM_AO_A = 1 << 2, PM_AO_W = 1 << 3, PM_F1_A = 1 << 4, PM_F1_W = 1 << 5, PM_F2_A = 1 << 6, PM_F2_W = 1 << 7,
       PM_Q = 1 << 8, PM_K = 1 << 9, PM_V = 1 << 10, PM_P = 1 << 11, PM_PQ = 1 << 12, PM_PK = 1 << 13, PM_RESID = 1 << 14 };
// (sum, M2) partials of 64-column blocks [M][nparts] -> (mean, 1 / sqrt(var + eps)) [M] over rows of H = 64 nparts values (rows.hip; Chan's merge in double, fixed order)
const char* glc_launch_ln_stats(hipStream_t st, const float2* part, int nparts, float2* stats, int M, int H, float eps, int rms = 0);   // rms: (0, 1 / sqrt(E[x^2] + eps))
// decoder backbone: plain fp32 rows -> raw group-split rows + RMSNorm statistics (0, rstd) per row
const char* glc_launch_rows_to_gs_rms(hipStream_t st, const float* X, void* Y, float2* stats, float eps, int M, int H, int gx = 0);
const char* glc_launch_gemm(hipStream_t st, int dtype, int epi, const GemmArgs& a);       // 128x128 tile, any T
bool glc_gemm256_supported(int dtype, const GemmArgs& a);                                 // shapes the 256x256 LDS-DMA kernel takes (16-bit T)
const char* glc_launch_gemm256s(hipStream_t st, int dtype, int epi, const GemmArgs& a);   // 256x256 tile, staggered wave groups (gemm256s.hip)
void glc_gemm_set_full_lines(int on);       // gemm256s.hip: full-line (operand-major) ring stages on / off, process-wide (developer A/B; default on, GLC_GEMM_FL=0)
bool glc_gemm_small_m(const GemmArgs& a);   // gemm256s.hip: too few 256x256 tiles for this device -> use the 128x128 kernel
// picks the 256x256 LDS-DMA kernel when the shape allows it, else the 128x128 one
inline const char* glc_launch_gemm_auto(hipStream_t st, int dtype, int epi, const GemmArgs& a) {
    const bool qkv_ok = epi != EPI_QKV || a.H % 256 == 0;
    if (!(glc_gemm256_supported(dtype, a) && qkv_ok)) return glc_launch_gemm(st, dtype, epi, a);
    // Small M (the reference's own batches of 8 short texts): too few 256x256 tiles to cover the CUs, and each tile runs its
    // whole K loop alone — the 128x128 kernel gives 4x the workgroups.  Threshold: fewer 256-tiles than half the CUs.
    if (glc_gemm_small_m(a)) return glc_launch_gemm(st, dtype, epi, a);
    return glc_launch_gemm256s(st, dtype, epi, a);
}

// Row LayerNorm: Y[m,:] = LN(X[m,:]) * gamma + beta, rows [0, M).  X, Y element type T.
const char* glc_launch_layernorm(hipStream_t st, int dtype, const void* X, void* Y, const float* gamma,
                                 const float* beta, float eps, int M, int H);

// ---- group-split ("GS") activations of the fp32 mode (rows.hip): a row of K fp32 values kept in the same 4 K bytes as K / 32 groups
// of [32 hi halves | 32 lo halves] (x = hi + lo), the operand image of the split-f16 GEMMs ----
// (gx != 0: the rows are written / read in the GX format of the MX cross-term GEMM — same bytes per group, glc_common.h)
const char* glc_launch_layernorm_gs(hipStream_t st, const float* X, void* Y, const float* gamma, const float* beta, float eps, int M, int H, int gx = 0);
const char* glc_launch_embed_gs(hipStream_t st, const int64_t* ids, const int64_t* mask, const float* table, const float* gamma,
                                const float* beta, float eps, void* X, float* kbias, int B, int S, int Sp, int H, int vocab, int pad_id, int gx = 0);
const char* glc_launch_gather_rows_gs(hipStream_t st, const void* X, const int* cls_pos, int c_cap, float* Xs, int* sel_b, int* sel_q,
                                      unsigned char* tile_flag, int B, int Sp, int H, int C, int gx = 0);
// 256x256 LDS-DMA GEMM on GS operands (gemm256s.hip): A [Mpad, K] and W [N, K] in the GS format; three f16 MFMAs per product
// (a_lo*w_hi + a_hi*w_lo + a_hi*w_hi) on the 16-bit kernel's ring, every 64-byte part fetched once.  EPI_GELU / EPI_BIAS: C in the GS format;
// EPI_RESID: resid in the GS format, C plain fp32 (the LayerNorm input); EPI_QKV: Q / K / V^T as split-f16 units (qkv_split).
bool glc_gemm256s_gs_supported(const GemmArgs& a, int epi);
// MX cross-term GEMM on GX rows (gemm256x.hip; glc_common.h has the format): A [Mpad, K] and W [N, K] as GX rows (W with fp8 exponent
// mx_ws), one f16 MFMA pair + one block-scaled fp8 MFMA per product.  EPI_GELU / EPI_BIAS: C as GX rows (gs_c_plain: plain fp32);
// EPI_RESID: resid as GX rows, C raw GX rows + ln_part, or plain fp32; EPI_QKV: split-f16 units; LayerNorm fold arguments as gemm256s.
bool glc_gemm256x_supported(const GemmArgs& a, int epi);
const char* glc_launch_gemm256x(hipStream_t st, int epi, const GemmArgs& a);
const char* glc_launch_to_gx(hipStream_t st, void* w, size_t n, int sc, int worder);       // in place: n fp32 values -> GX rows, fp8 exponent sc; worder: weight rows
// the MX weight copies from the split-f16 (group-split) copies already on the device: largest magnitude (float bits, atomicMax into *d_bits), then the conversion
const char* glc_launch_gs_absmax(hipStream_t st, const void* gs, size_t n, unsigned* d_bits);
const char* glc_launch_gs_to_gx(hipStream_t st, const void* gs, void* gx, size_t n, int sc);
// GY rows (glc_common.h: e2m3 parts with block scales, gy_row_bytes(K) per row): plain fp32 rows -> GY (A order / worder != 0: W order), the
// projection weights from their split-f16 copies (W order), and back to fp32 (A order: x = hi + lo)
#ifdef GLC_DEVELOPER      // GY rows (e2m3 cross terms with block scales: csrc/dev/gemm256x_dev.hip) — developer builds only
const char* glc_launch_to_gy(hipStream_t st, const float* src, void* dst, size_t rows, int K, int worder);
const char* glc_launch_gs_to_gy(hipStream_t st, const void* gs, void* dst, size_t rows, int K);
#else
inline const char* glc_launch_to_gy(hipStream_t, const float*, void*, size_t, int, int) { return "GY rows exist in developer builds only (make DEV=1)"; }
inline const char* glc_launch_gs_to_gy(hipStream_t, const void*, void*, size_t, int) { return "GY rows exist in developer builds only (make DEV=1)"; }
#endif
#ifdef GLC_DEVELOPER
const char* glc_launch_gy_to_f32(hipStream_t st, const void* src, float* dst, size_t rows, int K);
#else
inline const char* glc_launch_gy_to_f32(hipStream_t, const void*, float*, size_t, int) { return "GY rows exist in developer builds only (make DEV=1)"; }
#endif
#ifndef GLC_GX_SHIFT
#define GLC_GX_SHIFT 11                     // GX rows: lo8 = e4m3((x - hi) * 2^(GLC_GX_SHIFT + sc)) (glc_common.h)
#endif
// fp8 exponent for a weight tensor whose largest magnitude is maxabs: 2^sc maxabs <= 240 (e4m3 saturates at 448)
inline int glc_gx_weight_exponent(float maxabs) {
    if (!(maxabs > 0.f)) return 0;
    int sc = (int)floorf(log2f(240.0f / maxabs));
    return sc < -30 ? -30 : (sc > 40 ? 40 : sc);
}
const char* glc_launch_gemm256s_gs(hipStream_t st, int epi, const GemmArgs& a);

// Embedding gather + LayerNorm + mask (modeling_deberta_v2.py:533,550,552-559) on the padded
// [B, Sp] grid (positions s >= S behave as padding); also writes the additive key bias
// kbias[b*Sp+s] = mask ? 0 : -1e30.
const char* glc_launch_embed(hipStream_t st, int dtype, const int64_t* ids, const int64_t* mask, const void* table,
                             const float* gamma, const float* beta, float eps, void* X, float* kbias,
                             int B, int S, int Sp, int H, int vocab, int pad_id);

// Per batch row: klen[b] = 1 + last valid key (0 if none), kfirst[b] = first masked key (S if none); ordered positions of class tokens
// cls_pos[b*c_cap + j] (-1 beyond the row's count) and cls_cnt[b].
const char* glc_launch_scan_rows(hipStream_t st, const int64_t* ids, const int64_t* mask, int B, int S,
                                 int class_token, int embed_class_token, int* klen, int* kfirst, int* cls_pos, int* cls_cnt, int c_cap);

struct AttnArgs {
    const void* Qh; const void* Kh; const void* Vt;   // as written by EPI_QKV (Q pre-scaled by 1/sqrt(3d))
    const void* PK; const void* PQ;                   // nh*P*64 T: key_proj(rel) (K layout), query_proj(rel)/sqrt(3d) (Q layout)
    const int32_t* dtab;                              // [2*Sp-1] clamp(bucket(q-k)+span)
    const float* kbias;                               // [B, Sp]
    const int* klen;                                  // [B] 1 + last valid key
    const int* kfirst;                                // [B] first masked key (S if none)
    void* CTX;                                        // [B*Sp, H] T  (or [nsel, H] with a row selection)
    int B, nh, Sp, H, P;
    // optional row selection (simple kernel only): query r is row sel_q[r] of sequence sel_b[r]; its projected
    // query is row r of the row-major Qrow [nsel, H]; output row r of CTX
    // band kernel: q-k >= rsat_pos => delta == P-1 ; q-k <= rsat_neg => delta == 0 (table saturation, host-computed;
    // Sp / -Sp when the table does not saturate)
    int rsat_pos = 1 << 30, rsat_neg = -(1 << 30);
    const unsigned char* tile_flag = nullptr;         // band kernel: [B, Sp/32] — process only query tiles whose flag is set
    const int* sel_b = nullptr; const int* sel_q = nullptr; const void* Qrow = nullptr; int nsel = 0;
    // band kernel, diagnostic build only: per (block < 64, wave) cycle sums of the band-tile segments [8] (s_memtime ticks)
    const int2* otab = nullptr;                       // band kernel: [2*Sp-1+128] byte offsets of row delta(q-k) in PQ (x) / PK (y), 64 clamped entries each side
    const int2* mtab = nullptr;                       // MX band kernel (round 6): [2 lane halves][2 Sp + 512] ready-made row offsets into the planar copies of PQ (.x) / PK (.y, the entries backwards) that follow the tile images at PQ / PK + nh * P * 256 bytes (engine.hip)
    unsigned long long* stamps = nullptr;
    int variant = 0;                                  // band kernel diagnostics: bit 1 = one wave per SIMD (LDS padding)
    int split = 0;                                    // band kernel, fp32 mode: operands are split-f16 units (GemmArgs::qkv_split), three f16 MFMAs per product
    int ctx_gs = 0;                                   // workgroup-shared kernel, split operands: write CTX rows in the GS format (1) or the GX format (2)
    int ksplit = 0;                                   // per-wave band kernel with tile_flag: a workgroup with ONE flagged query tile splits that tile's keys over its 4 waves
    int prec = 0;                                     // workgroup-shared kernel, split units: (engine mask >> 8) & 63 — bits Q, K, V, P, PQ, PK rounded to f16
    const void* idx16 = nullptr; const int4* tinfo = nullptr;   // attention_mx2.hip: the tables of glc_mx2_build_tables for this Sp
    unsigned* gx_sat = nullptr;                       // GX context rows: fp8 range guard counter (filled by the launchers from glc_gx_sat_ptr())
    int act_sc = 0;                                   // GX context rows: exponent of the activation images (engine.hip act_sc)
};
// impl: 1 = simple (any T), 2 = MFMA band kernel, one independent wave per 32-query tile (attention.hip)
const char* glc_launch_attention(hipStream_t st, int dtype, int impl, const AttnArgs& a);
// impl 3: workgroup-shared band kernel (attention_wg.hip): K / V^T tiles through an LDS-DMA ring, p2c band shared by the waves of a
// workgroup.  16-bit operands, or the fp32 mode's split-f16 units (a.split).
const char* glc_launch_attention_wg(hipStream_t st, int dtype, const AttnArgs& a);

// Workgroup-shared band kernel on MX tiles (attention_mx.hip; the attention of the MX pipeline): Qh / Kh / Vt / PQ / PK are MX tiles
// (glc_layout.h), CTX is written as GX rows; otab is the split-unit offset table.
const char* glc_launch_attention_mx(hipStream_t st, const AttnArgs& a);
// (Round 5's role-split kernel — a matrix wave and a softmax wave per SIMD, 1.32-1.36 ms against the band kernel's 1.12-1.17 — was deleted in round 6:
//  docs/LOG_r05.md 3g, git history.)
// Round 4: the same operands and outputs, position terms in bucket (delta) space, one independent wave per query tile (csrc/dev/attention_mx2.hip:
// measured 4-6 % slower than the band kernel — a DEVELOPER kernel since round 5, not in the product library).
// glc_mx2_build_tables: the kernel's two tables from the distance -> delta table of a padded length; false = this table does not have the
// structure the kernel needs (the caller keeps glc_launch_attention_mx).
#include <vector>
#ifdef GLC_DEVELOPER
const char* glc_launch_attention_mxd(hipStream_t st, const AttnArgs& a);      // csrc/dev/attention_mxd.hip: two query tiles per wave, one wave per SIMD (round 5)
const char* glc_launch_attention_mx2(hipStream_t st, const AttnArgs& a);
bool glc_mx2_build_tables(int Sp, int P, const int32_t* dtab, std::vector<unsigned char>& idx16, std::vector<int4>& tinfo);
#else
inline const char* glc_launch_attention_mxd(hipStream_t, const AttnArgs&) { return "attention(mxd): the two-tiles-per-wave kernel exists in developer builds only (make DEV=1)"; }
inline const char* glc_launch_attention_mx2(hipStream_t, const AttnArgs&) { return "attention(mx2): the bucket-space kernel exists in developer builds only (make DEV=1)"; }
inline bool glc_mx2_build_tables(int, int, const int32_t*, std::vector<unsigned char>&, std::vector<int4>&) { return false; }
#endif
// position tables at load: split-f16 units (Q / K layout, ntiles tiles of 32 rows x 64 columns) -> MX tiles; hl: (hi8 | lo8) order (PQ), else (lo8 | hi8) (PK)
const char* glc_launch_units_to_mxt(hipStream_t st, const void* src, void* dst, int ntiles, int hl, unsigned* sat = nullptr, int planar = 0);   // sat: fp8 range guard counter (glc_common.h)

#include <atomic>
// CU count of the CURRENT device, cached per device ordinal (a session may span GPUs of different sizes)
inline int glc_device_cus() {
    static std::atomic<int> cache[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
// Kernels that need more than 64 KiB of dynamic LDS raise the per-function limit once per device.
template <typename F> inline bool glc_raise_lds_limit(F* kernel, int bytes, std::atomic<unsigned>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) return false;
    if (done.load(std::memory_order_acquire) >> dev & 1u) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.fetch_or(1u << dev, std::memory_order_release);
    return true;
}

// Head: gather pooled rows -> Gt [B,H] and class-token rows -> Gc [B*C,H], both fp32.  Pooled row = position 0, or the last
// attended token (klen[b]-1) when klen is given.
const char* glc_launch_head_gather(hipStream_t st, int dtype, const void* X, const int* cls_pos, int c_cap,
                                   float* Gt, float* Gc, int B, int Sp, int H, int C, const int* klen = nullptr);
// pooling = 'avg': Gt[b,:] = mean over the attended positions (kbias == 0) of X[b,s,:]
const char* glc_launch_pool_avg(hipStream_t st, int dtype, const void* X, const float* kbias, float* Gt, int B, int Sp, int H);
// logits[b*C+j] = <Tt[b], Cc[b*C+j]> (* logit_scale when normalised)
const char* glc_launch_head_score(hipStream_t st, const float* Tt, const float* Cc, float* logits, int B, int C, int H,
                                  int normalize, float logit_scale);
// scorers 'weighted-dot' / 'mlp' (include/gliclass_hip.h): row shuffles between the head's GEMMs and the closing ReLU -> Linear(K, 1)
const char* glc_launch_scorer_wd_cat(hipStream_t st, const float* St, const float* Sc, float* cat, int B, int C, int H);
const char* glc_launch_scorer_pair(hipStream_t st, const float* Tt, const float* Cc, float* out, int B, int C, int H);
const char* glc_launch_relu(hipStream_t st, float* x, size_t n);
const char* glc_launch_relu_dot(hipStream_t st, const float* X, const float* w, const float* bias, float* logits, int rows, int K, float scale = 1.0f);
// normalize_features ahead of those scorers: X[r][:] /= (|X[r]| + 1e-8), rows x H fp32 in place
const char* glc_launch_l2norm_rows(hipStream_t st, float* X, int rows, int H);

// Pruned last layer: compact the rows the head reads. Row r < B: [CLS] of sequence r; row B + b*C + j: class
// token j of sequence b (sequence start if absent). Writes Xs[r,:] = X[row,:] and the (sequence, position) lists.
const char* glc_launch_gather_rows(hipStream_t st, int dtype, const void* X, const int* cls_pos, int c_cap, void* Xs,
                                   int* sel_b, int* sel_q, unsigned char* tile_flag, int B, int Sp, int H, int C);
// dst[r,:] = src[(sel_b[r]*Sp + sel_q[r]),:] for r < R
const char* glc_launch_gather_sel(hipStream_t st, int dtype, const void* src, const int* sel_b, const int* sel_q, void* dst, int R, int Sp, int H);
// Head gather from the compact rows: Gt[b] = Xs[b]; Gc[b*C+j] = Xs[B+b*C+j] or 0 when the class token is absent.
const char* glc_launch_head_gather_sel(hipStream_t st, int dtype, const void* Xs, const int* cls_pos, int c_cap,
                                       float* Gt, float* Gc, int B, int H, int C);

// ---- decoder-style backbone (decoder.hip) ----
// X[b*Sp + s, :] = table[ids[b,s]] (pad rows use pad_id); kbias = 0 / -1e30 from the mask
const char* glc_launch_embed_plain(hipStream_t st, int dtype, const int64_t* ids, const int64_t* mask, const void* table, void* X,
                                   float* kbias, int B, int S, int Sp, int H, int vocab, int pad_id);
// Y = w * X * rsqrt(mean(X^2) + eps) row-wise
// (glc_launch_rmsnorm_gs: fp32 X in, group-split Y out; glc_launch_swiglu_gs: plain fp32 [gate | up] rows in, group-split F out)
const char* glc_launch_rmsnorm_gs(hipStream_t st, const float* X, void* Y, const float* w, float eps, int M, int H);
const char* glc_launch_swiglu_gs(hipStream_t st, const float* GU, void* F, size_t M, int I);
const char* glc_launch_rmsnorm(hipStream_t st, int dtype, const void* X, void* Y, const float* w, float eps, int M, int H);
// in-place rotate-half RoPE on the Q and K heads of QKV [M, (nq+2nkv) d]; cs = [Sp][d/2][cos,sin]; Q additionally * qscale
const char* glc_launch_rope_qk(hipStream_t st, int dtype, void* QKV, const float* cs, int M, int Sp, int nq, int nkv, int d, float qscale);
// F[m,i] = silu(GU[m,i]) * GU[m,I+i]
const char* glc_launch_swiglu(hipStream_t st, int dtype, const void* GU, void* F, size_t M, int I, int inter = 0);   // inter: 16 gate / 16 up interleaved columns
// grouped-query attention on the row-major fused QKV (Q pre-scaled by log2e/sqrt(d)); CTX [B*Sp, nq*d]; impl 1 = straightforward
const char* glc_launch_attention_gqa(hipStream_t st, int dtype, int impl, const void* QKV, const float* kbias, const int* klen, void* CTX,
                                     int B, int Sp, int nq, int nkv, int d, int causal);
// 16-bit MFMA path: RoPE + scale + fragment-major Q / K / V^T (layouts in decoder.hip), then the flash-style kernel
const char* glc_launch_qkv_layout(hipStream_t st, int dtype, const void* QKV, const float* cs, void* Qf, void* Kf, void* Vt, int B, int Sp,
                                  int nq, int nkv, int d, float qscale);
// ctx_gs (fp32 mode only): write the context rows in the group-split format
const char* glc_launch_attention_gqa_mfma(hipStream_t st, int dtype, const void* Qf, const void* Kf, const void* Vt, const float* kbias,
                                          const int* klen, const int* kfirst, void* CTX, int B, int Sp, int nq, int nkv, int d, int causal, int ctx_gs = 0);

// MX pipeline (decoder_mx.hip, round 4): the fused fp32 projection -> RoPE + scale + MX tiles (f16 hi units + fp8 steps, 4 bytes per element),
// and the grouped-query attention on them (a_hi*b_hi in f16 MFMAs + both cross terms in one block-scaled fp8 MFMA); CTX as GX rows
const char* glc_launch_qkv_layout_mx(hipStream_t st, const void* QKV, const float* cs, void* Qm, void* Km, void* Vm, int B, int Sp, int nq, int nkv, int d, float qscale);
const char* glc_launch_attention_gqa_mx(hipStream_t st, const void* Qm, const void* Km, const void* Vm, const float* kbias, const int* klen, const int* kfirst, void* CTX,
                                        int B, int Sp, int nq, int nkv, int d, int causal);

// dtype conversion fp32 -> T (weights upload), n elements
const char* glc_launch_convert(hipStream_t st, int dtype, const float* src, void* dst, size_t n);
// In place: every group of 32 consecutive floats becomes [32 hi halves | 32 lo halves] (x = hi + lo), the LDS row image of the
// split-f16 GEMM (gemm.hip) — weights are split once at load instead of in every tile.  n % 32 == 0.
const char* glc_launch_presplit(hipStream_t st, void* w, size_t n);
// T [rows, H] -> fp32 (debug dumps)
const char* glc_launch_to_f32(hipStream_t st, int dtype, const void* src, float* dst, size_t n);
