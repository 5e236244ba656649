// Engine = what g_ort->CreateSession + g_ort->Run are to /root/reference/src/model.c:173,269:
// weight residency in HBM, workspace management, and the per-forward launch sequence on ONE HIP
// stream.  Implements include/gliclass_hip.h.  No CPU fallback exists: without a GPU every entry
// point fails loudly through glc_last_error().
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <algorithm>
#include <string>
#include <vector>

#include "../../../include/gliclass_hip.h"
#include "glc_kernels.h"
#include "glc_layout.h"

namespace {

thread_local std::string g_err;
void set_err(const std::string& s) { g_err = s; }

#define HIPCHK(expr, ret)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            set_err(std::string(#expr) + ": " + hipGetErrorString(_e));                            \
            return ret;                                                                            \
        }                                                                                          \
    } while (0)
#define KCHK(expr, ret)                                                                            \
    do {                                                                                           \
        const char* _m = (expr);                                                                   \
        if (_m) { set_err(_m); return ret; }                                                       \
    } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t esize(int dtype) { return dtype == GLC_F32 ? 4 : 2; }

enum { PC_SCAN = 0, PC_EMBED, PC_QKV, PC_ATTN, PC_ATTN_OUT, PC_LN, PC_FFN1, PC_FFN2, PC_HEAD, PC_LAST, PC_N };
const char* const kProfNames[PC_N] = {"scan_rows", "embed_ln", "gemm_qkv", "attention", "gemm_attn_out", "layernorm",
                                      "gemm_ffn1_gelu", "gemm_ffn2", "head", "last_layer_pruned"};

struct DecLayerW {                     // decoder-style backbone (decoder.hip)
    void *Wqkv = nullptr, *Wo = nullptr, *Wgu = nullptr, *Wd = nullptr;       // T: [(nq+2nkv)d, H], [H, nq d], [2I, H] (gate rows | up rows), [H, I]
    float *bqkv = nullptr, *ln1 = nullptr, *ln2 = nullptr;                    // f32
    void *Wqkvf = nullptr, *Wguf = nullptr;                                   // fp32 mode, RMSNorm folded into the GEMMs: Wqkv diag(ln1), Wgu diag(ln2), group-split
    void *Wqkvf_x = nullptr, *Wo_x = nullptr, *Wguf_x = nullptr, *Wd_x = nullptr;   // MX pipeline: the same four as GX rows + their fp8 exponents
    int ws_qkvf = 0, ws_o = 0, ws_guf = 0, ws_d = 0;
    float* bqkv_p = nullptr;            // bqkv in the row order of a Wqkvf_x built for the RoPE epilogue (glc_rope_perm128), else null
};

struct LayerW {
    void *Wqkv = nullptr, *Wo = nullptr, *W1 = nullptr, *W2 = nullptr;       // T
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;       // f32
    float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr; // f32
    void *PK = nullptr, *PQ = nullptr;                                        // T [nh, P, 64]
    void *PKs = nullptr, *PQs = nullptr;                                      // fp32 mode: the same tables as split-f16 units (band kernel, AttnArgs::split)
    // LayerNorm folded into the group-split GEMMs (GemmArgs::a_stats): W1 . diag(ln1 gamma), Wqkv . diag(previous layer's ln2 gamma)
    // as group-split rows, their row sums c and the folded biases d = W beta + b
    void *W1f = nullptr, *Wqkvf = nullptr;
    float *c1 = nullptr, *d1 = nullptr, *cq = nullptr, *dq = nullptr;
    // MX pipeline (glc_engine::mx): the projection weights once more as GX rows (glc_common.h) with their fp8 exponents
    void *PKm = nullptr, *PQm = nullptr;                                     // the position tables as MX tiles (attention_mx.hip)
    void *Wqkv_x = nullptr, *Wqkvf_x = nullptr, *Wo_x = nullptr, *W1f_x = nullptr, *W2_x = nullptr;
    int ws_qkv = 0, ws_qkvf = 0, ws_o = 0, ws_1f = 0, ws_2 = 0;
};

}  // namespace

struct glc_engine {
    glc_model_config cfg{};
    int dtype = GLC_F32, device = 0, attn_impl = 0;
    bool prune_last = true;         // last layer only on the rows the head reads (exact)
    bool w_presplit = false;        // weights of the split-f16 fp32 GEMMs are split once at load (encoder layers in fp32 mode; head projectors in every mode)
    bool dec_split = false;         // decoder backbone, fp32 mode: RoPE/layout pass writes split-f16 units, grouped-query attention on three-MFMA products
    bool attn_split = false;        // fp32 mode: band attention on split-f16 operands (three f16 MFMAs per product); GLICLASS_F32_ATTN=native turns it off
    bool mx_built = false, mx = false;   // MX cross-term projections (gemm256x.hip) on GX rows: allowed for this engine / pipeline selected (GLICLASS_MX, glc_debug_set_mx)
    bool mx_ready = false;               // ... and the GX copies of the projection weights exist: built from the split-f16 copies by the first forward that takes the pipeline
    size_t mx_bytes = 0;                 // their size (glc_debug_mx_weight_bytes)
    bool last_mx = false;                // the last forward ran the MX pipeline
    bool last_mx_attn = false;           // ... and its attention ran on MX tiles (attention_mx.hip)
    bool dec_rope_epi = true;            // decoder MX pipeline: RoPE + MX tiles as the QKV projection's epilogue (gemm256x EPI_QKVR); GLC_DEC_ROPE_EPI=0: the separate pass
    bool mx_attn = true;                 // MX pipeline: attention on MX tiles (attention_mx.hip); false: split-f16 units (GLC_MX_ATTN=0, glc_debug_set_mx_attention)
    int debug_stop = -1;                 // developer: leave run_forward after stage (10 * layer + k), k = 0 QKV, 1 attention, 2 attn-out, 3 FFN1, 4 FFN2 (+ LayerNorm): workspace inspection
    int prec_mask = 0;              // precision-budget switches (PM_* of glc_kernels.h; glc_debug_set_precision_mask): operands rounded to f16 in the group-split pipeline
    int gs_mode = 1;                // fp32 mode, group-split activations + 256-tile LDS-DMA GEMMs: 0 off, 1 auto (large shapes), 2 whenever the shapes allow (tests)
    bool last_gs = false;           // the last forward ran the group-split pipeline
    bool ln_fused = true;           // group-split pipeline: LayerNorm folded into the GEMMs around it (GLC_LNF=0: separate LayerNorm kernels)
    bool last_lnf = false;          // the last forward ran with LayerNorm / RMSNorm folded into its GEMMs
    float2 *statsA = nullptr, *statsB = nullptr, *ln_part = nullptr;     // (mean, rstd) per row of X / H1 when they hold raw sums; the producers' partials
    int max_buckets = 4;            // host-buffer forward: split a ragged batch into <= this many length groups (1 = off)
    int last_groups = 1;            // groups the last host-buffer forward ran as
    int range_retries = 0;          // host-buffer forwards repeated with the norms unfused because the folded one came out non-finite
    // fp8 range guard of the MX pipeline (glc_common.h gx_range_note): device counter of activation elements beyond the e4m3 range, its value after the
    // last checked forward, a pinned host slot for the device-resident path; forwards repeated on the split-f16 kernels because they counted
    // any; consecutive such forwards (the model has outlier channels: after kFp8Sticky of them the engine leaves the MX pipeline for good)
    unsigned* d_gxsat = nullptr; unsigned gxsat_seen[2] = {0, 0}; unsigned* h_gxsat = nullptr;      // two words: [0] activation rows (GX images, exponent act_sc), [1] Q / K / V MX tiles (exponent 0)
    int fp8_retries = 0, fp8_streak = 0; bool fp8_sticky_off = false, fp8_device_pending = false;
    int device_invalid = 0;                    // a device-resident forward since the last glc_engine_sync left the fp8 range (1: rows only, 2: tiles): its logits are not valid
    // Activation exponent of the MX pipeline's GX rows (hi8 = e4m3(x 2^act_sc), glc_common.h): 0 until a forward leaves the e4m3 range (|x| > 448);
    // the guard's FIRST answer is then kActScLow = -5 for this engine (rows hold |x| up to 14336, elements below 0.5 keep fewer hi8 bits — their
    // cross terms are 2^-16 of a unit product either way) and the forward is repeated on the MX pipeline; only what still leaves the range
    // (or an MX tile of the attention: Q, K, V, P keep exponent 0) goes to the split-f16 kernels.
    int act_sc = 0;
    float* splitk_ws = nullptr; size_t splitk_ws_bytes = 0;     // fp32 partial tiles of the split-K GEMM path (small M)
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::vector<void*> allocs;      // everything freed at destroy
    // weights
    void* emb = nullptr; float *eln_g = nullptr, *eln_b = nullptr;
    std::vector<LayerW> layers;
    std::vector<DecLayerW> dlayers; float* final_norm = nullptr;      // decoder backbone
    std::map<int, float*> ropes;                                      // Sp -> [Sp][d/2][cos,sin]
    void *QKV = nullptr, *GU = nullptr, *X2 = nullptr;                // decoder workspace: fused QKV rows, [gate|up] rows, second residual buffer
    bool fused_swiglu = false;                                        // Wgu rows interleaved 16 gate / 16 up: SwiGLU runs in the GEMM epilogue
    float* headw[8] = {nullptr};
    float* scw[8] = {nullptr};           // the scorer's own tensors (weighted-dot: 8, mlp: 6, simple: none), fp32
    float* scorer_ws = nullptr;          // its row buffers
    int P = 0;
    // workspace
    int capM = 0, capB = 0, capIds = 0, capC = 0, capHeadRows = 0, capSel = 0, capGU = 0;
    void *Xs = nullptr, *Qs = nullptr, *CTXs = nullptr, *T1s = nullptr, *H1s = nullptr, *FFs = nullptr;   // compact rows of the pruned last layer
    int *sel_b = nullptr, *sel_q = nullptr;
    unsigned char* tile_flag = nullptr; size_t capFlag = 0;
    void *X = nullptr, *Qh = nullptr, *Kh = nullptr, *Vt = nullptr, *CTX = nullptr, *T1 = nullptr, *H1 = nullptr, *FF = nullptr;
    float* kbias = nullptr; int *klen = nullptr, *kfirst = nullptr, *cls_pos = nullptr, *cls_cnt = nullptr;
    int64_t *d_ids = nullptr, *d_mask = nullptr;
    float *Gt = nullptr, *G1t = nullptr, *G2t = nullptr, *d_logits = nullptr;   // head rows: [text | class] groups, 128-aligned
    std::map<int, int32_t*> dtabs;
    std::map<int, int2*> mtabs;                // Sp -> the MX band kernel's ready-made row offsets (round 6; fp32 mode only): glc_kernels.h AttnArgs::mtab
    std::map<int, int2*> otabs;                // Sp -> byte offsets of the PQ / PK rows per relative distance (band kernel, 16-bit)
    std::map<int, std::pair<int, int>> dsat;   // Sp -> (rsat_pos, rsat_neg)
    std::map<int, std::pair<void*, int4*>> mx2tabs;   // Sp -> (idx16, tinfo) of attention_mx2.hip; (null, null): this table keeps the band kernel
    bool mx2 = false;                          // MX attention on the bucket-space kernel (attention_mx2.hip; needs its tables) instead of the band kernel (attention_mx.hip): opt-in
                                               // (GLC_ATTN_MX2=1, glc_debug_set_mx2) — measured 4-5 % slower at c3 (docs/LOG_r01-r05.md §3f)
    // last forward
    int lastB = 0, lastS = 0, lastSp = 0;
    // debug
    bool keep_hidden = false; void* hidden_dump = nullptr; size_t hidden_cap = 0;
    // timing / profile
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool profile = false;
    struct Ev { hipEvent_t a, b; int cls; };
    std::vector<Ev> evs; size_t ev_used = 0;
    float prof_ms[PC_N] = {0}; int prof_n[PC_N] = {0};
};

namespace {

constexpr int kActScLow = -5;
// the launchers' view of the engine for one forward (glc_kernels.h): fp8 guard counter and activation exponent, reset behind it
struct GxScope {
    explicit GxScope(glc_engine* e) { glc_gx_sat_ptr() = e->d_gxsat; glc_gx_act_sc() = e->act_sc; }
    ~GxScope() { glc_gx_sat_ptr() = nullptr; glc_gx_act_sc() = 0; }
};

void* dmalloc(glc_engine* e, size_t bytes, bool zero = true) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) { set_err("hipMalloc failed for " + std::to_string(bytes) + " bytes"); return nullptr; }
    if (zero && hipMemsetAsync(p, 0, bytes, e->stream) != hipSuccess) { set_err("hipMemset failed"); return nullptr; }
    e->allocs.push_back(p);
    return p;
}
void dfree(glc_engine* e, void* p) {
    if (!p) return;
    for (size_t i = 0; i < e->allocs.size(); ++i)
        if (e->allocs[i] == p) { e->allocs[i] = e->allocs.back(); e->allocs.pop_back(); break; }
    (void)hipFree(p);
}

struct Prof {
    glc_engine* e; int idx = -1;
    Prof(glc_engine* e_, int cls) : e(e_) {
        if (!e->profile) return;
        if (e->ev_used == e->evs.size()) {
            glc_engine::Ev v{};
            if (hipEventCreate(&v.a) != hipSuccess || hipEventCreate(&v.b) != hipSuccess) return;
            e->evs.push_back(v);
        }
        idx = (int)e->ev_used++;
        e->evs[idx].cls = cls;
        (void)hipEventRecord(e->evs[idx].a, e->stream);
    }
    ~Prof() { if (idx >= 0) (void)hipEventRecord(e->evs[idx].b, e->stream); }
};

void prof_collect(glc_engine* e) {
    for (size_t i = 0; i < e->ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->evs[i].a, e->evs[i].b) == hipSuccess) { e->prof_ms[e->evs[i].cls] += ms; e->prof_n[e->evs[i].cls]++; }
    }
    e->ev_used = 0;
}

// host f32 -> device f32
float* upload_f32(glc_engine* e, const float* src, size_t n) {
    float* d = (float*)dmalloc(e, n * sizeof(float), false);
    if (!d) return nullptr;
    if (hipMemcpyAsync(d, src, n * sizeof(float), hipMemcpyHostToDevice, e->stream) != hipSuccess) { set_err("H2D failed"); return nullptr; }
    return d;
}
// host f32 -> device T at dst (dst preallocated), via a temporary f32 staging buffer
bool upload_as(glc_engine* e, const float* src, size_t n, void* dst, float* staging) {
    if (hipMemcpyAsync(staging, src, n * sizeof(float), hipMemcpyHostToDevice, e->stream) != hipSuccess) { set_err("H2D failed"); return false; }
    const char* m = glc_launch_convert(e->stream, e->dtype, staging, dst, n);
    if (m) { set_err(m); return false; }
    return hipStreamSynchronize(e->stream) == hipSuccess;   // staging is reused by the caller
}

constexpr int kFp8Sticky = 2;
bool init_range_guard(glc_engine* e) {
    if (e->d_gxsat) return true;
    e->d_gxsat = (unsigned*)dmalloc(e, 2 * sizeof(unsigned), false);
    if (!e->d_gxsat || hipMemset(e->d_gxsat, 0, 2 * sizeof(unsigned)) != hipSuccess) { set_err("range-guard counter alloc failed"); return false; }
    if (hipHostMalloc((void**)&e->h_gxsat, 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) { e->h_gxsat = nullptr; set_err("range-guard host slot alloc failed"); return false; }
    e->h_gxsat[0] = e->h_gxsat[1] = 0;
    return true;
}
bool ensure_capacity(glc_engine* e, int B, int S, int C) {
    const glc_model_config& c = e->cfg;
    if (!init_range_guard(e)) return false;
    if (!e->splitk_ws) {
        e->splitk_ws_bytes = (size_t)64 << 20;
        e->splitk_ws = (float*)dmalloc(e, e->splitk_ws_bytes, false);
        if (!e->splitk_ws) { e->splitk_ws_bytes = 0; return false; }
    }
    const int Sp = round_up(S, 64), M = B * Sp, Mpad = round_up(M, 256);
    const size_t es = esize(e->dtype);
    const bool dec = c.backbone == GLC_BACKBONE_DECODER;
    if (Mpad > e->capM && dec) {
        const size_t nqd = (size_t)c.heads * c.head_dim, nkvd = (size_t)c.kv_heads * c.head_dim;
        void** bufs[] = {&e->X, &e->X2, &e->H1};
        for (void** b : bufs) { dfree(e, *b); *b = dmalloc(e, (size_t)Mpad * c.hidden * es); if (!*b) return false; }
        dfree(e, e->QKV); e->QKV = dmalloc(e, (size_t)Mpad * (nqd + 2 * nkvd) * es); if (!e->QKV) return false;
        dfree(e, e->CTX); e->CTX = dmalloc(e, (size_t)Mpad * nqd * es); if (!e->CTX) return false;
        if (e->dtype != GLC_F32 || e->dec_split) {          // fragment-major operands of the MFMA attention kernel (fp32 mode: split-f16 units, same bytes)
            dfree(e, e->Qh); e->Qh = dmalloc(e, (size_t)Mpad * nqd * es); if (!e->Qh) return false;
            dfree(e, e->Kh); e->Kh = dmalloc(e, (size_t)Mpad * nkvd * es); if (!e->Kh) return false;
            dfree(e, e->Vt); e->Vt = dmalloc(e, (size_t)Mpad * nkvd * es); if (!e->Vt) return false;
        }
        // [gate | up] rows: read only by the launches without the SwiGLU epilogue (unfused builds, the fp32 mode's small forwards) — allocated
        // by the forward that takes such a launch (need_gu), not for every capacity step (2.3 GB at c5 that the group-split pipeline never touches)
        dfree(e, e->GU); e->GU = nullptr; e->capGU = 0;
        dfree(e, e->FF); e->FF = dmalloc(e, (size_t)Mpad * c.inter * es); if (!e->FF) return false;
        if (e->dtype == GLC_F32) {             // RMSNorm statistics of the two residual-stream buffers + the producers' partials
            dfree(e, e->statsA); e->statsA = (float2*)dmalloc(e, (size_t)Mpad * sizeof(float2)); if (!e->statsA) return false;
            dfree(e, e->statsB); e->statsB = (float2*)dmalloc(e, (size_t)Mpad * sizeof(float2)); if (!e->statsB) return false;
            dfree(e, e->ln_part); e->ln_part = (float2*)dmalloc(e, (size_t)Mpad * ((c.hidden + 63) / 64) * sizeof(float2)); if (!e->ln_part) return false;
        }
        dfree(e, e->kbias); e->kbias = (float*)dmalloc(e, (size_t)Mpad * sizeof(float)); if (!e->kbias) return false;
        e->capM = Mpad;
        e->hidden_cap = 0;
    }
    if (Mpad > e->capM) {
        void** bufs[] = {&e->X, &e->Qh, &e->Kh, &e->Vt, &e->CTX, &e->T1, &e->H1};
        for (void** b : bufs) { dfree(e, *b); *b = dmalloc(e, (size_t)Mpad * c.hidden * es); if (!*b) return false; }
        dfree(e, e->FF); e->FF = dmalloc(e, (size_t)Mpad * c.inter * es); if (!e->FF) return false;
        {                                      // (small; kept whether or not the fold is switched on: glc_debug_set_ln_fused)
            dfree(e, e->statsA); e->statsA = (float2*)dmalloc(e, (size_t)Mpad * sizeof(float2)); if (!e->statsA) return false;
            dfree(e, e->statsB); e->statsB = (float2*)dmalloc(e, (size_t)Mpad * sizeof(float2)); if (!e->statsB) return false;
            dfree(e, e->ln_part); e->ln_part = (float2*)dmalloc(e, (size_t)Mpad * ((c.hidden + 63) / 64) * sizeof(float2)); if (!e->ln_part) return false;
        }
        dfree(e, e->kbias); e->kbias = (float*)dmalloc(e, (size_t)Mpad * sizeof(float)); if (!e->kbias) return false;
        e->capM = Mpad;
        e->hidden_cap = 0;   // dump buffer is re-made lazily
    }
    if (B > e->capB || C > e->capC) {
        const int nb = B > e->capB ? B : e->capB, nc = C > e->capC ? C : e->capC;
        dfree(e, e->klen); dfree(e, e->kfirst); dfree(e, e->cls_cnt); dfree(e, e->cls_pos); dfree(e, e->d_logits);
        e->klen = (int*)dmalloc(e, (size_t)nb * sizeof(int));
        e->kfirst = (int*)dmalloc(e, (size_t)nb * sizeof(int));
        e->cls_cnt = (int*)dmalloc(e, (size_t)nb * sizeof(int));
        e->cls_pos = (int*)dmalloc(e, (size_t)nb * (nc > 0 ? nc : 1) * sizeof(int));
        e->d_logits = (float*)dmalloc(e, (size_t)nb * (nc > 0 ? nc : 1) * sizeof(float));
        if (!e->klen || !e->kfirst || !e->cls_cnt || !e->cls_pos || !e->d_logits) return false;
        e->capB = nb; e->capC = nc;
    }
    if (B * S > e->capIds) {
        dfree(e, e->d_ids); dfree(e, e->d_mask);
        e->d_ids = (int64_t*)dmalloc(e, (size_t)B * S * sizeof(int64_t));
        e->d_mask = (int64_t*)dmalloc(e, (size_t)B * S * sizeof(int64_t));
        if (!e->d_ids || !e->d_mask) return false;
        e->capIds = B * S;
    }
    const int hr = round_up(B, 128) + round_up(B * (C > 0 ? C : 1), 128);      // text rows then class rows, each group 128-aligned
    if (hr > e->capHeadRows) {
        float** bufs[] = {&e->Gt, &e->G1t, &e->G2t};
        for (float** b : bufs) { dfree(e, *b); *b = (float*)dmalloc(e, (size_t)hr * c.hidden * sizeof(float)); if (!*b) return false; }
        if (c.scorer != GLC_SCORER_DOT) {      // weighted-dot: [hr, 2H] + [rc, 3H] + [rc, 4H]; mlp: [rc, 2H] + [rc, 256] + [rc, 128] (rc <= hr)
            dfree(e, e->scorer_ws);
            e->scorer_ws = (float*)dmalloc(e, (size_t)hr * (9 * (size_t)c.hidden + 2 * GLC_SCORER_MLP_HIDDEN) * sizeof(float));
            if (!e->scorer_ws) return false;
        }
        e->capHeadRows = hr;
    }
    if (dec) {
        if (!e->ropes.count(Sp)) {
            // Q2:86 inv_freq = 1 / base^(arange(0,d,2)/d) in float32; Q2:97-100 freqs = inv_freq * position, then cos / sin
            const int hd2 = c.head_dim / 2;
            std::vector<float> t((size_t)Sp * hd2 * 2);
            for (int s = 0; s < Sp; ++s)
                for (int i = 0; i < hd2; ++i) {
                    const float inv = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)c.head_dim);
                    const float f = inv * (float)s;
                    t[((size_t)s * hd2 + i) * 2] = cosf(f);
                    t[((size_t)s * hd2 + i) * 2 + 1] = sinf(f);
                }
            float* d = (float*)dmalloc(e, t.size() * sizeof(float), false);
            if (!d) return false;
            if (hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { set_err("rope table upload failed"); return false; }
            e->ropes[Sp] = d;
        }
        return true;
    }
    const int rsel = round_up(B * (1 + (C > 0 ? C : 0)), 256);
    if (rsel > e->capSel) {
        void** bufs[] = {&e->Xs, &e->Qs, &e->CTXs, &e->T1s, &e->H1s};
        for (void** b : bufs) { dfree(e, *b); *b = dmalloc(e, (size_t)rsel * c.hidden * es); if (!*b) return false; }
        dfree(e, e->FFs); e->FFs = dmalloc(e, (size_t)rsel * c.inter * es); if (!e->FFs) return false;
        dfree(e, e->sel_b); dfree(e, e->sel_q);
        e->sel_b = (int*)dmalloc(e, (size_t)rsel * sizeof(int)); e->sel_q = (int*)dmalloc(e, (size_t)rsel * sizeof(int));
        if (!e->sel_b || !e->sel_q) return false;
        e->capSel = rsel;
    }
    {
        const size_t nf = (size_t)round_up(B * Sp, 256) >> 5;      // one byte per 32-row tile of the padded row grid (the QKV GEMM reads 8 per 256-row tile)
        if (nf > e->capFlag) { dfree(e, e->tile_flag); e->tile_flag = (unsigned char*)dmalloc(e, nf); if (!e->tile_flag) return false; e->capFlag = nf; }
    }
    if (!e->dtabs.count(Sp)) {
        std::vector<int32_t> t(2 * Sp - 1);
        glc_delta_table(Sp, c.pos_buckets, c.max_rel_pos, t.data());
        int32_t* d = (int32_t*)dmalloc(e, t.size() * sizeof(int32_t), false);
        if (!d) return false;
        if (hipMemcpy(d, t.data(), t.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { set_err("dtab upload failed"); return false; }
        e->dtabs[Sp] = d;
        {
            // band-kernel addressing table: entry j <-> relative distance clamp(j - 64, 0, 2Sp-2) - (Sp-1); x = byte offset of PQ row
            // delta in the Q fragment layout, y = byte offset of PK row delta in the K layout (pi on the row) — glc_layout.h
            std::vector<int2> o(t.size() + 128);
            for (size_t j = 0; j < o.size(); ++j) {
                const long long r = (long long)j - 64;
                const int dl = t[r < 0 ? 0 : (r > (long long)t.size() - 1 ? t.size() - 1 : (size_t)r)];
                o[j].x = ((dl >> 5) * 2048 + (dl & 31) * 8) * (int)es;
                o[j].y = ((dl >> 5) * 2048 + glc_pi32(dl & 31) * 8) * (int)es;
            }
            int2* od = (int2*)dmalloc(e, o.size() * sizeof(int2), false);
            if (!od) return false;
            if (hipMemcpy(od, o.data(), o.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) { set_err("otab upload failed"); return false; }
            e->otabs[Sp] = od;
            if (es == 4) {
                // MX band kernel (attention_mx.hip, round 6): the same rows as ready-made load offsets into the PLANAR tile images of the position tables
                // (glc_layout.h) — per lane half hh and entry j (NE = 2 Sp + 512 entries: 64 clamped ones in front, the rest behind, so that no index the
                // kernel forms needs a clamp): .x = offset of the PQ row of entry j, .y = offset of the PK row of entry NE - 1 - j (the PK half runs
                // BACKWARDS: a lane that adds its column c to a wave-uniform base walks the PQ rows up and the PK rows down with the same per-lane
                // register); offset = tile * 8192 + slot * 16 + hh * 512: the lane's 16 bytes of f16 unit 0; unit s at + s * 1024, MX plane k at + 4096 + k * 1024
                const int NE = 2 * Sp + 512;
                std::vector<int2> m((size_t)2 * NE);
                auto dl_of = [&](int j) { const long long r = (long long)j - 64; return t[r < 0 ? 0 : (r > (long long)t.size() - 1 ? t.size() - 1 : (size_t)r)]; };
                for (int hh = 0; hh < 2; ++hh)
                    for (int j = 0; j < NE; ++j) {
                        const int dq = dl_of(j), dk = dl_of(NE - 1 - j);
                        m[(size_t)hh * NE + j] = make_int2((dq >> 5) * 8192 + (dq & 31) * 16 + hh * 512, (dk >> 5) * 8192 + glc_pi32(dk & 31) * 16 + hh * 512);
                    }
                int2* md = (int2*)dmalloc(e, m.size() * sizeof(int2), false);
                if (!md) return false;
                if (hipMemcpy(md, m.data(), m.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) { set_err("mtab upload failed"); return false; }
                e->mtabs[Sp] = md;
            }
        }
        // saturation points of the table: delta == P-1 for every q-k >= rsat_pos, delta == 0 for every q-k <= rsat_neg
        int rp = Sp, rn = -Sp;
        for (int r = Sp - 1; r >= -(Sp - 1) && t[r + Sp - 1] == 2 * (c.pos_buckets > 0 ? c.pos_buckets : c.max_rel_pos) - 1; --r) rp = r;
        for (int r = -(Sp - 1); r <= Sp - 1 && t[r + Sp - 1] == 0; ++r) rn = r;
        e->dsat[Sp] = std::make_pair(rp, rn);
        // tables of the bucket-space MX attention (attention_mx2.hip); a table without the structure it needs keeps the band kernel
        std::vector<unsigned char> idx16;
        std::vector<int4> tinfo;
        void* d_idx = nullptr; int4* d_ti = nullptr;
        if (e->dtype == GLC_F32 && glc_mx2_build_tables(Sp, e->P, t.data(), idx16, tinfo)) {
            d_idx = dmalloc(e, idx16.size(), false);
            d_ti = (int4*)dmalloc(e, tinfo.size() * sizeof(int4), false);
            if (!d_idx || !d_ti) return false;
            if (hipMemcpy(d_idx, idx16.data(), idx16.size(), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_ti, tinfo.data(), tinfo.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) { set_err("mx2 table upload failed"); return false; }
        }
        e->mx2tabs[Sp] = std::make_pair(d_idx, d_ti);
    }
    return true;
}

// the scorer's tensors (after the 8 projector tensors): matrices that a GEMM reads are pre-split like the projectors', the last Linear
// (out features 1) stays a plain vector
bool upload_scorer(glc_engine* e, const float* const* t) {
    const int H = e->cfg.hidden, Mh = GLC_SCORER_MLP_HIDDEN;
    size_t n[8] = {0};
    bool mat[8] = {false};
    int cnt = 0;
    if (e->cfg.scorer == GLC_SCORER_WEIGHTED_DOT) {
        const size_t v[8] = {2 * (size_t)H * H, 2 * (size_t)H, 2 * (size_t)H * H, 2 * (size_t)H, 12 * (size_t)H * H, 4 * (size_t)H, 4 * (size_t)H, 1};
        for (int i = 0; i < 8; ++i) n[i] = v[i];
        mat[0] = mat[2] = mat[4] = true; cnt = 8;
    } else if (e->cfg.scorer == GLC_SCORER_MLP) {
        const size_t v[6] = {(size_t)Mh * 2 * H, (size_t)Mh, (size_t)Mh / 2 * Mh, (size_t)Mh / 2, (size_t)Mh / 2, 1};
        for (int i = 0; i < 6; ++i) n[i] = v[i];
        mat[0] = mat[2] = true; cnt = 6;
    }
    for (int i = 0; i < cnt; ++i) {
        e->scw[i] = upload_f32(e, t[i], n[i]);
        if (!e->scw[i]) return false;
        if (mat[i] && e->w_presplit) { const char* pm = glc_launch_presplit(e->stream, e->scw[i], n[i]); if (pm) { set_err(pm); return false; } }
    }
    return true;
}

// GEMM launches of the forward carry the engine's split-K workspace (used only when a shape has too few tiles, gemm.hip)
static bool presplit_weight(const glc_engine* e, int dt, const void* W) {
    if (!e->w_presplit || dt != GLC_F32) return false;
    return true;                                                                 // head projectors and every layer weight of the fp32 mode
}
const char* launch_gemm_auto(glc_engine* e, int dt, int epi, GemmArgs a) {
    a.ws = e->splitk_ws; a.ws_bytes = e->splitk_ws_bytes; a.w_presplit = presplit_weight(e, dt, a.W);
    return glc_launch_gemm_auto(e->stream, dt, epi, a);
}
const char* launch_gemm128(glc_engine* e, int dt, int epi, GemmArgs a) {
    a.ws = e->splitk_ws; a.ws_bytes = e->splitk_ws_bytes; a.w_presplit = presplit_weight(e, dt, a.W);
    return glc_launch_gemm(e->stream, dt, epi, a);
}

// MX pipeline: the GX copies of the projection weights (hi f16 | fp8 parts, glc_common.h; each tensor with the fp8 exponent its largest
// magnitude allows), made on the device from the split-f16 (group-split) copies by the FIRST forward that takes the pipeline — an engine
// whose forwards never reach the 256-tile pipeline (the reference's own batches of 8 short texts) never pays their memory.
// (the body; `made` collects every pointer slot it fills so that build_mx_weights can undo a partial build)
static bool build_mx_weights_impl(glc_engine* e, std::vector<void**>& made) {
    unsigned* d_bits = (unsigned*)e->splitk_ws;          // (scratch: the split-K workspace is idle between launches)
    if (!d_bits) { set_err("MX weights: no scratch"); return false; }
    auto copy = [&](const void* gs, size_t n, void*& dst, int& ws) -> bool {
        if (!gs) { set_err("MX weights: a split-f16 source copy is missing"); return false; }
        unsigned bits = 0;
        HIPCHK(hipMemsetAsync(d_bits, 0, sizeof(unsigned), e->stream), false);
        KCHK(glc_launch_gs_absmax(e->stream, gs, n, d_bits), false);
        HIPCHK(hipMemcpyAsync(&bits, d_bits, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream), false);
        HIPCHK(hipStreamSynchronize(e->stream), false);
        float mxv; memcpy(&mxv, &bits, sizeof(float));
        ws = glc_gx_weight_exponent(mxv);
        dst = dmalloc(e, n * sizeof(float), false);
        if (!dst) return false;
        made.push_back(&dst);
        KCHK(glc_launch_gs_to_gx(e->stream, gs, dst, n, ws), false);
        e->mx_bytes += n * sizeof(float);
        return true;
    };
    const glc_model_config& c = e->cfg;
    const size_t H = c.hidden, I = c.inter;
    if (c.backbone == GLC_BACKBONE_DECODER) {
        const size_t NQ = (size_t)c.heads * c.head_dim, NQKV = NQ + 2 * (size_t)c.kv_heads * c.head_dim;
        // RoPE + MX tiles as the epilogue of the QKV projection (gemm256x EPI_QKVR; head_dim 128, even head counts): the rows of every Q / K head
        // of the GX copy (and a copy of the bias) go into the order glc_rope_perm128 names — 32-row blocks 1 and 2 of the head trade places
        const bool rope_epi = e->dec_rope_epi && c.head_dim == 128 && c.heads % 2 == 0 && c.kv_heads % 2 == 0;
        void* tmp = nullptr;
        struct TmpGuard { glc_engine* e; void*& p; ~TmpGuard() { if (p) { (void)hipStreamSynchronize(e->stream); dfree(e, p); p = nullptr; } } } tmp_guard{e, tmp};      // (every exit path)
        const size_t blk = (size_t)32 * H * sizeof(float);        // 32 GX rows
        if (rope_epi) { tmp = dmalloc(e, blk, false); if (!tmp) return false; }
        for (auto& w : e->dlayers) {
            if (!copy(w.Wqkvf, NQKV * H, w.Wqkvf_x, w.ws_qkvf) || !copy(w.Wo, H * NQ, w.Wo_x, w.ws_o) ||
                !copy(w.Wguf, 2 * I * H, w.Wguf_x, w.ws_guf) || !copy(w.Wd, H * I, w.Wd_x, w.ws_d)) return false;
            if (!rope_epi) continue;
            w.bqkv_p = (float*)dmalloc(e, NQKV * sizeof(float), false);
            if (!w.bqkv_p) return false;
            made.push_back(reinterpret_cast<void**>(&w.bqkv_p));
            HIPCHK(hipMemcpyAsync(w.bqkv_p, w.bqkv, NQKV * sizeof(float), hipMemcpyDeviceToDevice, e->stream), false);
            for (int hd = 0; hd < c.heads + c.kv_heads; ++hd) {
                char* r1 = (char*)w.Wqkvf_x + ((size_t)hd * 128 + 32) * H * sizeof(float);
                char* r2 = r1 + blk;
                HIPCHK(hipMemcpyAsync(tmp, r1, blk, hipMemcpyDeviceToDevice, e->stream), false);
                HIPCHK(hipMemcpyAsync(r1, r2, blk, hipMemcpyDeviceToDevice, e->stream), false);
                HIPCHK(hipMemcpyAsync(r2, tmp, blk, hipMemcpyDeviceToDevice, e->stream), false);
                HIPCHK(hipMemcpyAsync(w.bqkv_p + hd * 128 + 32, w.bqkv + hd * 128 + 64, 32 * sizeof(float), hipMemcpyDeviceToDevice, e->stream), false);
                HIPCHK(hipMemcpyAsync(w.bqkv_p + hd * 128 + 64, w.bqkv + hd * 128 + 32, 32 * sizeof(float), hipMemcpyDeviceToDevice, e->stream), false);
            }
        }
    } else {
        for (size_t l = 0; l < e->layers.size(); ++l) {
            LayerW& w = e->layers[l];
            if (!copy(w.Wqkv, 3 * H * H, w.Wqkv_x, w.ws_qkv) || !copy(w.Wo, H * H, w.Wo_x, w.ws_o) || !copy(w.W2, H * I, w.W2_x, w.ws_2) ||
                !copy(w.W1f, I * H, w.W1f_x, w.ws_1f)) return false;
            if (l > 0 && !copy(w.Wqkvf, 3 * H * H, w.Wqkvf_x, w.ws_qkvf)) return false;
        }
    }
    HIPCHK(hipStreamSynchronize(e->stream), false);
    return true;
}

// The GX copies of the projection weights (+0.4 / 1.4 / 5.2 GB for base / large / the Qwen2-1.5B shape), built by the first forward that takes the
// MX pipeline.  A failure (out of memory, most likely) must not fail that forward, leak per call or leave a layer half converted:
// whatever this call allocated is freed, the pointers are cleared, and the engine stays on the split-f16 kernels (as GLICLASS_MX=0) from then on.
bool build_mx_weights(glc_engine* e) {
    if (e->mx_ready) return true;
    if (!e->mx_built) return false;
    std::vector<void**> made;
    if (build_mx_weights_impl(e, made)) { e->mx_ready = true; return true; }
    (void)hipStreamSynchronize(e->stream);
    (void)hipGetLastError();
    for (void** slot : made) { dfree(e, *slot); *slot = nullptr; }
    e->mx_bytes = 0;
    e->mx_built = false;
    e->mx = false;
    fprintf(stderr, "gliclass: the MX weight copies could not be built (%s); this engine keeps the split-f16 arithmetic (as GLICLASS_MX=0)\n", glc_last_error());
    return false;
}

// The head after the row gather: the two FeaturesProjectors (Linear -> GELU -> Linear; text rows [0, rt) and class rows [rt, rt + rc)
// share one fp32 buffer, one two-group GEMM per stage), then the scorer (include/gliclass_hip.h).
bool run_head_tail(glc_engine* e, int B, int C, float* d_logits) {
    const glc_model_config& c = e->cfg;
    const int H = c.hidden;
    hipStream_t st = e->stream;
    const int rt = round_up(B, 128), rc = round_up(B * C, 128);
    GemmArgs h;
    h.N = H; h.K = H; h.Mpad = rt + rc; h.m_split = rt;
    h.A = e->Gt; h.W = e->headw[0]; h.bias = e->headw[1]; h.W2 = e->headw[4]; h.bias2 = e->headw[5]; h.C = e->G1t;
    KCHK(launch_gemm128(e, GLC_F32, EPI_GELU, h), false);
    h.A = e->G1t; h.W = e->headw[2]; h.bias = e->headw[3]; h.W2 = e->headw[6]; h.bias2 = e->headw[7]; h.C = e->G2t;
    KCHK(launch_gemm128(e, GLC_F32, EPI_BIAS, h), false);
    const float* Tt = e->G2t;
    const float* Cc = e->G2t + (size_t)rt * H;
    // normalize_features (upstream: projected text and class features are L2-normalised before ANY scorer and the logits multiplied by
    // logit_scale): the dot scorer does both inside its kernel; the other scorers get a row pass over G2 and the scale in their last step
    const float lscale = c.normalize_features ? c.logit_scale : 1.0f;
    if (c.scorer != GLC_SCORER_DOT && c.normalize_features) KCHK(glc_launch_l2norm_rows(st, e->G2t, rt + rc, H), false);
    if (c.scorer == GLC_SCORER_DOT) {
        KCHK(glc_launch_head_score(st, Tt, Cc, d_logits, B, C, H, c.normalize_features, c.logit_scale), false);
    } else if (c.scorer == GLC_SCORER_WEIGHTED_DOT) {
        float* S1 = e->scorer_ws;                                // [rt + rc, 2H]: proj_text on the text rows, proj_label on the class rows
        float* cat = S1 + (size_t)(rt + rc) * 2 * H;             // [rc, 3H]
        float* hid = cat + (size_t)rc * 3 * H;                   // [rc, 4H]
        GemmArgs g;
        g.N = 2 * H; g.K = H; g.Mpad = rt + rc; g.m_split = rt;
        g.A = e->G2t; g.W = e->scw[0]; g.bias = e->scw[1]; g.W2 = e->scw[2]; g.bias2 = e->scw[3]; g.C = S1;
        KCHK(launch_gemm128(e, GLC_F32, EPI_BIAS, g), false);
        KCHK(glc_launch_scorer_wd_cat(st, S1, S1 + (size_t)rt * 2 * H, cat, B, C, H), false);
        GemmArgs o;
        o.N = 4 * H; o.K = 3 * H; o.Mpad = rc; o.A = cat; o.W = e->scw[4]; o.bias = e->scw[5]; o.C = hid;
        KCHK(launch_gemm128(e, GLC_F32, EPI_BIAS, o), false);
        KCHK(glc_launch_relu_dot(st, hid, e->scw[6], e->scw[7], d_logits, B * C, 4 * H, lscale), false);
    } else {
        const int Mh = GLC_SCORER_MLP_HIDDEN;
        float* pair = e->scorer_ws;                              // [rc, 2H] = [text | class]
        float* h1 = pair + (size_t)rc * 2 * H;                   // [rc, 256]
        float* h2 = h1 + (size_t)rc * Mh;                        // [rc, 128]
        KCHK(glc_launch_scorer_pair(st, Tt, Cc, pair, B, C, H), false);
        GemmArgs g;
        g.N = Mh; g.K = 2 * H; g.Mpad = rc; g.A = pair; g.W = e->scw[0]; g.bias = e->scw[1]; g.C = h1;
        KCHK(launch_gemm128(e, GLC_F32, EPI_BIAS, g), false);
        KCHK(glc_launch_relu(st, h1, (size_t)rc * Mh), false);
        GemmArgs g2;
        g2.N = Mh / 2; g2.K = Mh; g2.Mpad = rc; g2.A = h1; g2.W = e->scw[2]; g2.bias = e->scw[3]; g2.C = h2;
        KCHK(launch_gemm128(e, GLC_F32, EPI_BIAS, g2), false);
        KCHK(glc_launch_relu_dot(st, h2, e->scw[4], e->scw[5], d_logits, B * C, Mh / 2, lscale), false);
    }
    return true;
}

// Decoder-style backbone: one launch sequence per batch (Q2:384-398).  Pre-norm residual stream X (operand type T).
bool run_forward_decoder(glc_engine* e, const int64_t* ids, const int64_t* mask, int B, int S, int C, float* d_logits) {
    const glc_model_config& c = e->cfg;
    const int H = c.hidden, I = c.inter, nq = c.heads, nkv = c.kv_heads, d = c.head_dim, L = c.layers;
    const int NQ = nq * d, NQKV = (nq + 2 * nkv) * d;
    const int Sp = round_up(S, 64), M = B * Sp, Mpad = round_up(M, 256);
    hipStream_t st = e->stream;
    const int dt = e->dtype;
    const size_t es = esize(dt);
    const int ccap = e->capC > 0 ? e->capC : 1;
    if (e->profile) { e->ev_used = 0; }
    GxScope gx_scope(e);                     // fp8 range guard + activation exponent: the launchers of this thread hand them to every producer / reader of GX rows and MX tiles
    if (e->keep_hidden) {
        const size_t need = (size_t)(L + 1) * M * H * es;
        if (need > e->hidden_cap) { dfree(e, e->hidden_dump); e->hidden_dump = dmalloc(e, need); if (!e->hidden_dump) return false; e->hidden_cap = need; }
    }
    { Prof p(e, PC_SCAN);
      KCHK(glc_launch_scan_rows(st, ids, mask, B, S, c.class_token_index, c.embed_class_token, e->klen, e->kfirst, e->cls_pos, e->cls_cnt, ccap), false); }
    { Prof p(e, PC_EMBED);
      KCHK(glc_launch_embed_plain(st, dt, ids, mask, e->emb, e->X, e->kbias, B, S, Sp, H, c.vocab, c.pad_id), false); }
    if (e->keep_hidden) HIPCHK(hipMemcpyAsync(e->hidden_dump, e->X, (size_t)M * H * es, hipMemcpyDeviceToDevice, st), false);
    const float qscale = 1.4426950408889634f / sqrtf((float)d);         // Q2:186 scaling, times log2(e) for the exp2 softmax
    const bool mfma = (dt != GLC_F32 || e->dec_split) && e->attn_impl != 1;   // flash-style MFMA kernel (fp32 mode: on split-f16 units); impl 1 / native fp32: straightforward kernel
    // group-split pipeline of the fp32 mode (see run_forward): the GEMM A operands (RMSNorm outputs, context rows, SwiGLU output) are
    // written as [32 hi | 32 lo] groups by their producers and every projection runs on the 256-tile LDS-DMA kernel; the residual
    // stream X, the fused QKV rows (RoPE / layout pass) and the [gate | up] rows stay plain fp32
    bool gs = false;
    if (dt == GLC_F32 && e->gs_mode > 0 && e->w_presplit && e->dec_split && mfma && !e->keep_hidden && H % 256 == 0 && (2 * I) % 256 == 0 &&
        NQKV % 256 == 0 && NQ % 32 == 0 && I % 32 == 0) {
        GemmArgs t; t.Mpad = Mpad; t.N = H; t.K = H;
        gs = e->gs_mode == 2 || !glc_gemm_small_m(t);
    }
    e->last_gs = gs;
    void *X = e->X, *Xn = e->X2;
    // RMSNorm folded away (group-split pipeline, e->ln_fused; docs/LOG_r01-r05.md §3d): the residual stream is kept as RAW group-split rows plus
    // (0, rstd) per row — statsA for the buffer X points at, statsB for the other — the projections run on weights with the RMSNorm gain
    // folded in and scale their accumulators by rstd in the epilogue; the residual GEMMs emit the partials of the next statistics.
    const bool rnf = gs && e->ln_fused && e->fused_swiglu && L > 0 && e->dlayers[0].Wqkvf && e->dlayers[0].Wguf && e->statsA && e->statsB && e->ln_part;
    float2 *sX = e->statsA, *sXn = e->statsB;
    e->last_lnf = rnf;
    // MX pipeline (round 3, as the encoder's): GX rows + gemm256x for the four projections of every layer; attention stays on split units
    bool mx = rnf && e->mx && e->mx_built && e->prec_mask == 0;
    if (mx && !build_mx_weights(e)) mx = false;          // (no copies: this and every later forward stay on the split-f16 kernels)
    for (int l = 0; mx && l < L; ++l) mx = e->dlayers[l].Wqkvf_x && e->dlayers[l].Wo_x && e->dlayers[l].Wguf_x && e->dlayers[l].Wd_x;
    e->last_mx = mx;
    const bool mxa = mx && mfma && e->mx_attn;      // round 4: the attention of the MX pipeline on MX tiles too (decoder_mx.hip)
    e->last_mx_attn = mxa;
    auto gemm_gs = [&](int epi, const GemmArgs& ga) -> const char* {
        if (!mx) return glc_launch_gemm256s_gs(st, epi, ga);
        GemmArgs gx = ga; gx.gx_rows = M;          // fp8 range guard: the M rows of this forward, not the slack rows up to Mpad
        return glc_launch_gemm256x(st, epi, gx);
    };
    if (rnf) {      // the embedding rows enter the pipeline: plain fp32 (X2) -> raw group-split rows (X) + statistics
        HIPCHK(hipMemcpyAsync(e->X2, e->X, (size_t)M * H * es, hipMemcpyDeviceToDevice, st), false);
        KCHK(glc_launch_rows_to_gs_rms(st, (const float*)e->X2, e->X, sX, c.ln_eps, M, H, mx ? 1 : 0), false);
    }
    for (int l = 0; l < L; ++l) {
        const DecLayerW& w = e->dlayers[l];
        if (!rnf) { Prof p(e, PC_LN); KCHK(gs ? glc_launch_rmsnorm_gs(st, (const float*)X, e->H1, w.ln1, c.ln_eps, M, H)
                                    : glc_launch_rmsnorm(st, dt, X, e->H1, w.ln1, c.ln_eps, M, H), false); }                        // Q2:280
        GemmArgs g;
        g.A = e->H1; g.W = w.Wqkv; g.bias = w.bqkv; g.C = e->QKV; g.Mpad = Mpad; g.N = NQKV; g.K = H; g.gs_c_plain = 1;
        if (rnf) { g.A = X; g.W = w.Wqkvf; g.a_stats = sX; }
        if (mx) { g.W = w.Wqkvf_x; g.mx_ws = w.ws_qkvf; }
        const bool perm = mx && w.bqkv_p;            // Wqkvf_x rows in the RoPE-epilogue order
        if (perm) { g.bias = w.bqkv_p; g.perm_cols = (nq + nkv) * d; }
        const bool rope_epi = perm && mxa;           // Q2:206-211 in one launch: projection, RoPE, scale, MX tiles (gemm256x.hip EPI_QKVR)
        if (rope_epi) { g.rope_cs = e->ropes[Sp]; g.qscale = qscale; g.nq = nq; g.nkv = nkv; g.Sp = Sp; g.Mvalid = M; g.Qh = e->Qh; g.Kh = e->Kh; g.Vt = e->Vt; }
        { Prof p(e, PC_QKV);
          if (rope_epi) KCHK(gemm_gs(EPI_QKVR, g), false);
          else {
          KCHK(gs ? gemm_gs(EPI_BIAS, g) : launch_gemm_auto(e, dt, EPI_BIAS, g), false);        // Q2:206-208
          if (mxa) KCHK(glc_launch_qkv_layout_mx(st, e->QKV, e->ropes[Sp], e->Qh, e->Kh, e->Vt, B, Sp, nq, nkv, d, qscale), false);     // Q2:211 RoPE, MX tiles (decoder_mx.hip)
          else if (mfma) KCHK(glc_launch_qkv_layout(st, dt, e->QKV, e->ropes[Sp], e->Qh, e->Kh, e->Vt, B, Sp, nq, nkv, d, qscale), false);   // Q2:211 RoPE
          else KCHK(glc_launch_rope_qk(st, dt, e->QKV, e->ropes[Sp], M, Sp, nq, nkv, d, qscale), false); } }
        { Prof p(e, PC_ATTN);
          if (mxa) KCHK(glc_launch_attention_gqa_mx(st, e->Qh, e->Kh, e->Vt, e->kbias, e->klen, e->kfirst, e->CTX, B, Sp, nq, nkv, d, c.causal), false);
          else if (mfma) KCHK(glc_launch_attention_gqa_mfma(st, dt, e->Qh, e->Kh, e->Vt, e->kbias, e->klen, e->kfirst, e->CTX, B, Sp, nq, nkv, d, c.causal, mx ? 2 : (gs ? 1 : 0)), false);
          else KCHK(glc_launch_attention_gqa(st, dt, 1, e->QKV, e->kbias, e->klen, e->CTX, B, Sp, nq, nkv, d, c.causal), false); }
        GemmArgs o;
        o.A = e->CTX; o.W = w.Wo; o.bias = nullptr; o.C = Xn; o.resid = X; o.Mpad = Mpad; o.N = H; o.K = NQ; o.gs_resid_plain = 1;
        if (rnf) { o.gs_resid_plain = 0; o.ln_part = e->ln_part; }        // raw group-split residual in, raw group-split sum + partials out
        if (mx) { o.W = w.Wo_x; o.mx_ws = w.ws_o; }
        { Prof p(e, PC_ATTN_OUT); KCHK(gs ? gemm_gs(EPI_RESID, o) : launch_gemm_auto(e, dt, EPI_RESID, o), false); }   // Q2:233, :291
        std::swap(X, Xn); std::swap(sX, sXn);
        { Prof p(e, PC_LN); KCHK(rnf ? glc_launch_ln_stats(st, e->ln_part, H / 64, sX, M, H, c.ln_eps, 1)
                                 : gs ? glc_launch_rmsnorm_gs(st, (const float*)X, e->H1, w.ln2, c.ln_eps, M, H)
                                      : glc_launch_rmsnorm(st, dt, X, e->H1, w.ln2, c.ln_eps, M, H), false); }                        // Q2:295
        const bool need_gu = !(e->fused_swiglu && (gs || dt != GLC_F32));
        if (need_gu && e->capGU < Mpad) {
            dfree(e, e->GU); e->GU = dmalloc(e, (size_t)Mpad * 2 * I * es);
            if (!e->GU) return false;
            e->capGU = Mpad;
        }
        GemmArgs f1;
        f1.A = e->H1; f1.W = w.Wgu; f1.bias = nullptr; f1.C = e->GU; f1.Mpad = Mpad; f1.N = 2 * I; f1.K = H;
        if (rnf) { f1.A = X; f1.W = w.Wguf; f1.a_stats = sX; }
        if (mx) { f1.W = w.Wguf_x; f1.mx_ws = w.ws_guf; }
        { Prof p(e, PC_FFN1);                                                                                                  // Q2:47 silu(gate) * up
          if (e->fused_swiglu && gs) { f1.C = e->FF; KCHK(gemm_gs(EPI_SWIGLU, f1), false); }
          else if (e->fused_swiglu && dt == GLC_F32) {      // small forward of the fp32 mode: plain rows, interleaved [gate | up] columns
              KCHK(launch_gemm_auto(e, dt, EPI_BIAS, f1), false); KCHK(glc_launch_swiglu(st, dt, e->GU, e->FF, (size_t)M, I, 1), false); }
          else if (e->fused_swiglu) { f1.C = e->FF; KCHK(glc_launch_gemm256s(st, dt, EPI_SWIGLU, f1), false); }
          else if (gs) { f1.gs_c_plain = 1; KCHK(glc_launch_gemm256s_gs(st, EPI_BIAS, f1), false); KCHK(glc_launch_swiglu_gs(st, (const float*)e->GU, e->FF, (size_t)M, I), false); }
          else { KCHK(launch_gemm_auto(e, dt, EPI_BIAS, f1), false); KCHK(glc_launch_swiglu(st, dt, e->GU, e->FF, (size_t)M, I), false); } }
        GemmArgs f2;
        f2.A = e->FF; f2.W = w.Wd; f2.bias = nullptr; f2.C = Xn; f2.resid = X; f2.Mpad = Mpad; f2.N = H; f2.K = I; f2.gs_resid_plain = 1;
        if (rnf) { f2.gs_resid_plain = 0; if (l + 1 < L) f2.ln_part = e->ln_part; }      // (the last layer hands plain fp32 rows to the final norm)
        if (mx) { f2.W = w.Wd_x; f2.mx_ws = w.ws_d; }
        { Prof p(e, PC_FFN2); KCHK(gs ? gemm_gs(EPI_RESID, f2) : launch_gemm_auto(e, dt, EPI_RESID, f2), false); }
        std::swap(X, Xn); std::swap(sX, sXn);
        if (rnf && l + 1 < L) { Prof p(e, PC_LN); KCHK(glc_launch_ln_stats(st, e->ln_part, H / 64, sX, M, H, c.ln_eps, 1), false); }
        if (e->keep_hidden && l + 1 < L)
            HIPCHK(hipMemcpyAsync((char*)e->hidden_dump + (size_t)(l + 1) * M * H * es, X, (size_t)M * H * es, hipMemcpyDeviceToDevice, st), false);
    }
    { Prof p(e, PC_LN); KCHK(glc_launch_rmsnorm(st, dt, X, e->H1, e->final_norm, c.ln_eps, M, H), false); }                    // Q2:398
    if (e->keep_hidden) HIPCHK(hipMemcpyAsync((char*)e->hidden_dump + (size_t)L * M * H * es, e->H1, (size_t)M * H * es, hipMemcpyDeviceToDevice, st), false);
    if (C > 0) {
        Prof p(e, PC_HEAD);
        float* Gc = e->Gt + (size_t)round_up(B, 128) * H;
        KCHK(glc_launch_head_gather(st, dt, e->H1, e->cls_pos, ccap, e->Gt, Gc, B, Sp, H, C, c.pooling == GLC_POOL_LAST ? e->klen : nullptr), false);
        if (c.pooling == GLC_POOL_AVG) KCHK(glc_launch_pool_avg(st, dt, e->H1, e->kbias, e->Gt, B, Sp, H), false);
        if (!run_head_tail(e, B, C, d_logits)) return false;
    }
    HIPCHK(hipGetLastError(), false);
    e->lastB = B; e->lastS = S; e->lastSp = Sp;
    return true;
}

// The launch sequence for one batch.  ids/mask are device pointers.
bool run_forward(glc_engine* e, const int64_t* ids, const int64_t* mask, int B, int S, int C, float* d_logits) {
    if (e->cfg.backbone == GLC_BACKBONE_DECODER) return run_forward_decoder(e, ids, mask, B, S, C, d_logits);
    const glc_model_config& c = e->cfg;
    const int H = c.hidden, I = c.inter, nh = c.heads;
    const int Sp = round_up(S, 64), M = B * Sp, Mpad = round_up(M, 256);
    hipStream_t st = e->stream;
    const int dt = e->dtype;
    const size_t es = esize(dt);
    const int ccap = e->capC > 0 ? e->capC : 1;
    if (e->profile) { e->ev_used = 0; }
    GxScope gx_scope(e);                     // fp8 range guard + activation exponent (see run_forward_decoder)

    if (e->keep_hidden) {
        const size_t need = (size_t)(c.layers + 1) * M * H * es;
        if (need > e->hidden_cap) { dfree(e, e->hidden_dump); e->hidden_dump = dmalloc(e, need); if (!e->hidden_dump) return false; e->hidden_cap = need; }
    }
    { Prof p(e, PC_SCAN);
      KCHK(glc_launch_scan_rows(st, ids, mask, B, S, c.class_token_index, c.embed_class_token, e->klen, e->kfirst, e->cls_pos, e->cls_cnt, ccap), false); }
    // attention kernel: 3 = workgroup-shared band kernel (attention_wg.hip), 2 = per-wave band kernel (every operand type; fp32 native:
    // 32x32x2 MFMAs), 1 = straightforward kernel.  Default, measured same-box at c3 (scripts/attn_bench.py): split-f16 units of the
    // fp32 mode -> 3 (1.45 vs 1.80 ms per launch: MFMA-heavy, the 3 shared p2c MFMA blocks and the shared K / V^T ring pay);
    // 16-bit operands -> 2 (0.70 vs 0.76 ms: at one MFMA per product the kernel is bound by instruction issue and latency, and
    // the workgroup barrier per tile costs more than the 3 MFMAs and 3 loads it saves).  GLC_ATTN_WG=1 / 0 forces one or the other.
    static const int wg_env = glc_dev_env("GLC_ATTN_WG") ? atoi(glc_dev_env("GLC_ATTN_WG")) : -1;      // developer A/B switch
    const bool wg_pick = wg_env >= 0 ? wg_env != 0 : dt == GLC_F32;
    int impl = e->attn_impl ? e->attn_impl : ((wg_pick && (dt != GLC_F32 || e->attn_split)) ? 3 : 2);
    if (impl == 3 && dt == GLC_F32 && !e->attn_split) impl = 2;
    const bool asplit = e->attn_split && impl >= 2 && dt == GLC_F32;        // fp32 mode: split-f16 operand units for the band kernels
    auto launch_band = [&](const AttnArgs& aa) -> const char* {
        return impl == 3 ? glc_launch_attention_wg(st, dt, aa) : glc_launch_attention(st, dt, impl, aa);
    };
    const bool prune = e->prune_last && !e->keep_hidden && c.pooling == GLC_POOL_FIRST;
    // Group-split pipeline of the fp32 mode: the activations that feed GEMMs (X, H1, FF, CTX) are kept as [32 hi | 32 lo] f16 groups
    // (same bytes as fp32), written by their producers, so that every projection runs on the 256-tile LDS-DMA kernel with three
    // f16 MFMAs per product (gemm256s.hip, GS).  Needs the split-f16 weights and attention, the pruned last layer (its compact rows go back to plain
    // fp32 and the small-M kernels) and shapes the 256-tile kernel takes; small forwards stay on the 128-tile split-K kernels.
    bool gs = false;
    if (dt == GLC_F32 && e->gs_mode > 0 && e->w_presplit && asplit && impl == 3 && prune && H % 256 == 0 && I % 256 == 0) {
        GemmArgs t; t.Mpad = Mpad; t.N = H; t.K = H;
        gs = e->gs_mode == 2 || !glc_gemm_small_m(t);
    }
    e->last_gs = gs;
    // MX pipeline (round 3): the group-split pipeline with GX rows and the MX cross-term GEMM (gemm256x.hip) for every projection of the
    // full layers — a_hi*w_hi as f16 MFMAs, both cross terms as one block-scaled fp8 MFMA.  Needs the LayerNorm fold on every layer.
    bool mx = gs && e->mx && e->mx_built && e->ln_fused && e->prec_mask == 0 && c.layers >= 2;
    for (int l = 0; mx && l < c.layers; ++l) mx = e->layers[l].W1f && (l == 0 || e->layers[l].Wqkvf);       // (the folded split-f16 copies the GX copies are made from)
    if (mx && !build_mx_weights(e)) mx = false;          // (no copies: this and every later forward stay on the split-f16 kernels)
    for (int l = 0; mx && l < c.layers; ++l) mx = e->layers[l].W1f_x && e->layers[l].Wqkv_x && e->layers[l].Wo_x && e->layers[l].W2_x && (l == 0 || e->layers[l].Wqkvf_x);
    e->last_mx = mx;
    e->last_mx_attn = false;
    auto gemm_gs = [&](int epi, const GemmArgs& ga) -> const char* {
        if (!mx) return glc_launch_gemm256s_gs(st, epi, ga);
        GemmArgs gx = ga; gx.gx_rows = M;          // fp8 range guard: the M rows of this forward, not the slack rows up to Mpad
        return glc_launch_gemm256x(st, epi, gx);
    };
    // 16-bit modes: the same LayerNorm fold on plain rows of T, when all four projections of a layer run on the staggered 256-tile kernel
    bool fold16 = false;
    if (dt != GLC_F32 && e->ln_fused && prune && !e->keep_hidden && e->attn_impl != 1 && H % 256 == 0 && I % 256 == 0 && e->statsA && e->statsB && e->ln_part) {
        GemmArgs t; t.Mpad = Mpad; t.N = H; t.K = H;
        fold16 = glc_gemm256_supported(dt, t) && !glc_gemm_small_m(t);
    }
    { Prof p(e, PC_EMBED);
      if (gs) KCHK(glc_launch_embed_gs(st, ids, mask, (const float*)e->emb, e->eln_g, e->eln_b, c.ln_eps, e->X, e->kbias, B, S, Sp, H, c.vocab, c.pad_id, mx ? 1 : 0), false);
      else KCHK(glc_launch_embed(st, dt, ids, mask, e->emb, e->eln_g, e->eln_b, c.ln_eps, e->X, e->kbias, B, S, Sp, H, c.vocab, c.pad_id), false); }
    if (e->keep_hidden) HIPCHK(hipMemcpyAsync(e->hidden_dump, e->X, (size_t)M * H * es, hipMemcpyDeviceToDevice, st), false);

    bool x_is_raw = false;
    for (int l = 0; l < c.layers; ++l) {
        const LayerW& w = e->layers[l];
        const bool last = prune && l == c.layers - 1;
        // LayerNorm folded away (group-split pipeline, e->ln_fused): X / H1 then hold the RAW residual sums of the producer GEMMs plus
        // (mean, rstd) per row in statsA / statsB; the consumers run on weights with gamma folded in.  Normalised rows stay where a
        // kernel outside the pipeline reads them: the embedding output (layer 0) and the input of the pruned last layer.
        const bool lnf = (gs || fold16) && e->ln_fused && w.W1f && (l == 0 || w.Wqkvf);
        if (l == 0) e->last_lnf = lnf;
        const bool x_raw = x_is_raw;         // X holds raw sums + statsA (written by the previous layer's FFN2)
        GemmArgs g;
        g.A = e->X; g.W = w.Wqkv; g.bias = w.bqkv; g.Qh = e->Qh; g.Kh = e->Kh; g.Vt = e->Vt;
        g.Mpad = Mpad; g.N = 3 * H; g.K = H; g.Mvalid = M; g.Sp = Sp; g.nh = nh; g.H = H; g.qkv_split = asplit;
        if (last) break;
        if (x_raw) { g.W = w.Wqkvf; g.bias = w.dq; g.a_stats = e->statsA; g.ln_c = w.cq; }
        const bool mxa = mx && e->mx_attn && w.PKm && w.PQm;       // attention of this layer on MX tiles
        if (mx) { g.W = x_raw ? w.Wqkvf_x : w.Wqkv_x; g.mx_ws = x_raw ? w.ws_qkvf : w.ws_qkv; g.qkv_mxt = mxa ? 1 : 0; }
        const int pm = gs ? e->prec_mask : 0;
        g.prec = pm & 3;
        { Prof p(e, PC_QKV); KCHK(gs ? gemm_gs(EPI_QKV, g) : launch_gemm_auto(e, dt, EPI_QKV, g), false); }
        if (e->debug_stop >= 10 * l && e->debug_stop <= 10 * l + 4) { e->lastB = B; e->lastS = S; e->lastSp = Sp; }      // (a stopped forward still describes the workspace it leaves)
        if (e->debug_stop == 10 * l + 0) return true;
        AttnArgs a{e->Qh, e->Kh, e->Vt, asplit ? w.PKs : w.PK, asplit ? w.PQs : w.PQ, e->dtabs[Sp], e->kbias, e->klen, e->kfirst, e->CTX, B, nh, Sp, H, e->P};
        a.split = asplit; a.ctx_gs = mx ? 2 : (gs ? 1 : 0); a.prec = (pm >> 8) & 63;
        static const bool nosat = glc_dev_env("GLC_ATTN_NOSAT") != nullptr;      // A/B switch (developer)
        if (!nosat) { a.rsat_pos = e->dsat[Sp].first; a.rsat_neg = e->dsat[Sp].second; }
        a.otab = e->otabs[Sp]; a.mtab = e->mtabs.count(Sp) ? e->mtabs[Sp] : nullptr;
        if (mxa) { a.PK = w.PKm; a.PQ = w.PQm; e->last_mx_attn = true; a.idx16 = e->mx2tabs[Sp].first; a.tinfo = e->mx2tabs[Sp].second; }
        const bool mxa2 = mxa && e->mx2 && a.idx16 && a.tinfo;      // bucket-space kernel (round 4) when this length's table has the structure it needs
        { Prof p(e, PC_ATTN); KCHK(mxa2 ? glc_launch_attention_mx2(st, a) : mxa ? glc_launch_attention_mx(st, a) : launch_band(a), false); }
        if (e->debug_stop == 10 * l + 1) return true;
        GemmArgs o;
        o.A = e->CTX; o.W = w.Wo; o.bias = w.bo; o.C = e->T1; o.resid = e->X; o.Mpad = Mpad; o.N = H; o.K = H;
        if (x_raw) { const LayerW& wp = e->layers[l - 1]; o.r_stats = e->statsA; o.r_gamma = wp.ln2g; o.r_beta = wp.ln2b; }
        if (lnf) { o.C = e->H1; o.ln_part = e->ln_part; }      // raw sum -> H1 (group-split rows) + partials
        o.prec = ((pm >> 2) & 3) | ((pm & PM_RESID) ? 4 : 0);
        if (mx) { o.W = w.Wo_x; o.mx_ws = w.ws_o; }
        { Prof p(e, PC_ATTN_OUT); KCHK(gs ? gemm_gs(EPI_RESID, o) : launch_gemm_auto(e, dt, EPI_RESID, o), false); }
        { Prof p(e, PC_LN); KCHK(lnf ? glc_launch_ln_stats(st, e->ln_part, H / 64, e->statsB, M, H, c.ln_eps)
                                 : gs ? glc_launch_layernorm_gs(st, (const float*)e->T1, e->H1, w.ln1g, w.ln1b, c.ln_eps, M, H)
                                      : glc_launch_layernorm(st, dt, e->T1, e->H1, w.ln1g, w.ln1b, c.ln_eps, M, H), false); }
        if (e->debug_stop == 10 * l + 2) return true;
        GemmArgs f1;
        f1.A = e->H1; f1.W = w.W1; f1.bias = w.b1; f1.C = e->FF; f1.Mpad = Mpad; f1.N = I; f1.K = H;
        if (lnf) { f1.W = w.W1f; f1.bias = w.d1; f1.a_stats = e->statsB; f1.ln_c = w.c1; }
        f1.prec = (pm >> 4) & 3;
        if (mx) { f1.W = w.W1f_x; f1.mx_ws = w.ws_1f; }
        { Prof p(e, PC_FFN1); KCHK(gs ? gemm_gs(EPI_GELU, f1) : launch_gemm_auto(e, dt, EPI_GELU, f1), false); }
        if (e->debug_stop == 10 * l + 3) return true;
        GemmArgs f2;
        f2.A = e->FF; f2.W = w.W2; f2.bias = w.b2; f2.C = e->T1; f2.resid = e->H1; f2.Mpad = Mpad; f2.N = H; f2.K = I;
        if (lnf) { f2.r_stats = e->statsB; f2.r_gamma = w.ln1g; f2.r_beta = w.ln1b; }
        f2.prec = ((pm >> 6) & 3) | ((pm & PM_RESID) ? 4 : 0);
        // the next consumer of X takes raw rows only if it is a folded QKV of the pipeline (not the pruned last layer, which gathers normalised rows)
        const bool next_raw = lnf && l + 1 < c.layers && e->layers[l + 1].Wqkvf && !(prune && l + 1 == c.layers - 1);
        if (next_raw) { f2.C = e->X; f2.ln_part = e->ln_part; }
        if (mx) { f2.W = w.W2_x; f2.mx_ws = w.ws_2; }
        { Prof p(e, PC_FFN2); KCHK(gs ? gemm_gs(EPI_RESID, f2) : launch_gemm_auto(e, dt, EPI_RESID, f2), false); }
        { Prof p(e, PC_LN); KCHK(next_raw ? glc_launch_ln_stats(st, e->ln_part, H / 64, e->statsA, M, H, c.ln_eps)
                                 : gs ? glc_launch_layernorm_gs(st, (const float*)e->T1, e->X, w.ln2g, w.ln2b, c.ln_eps, M, H, mx ? 1 : 0)
                                      : glc_launch_layernorm(st, dt, e->T1, e->X, w.ln2g, w.ln2b, c.ln_eps, M, H), false); }
        x_is_raw = next_raw;
        if (e->debug_stop == 10 * l + 4) return true;
        if (e->keep_hidden)
            HIPCHK(hipMemcpyAsync((char*)e->hidden_dump + (size_t)(l + 1) * M * H * es, e->X, (size_t)M * H * es, hipMemcpyDeviceToDevice, st), false);
    }
    const int Cc = C > 0 ? C : 0;
    if (prune) {
        // Last layer, exact pruning: the head reads only the [CLS] row and the class-token rows, so only those
        // R = B*(1+C) rows need Q, attention output, the output projection and the FFN.  K and V^T are still
        // produced for every position (they are what the selected queries attend to).
        Prof p(e, PC_LAST);
        const LayerW& w = e->layers[c.layers - 1];
        const int R = B * (1 + Cc), Rpad = round_up(R, 256);
        const bool band_sel = impl >= 2;     // band kernel on the query tiles that hold selected rows; impl 1: row-selection kernel
        GemmArgs g;
        g.A = e->X; g.W = w.Wqkv; g.bias = w.bqkv; g.Qh = e->Qh; g.Kh = e->Kh; g.Vt = e->Vt; g.qkv_skip_q = band_sel ? 0 : 1;
        g.Mpad = Mpad; g.N = 3 * H; g.K = H; g.Mvalid = M; g.Sp = Sp; g.nh = nh; g.H = H; g.qkv_split = asplit && band_sel;
        if (band_sel) HIPCHK(hipMemsetAsync(e->tile_flag, 0, (size_t)Mpad >> 5, st), false);
        // (group-split pipeline: the compact rows leave it here — plain fp32, the rest of this layer runs on the small-M kernels)
        if (gs) KCHK(glc_launch_gather_rows_gs(st, e->X, e->cls_pos, ccap, (float*)e->Xs, e->sel_b, e->sel_q, band_sel ? e->tile_flag : nullptr, B, Sp, H, Cc, mx ? 1 : 0), false);
        else KCHK(glc_launch_gather_rows(st, dt, e->X, e->cls_pos, ccap, e->Xs, e->sel_b, e->sel_q, band_sel ? e->tile_flag : nullptr, B, Sp, H, Cc), false);
        if (gs) g.prec = e->prec_mask & 3;
        if (mx) { g.W = w.Wqkv_x; g.mx_ws = w.ws_qkv; }
        if (band_sel) g.q_tile_flag = e->tile_flag;       // the gather has flagged the query tiles that hold selected rows: the Q third skips the others
        KCHK(gs ? gemm_gs(EPI_QKV, g) : launch_gemm_auto(e, dt, EPI_QKV, g), false);
        if (band_sel) {
            AttnArgs a{e->Qh, e->Kh, e->Vt, asplit ? w.PKs : w.PK, asplit ? w.PQs : w.PQ, e->dtabs[Sp], e->kbias, e->klen, e->kfirst, e->CTX, B, nh, Sp, H, e->P};
            a.rsat_pos = e->dsat[Sp].first; a.rsat_neg = e->dsat[Sp].second; a.tile_flag = e->tile_flag; a.otab = e->otabs[Sp]; a.split = asplit;
            // selected query tiles only: the per-wave kernel skips every other tile; a workgroup of the shared kernel would run all
            // its waves for the one tile that holds the [CLS] / class-token rows
            static const bool ksplit_on = !(glc_dev_env("GLC_ATTN_KSPLIT") && atoi(glc_dev_env("GLC_ATTN_KSPLIT")) == 0);     // developer A/B switch
            a.ksplit = ksplit_on ? 1 : 0;       // a workgroup with one selected query tile splits that tile's keys over its four waves
            KCHK(e->attn_impl == 3 ? launch_band(a) : glc_launch_attention(st, dt, 2, a), false);
            KCHK(glc_launch_gather_sel(st, dt, e->CTX, e->sel_b, e->sel_q, e->CTXs, R, Sp, H), false);
        } else {
            GemmArgs gq;     // Q rows of the selection: the first H rows of the fused [3H,H] weight are the (pre-scaled) query projection
            gq.A = e->Xs; gq.W = w.Wqkv; gq.bias = w.bqkv; gq.C = e->Qs; gq.Mpad = Rpad; gq.N = H; gq.K = H;
            KCHK(launch_gemm_auto(e, dt, EPI_BIAS, gq), false);
            AttnArgs a{nullptr, e->Kh, e->Vt, w.PK, w.PQ, e->dtabs[Sp], e->kbias, e->klen, e->kfirst, e->CTXs, B, nh, Sp, H, e->P};
            a.sel_b = e->sel_b; a.sel_q = e->sel_q; a.Qrow = e->Qs; a.nsel = R;
            KCHK(glc_launch_attention(st, dt, 1, a), false);
        }
        GemmArgs o;
        o.A = e->CTXs; o.W = w.Wo; o.bias = w.bo; o.C = e->T1s; o.resid = e->Xs; o.Mpad = Rpad; o.N = H; o.K = H;
        KCHK(launch_gemm_auto(e, dt, EPI_RESID, o), false);
        KCHK(glc_launch_layernorm(st, dt, e->T1s, e->H1s, w.ln1g, w.ln1b, c.ln_eps, R, H), false);
        GemmArgs f1;
        f1.A = e->H1s; f1.W = w.W1; f1.bias = w.b1; f1.C = e->FFs; f1.Mpad = Rpad; f1.N = I; f1.K = H;
        KCHK(launch_gemm_auto(e, dt, EPI_GELU, f1), false);
        GemmArgs f2;
        f2.A = e->FFs; f2.W = w.W2; f2.bias = w.b2; f2.C = e->T1s; f2.resid = e->H1s; f2.Mpad = Rpad; f2.N = H; f2.K = I;
        KCHK(launch_gemm_auto(e, dt, EPI_RESID, f2), false);
        KCHK(glc_launch_layernorm(st, dt, e->T1s, e->Xs, w.ln2g, w.ln2b, c.ln_eps, R, H), false);
    }
    if (C > 0) {
        Prof p(e, PC_HEAD);
        float* Gc = e->Gt + (size_t)round_up(B, 128) * H;
        if (prune) KCHK(glc_launch_head_gather_sel(st, dt, e->Xs, e->cls_pos, ccap, e->Gt, Gc, B, H, C), false);
        else {
            KCHK(glc_launch_head_gather(st, dt, e->X, e->cls_pos, ccap, e->Gt, Gc, B, Sp, H, C, c.pooling == GLC_POOL_LAST ? e->klen : nullptr), false);
            if (c.pooling == GLC_POOL_AVG) KCHK(glc_launch_pool_avg(st, dt, e->X, e->kbias, e->Gt, B, Sp, H), false);
        }
        if (!run_head_tail(e, B, C, d_logits)) return false;
    }
    HIPCHK(hipGetLastError(), false);
    e->lastB = B; e->lastS = S; e->lastSp = Sp;
    return true;
}

// Decoder backbone: upload + convert weights.  Fused operands: Wqkv = [q_proj; k_proj; v_proj] rows, Wgu = [gate_proj; up_proj]
// rows, so each layer runs four GEMMs (Q2:206-208 as one, Q2:233, Q2:47 gate|up as one, down).
bool create_decoder(glc_engine* e, const float* const* tensors) {
    const glc_model_config& c = e->cfg;
    const int H = c.hidden, I = c.inter, L = c.layers, d = c.head_dim;
    const size_t NQ = (size_t)c.heads * d, NKV = (size_t)c.kv_heads * d, NQKV = NQ + 2 * NKV;
    const size_t es = esize(e->dtype);
    if (NQ % 128 || NKV % 64 || NQKV % 128) { set_err("engine_create: decoder projection widths must be multiples of 128"); return false; }
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { set_err("stream create failed"); return false; }
    if (hipEventCreate(&e->t0) != hipSuccess || hipEventCreate(&e->t1) != hipSuccess) { set_err("event create failed"); return false; }
    size_t stage_n = (size_t)c.vocab * H;
    if ((size_t)I * H > stage_n) stage_n = (size_t)I * H;
    if (NQ * H > stage_n) stage_n = NQ * H;
    if (2 * (size_t)I * H > stage_n) stage_n = 2 * (size_t)I * H;
    float* staging = nullptr;
    if (hipMalloc((void**)&staging, stage_n * sizeof(float)) != hipSuccess) { set_err("staging alloc failed"); return false; }
    bool ok = false;
    do {
        e->emb = dmalloc(e, (size_t)c.vocab * H * es, false);
        if (!e->emb || !upload_as(e, tensors[0], (size_t)c.vocab * H, e->emb, staging)) break;
        e->dlayers.resize(L);
        // SwiGLU in the epilogue of the staggered 256-tile GEMM when the shapes allow it (16-bit operands)
        // (fp32 mode: the group-split 256-tile GEMM has the same epilogue; its small-forward fallback un-fuses on the interleaved columns)
        e->fused_swiglu = (e->dtype != GLC_F32 || (e->w_presplit && e->dec_split && H % 256 == 0)) && (2 * I) % 256 == 0 && H % 32 == 0 && I % 16 == 0 &&
                          glc_dev_env("GLC_NO_FUSED_SWIGLU") == nullptr;
        std::vector<float> bqkv(NQKV), gu_host(e->fused_swiglu ? 2 * (size_t)I * H : 0), fold_host;
        bool lok = true;
        for (int l = 0; l < L && lok; ++l) {
            const float* const* t = tensors + 1 + GLC_DEC_TENSORS_PER_LAYER * l;    // ln1 qw qb kw kb vw vb ow ln2 gw uw dw
            DecLayerW& w = e->dlayers[l];
            w.Wqkv = dmalloc(e, NQKV * H * es, false);
            w.Wo = dmalloc(e, (size_t)H * NQ * es, false);
            w.Wgu = dmalloc(e, 2 * (size_t)I * H * es, false);
            w.Wd = dmalloc(e, (size_t)H * I * es, false);
            if (!w.Wqkv || !w.Wo || !w.Wgu || !w.Wd) { lok = false; break; }
            lok = upload_as(e, t[1], NQ * H, w.Wqkv, staging) && upload_as(e, t[3], NKV * H, (char*)w.Wqkv + NQ * H * es, staging) &&
                  upload_as(e, t[5], NKV * H, (char*)w.Wqkv + (NQ + NKV) * H * es, staging) && upload_as(e, t[7], (size_t)H * NQ, w.Wo, staging) &&
                  upload_as(e, t[11], (size_t)H * I, w.Wd, staging);
            if (lok && e->fused_swiglu) {
                // rows [32 f, 32 f + 16) = gate rows of features 16 f .. 16 f + 15, rows [32 f + 16, 32 f + 32) = their up rows
                for (int f = 0; f < I / 16; ++f) {
                    memcpy(gu_host.data() + (size_t)(32 * f) * H, t[9] + (size_t)(16 * f) * H, (size_t)16 * H * sizeof(float));
                    memcpy(gu_host.data() + (size_t)(32 * f + 16) * H, t[10] + (size_t)(16 * f) * H, (size_t)16 * H * sizeof(float));
                }
                lok = upload_as(e, gu_host.data(), 2 * (size_t)I * H, w.Wgu, staging);
            } else if (lok) {
                lok = upload_as(e, t[9], (size_t)I * H, w.Wgu, staging) && upload_as(e, t[10], (size_t)I * H, (char*)w.Wgu + (size_t)I * H * es, staging);
            }
            if (!lok) break;
            if (e->w_presplit && e->dtype == GLC_F32) {     // group-split weight rows (the operand image of both split-f16 GEMM kernels)
                const char* pm = glc_launch_presplit(e->stream, w.Wqkv, NQKV * H);
                if (!pm) pm = glc_launch_presplit(e->stream, w.Wo, (size_t)H * NQ);
                if (!pm) pm = glc_launch_presplit(e->stream, w.Wgu, 2 * (size_t)I * H);
                if (!pm) pm = glc_launch_presplit(e->stream, w.Wd, (size_t)H * I);
                if (pm) { set_err(pm); lok = false; break; }
            }
            if (e->ln_fused && e->w_presplit && e->dtype == GLC_F32 && e->fused_swiglu && H % 64 == 0) {
                // RMSNorm folded into the consumer GEMMs of the group-split pipeline: RMSNorm(x) W^T = rstd (x (W diag(gamma))^T)
                auto fold_rows = [&](const float* Wsrc, size_t rows, const float* gam, float* dst) {
                    for (size_t n = 0; n < rows; ++n) for (int k = 0; k < H; ++k) dst[n * H + k] = Wsrc[n * H + k] * gam[k];
                };
                std::vector<float>& wf = fold_host;
                wf.resize(2 * (size_t)I * H > NQKV * H ? 2 * (size_t)I * H : NQKV * H);
                fold_rows(t[1], NQ, t[0], wf.data()); fold_rows(t[3], NKV, t[0], wf.data() + NQ * H); fold_rows(t[5], NKV, t[0], wf.data() + (NQ + NKV) * H);
                w.Wqkvf = dmalloc(e, NQKV * H * es, false);
                if (!w.Wqkvf || !upload_as(e, wf.data(), NQKV * H, w.Wqkvf, staging)) { lok = false; break; }
                const char* pm = glc_launch_presplit(e->stream, w.Wqkvf, NQKV * H);
                if (pm) { set_err(pm); lok = false; break; }
                fold_rows(gu_host.data(), 2 * (size_t)I, t[8], wf.data());           // the interleaved [16 gate | 16 up] row order is kept
                w.Wguf = dmalloc(e, 2 * (size_t)I * H * es, false);
                if (!w.Wguf || !upload_as(e, wf.data(), 2 * (size_t)I * H, w.Wguf, staging)) { lok = false; break; }
                pm = glc_launch_presplit(e->stream, w.Wguf, 2 * (size_t)I * H);
                if (pm) { set_err(pm); lok = false; break; }
            }
            for (size_t i = 0; i < NQ; ++i) bqkv[i] = t[2][i];
            for (size_t i = 0; i < NKV; ++i) { bqkv[NQ + i] = t[4][i]; bqkv[NQ + NKV + i] = t[6][i]; }
            w.bqkv = upload_f32(e, bqkv.data(), NQKV);
            w.ln1 = upload_f32(e, t[0], H); w.ln2 = upload_f32(e, t[8], H);
            if (!w.bqkv || !w.ln1 || !w.ln2) { lok = false; break; }
            if (hipStreamSynchronize(e->stream) != hipSuccess) { set_err("sync failed"); lok = false; break; }   // bqkv host buffer is reused
        }
        if (!lok) break;
        e->final_norm = upload_f32(e, tensors[1 + GLC_DEC_TENSORS_PER_LAYER * L], H);
        if (!e->final_norm) break;
        const float* const* ht = tensors + 2 + GLC_DEC_TENSORS_PER_LAYER * L;
        bool hok = true;
        for (int i = 0; i < 8 && hok; ++i) {
            e->headw[i] = upload_f32(e, ht[i], (i % 2 == 0) ? (size_t)H * H : (size_t)H); hok = e->headw[i] != nullptr;
            if (hok && (i % 2 == 0) && e->w_presplit) { const char* pm = glc_launch_presplit(e->stream, e->headw[i], (size_t)H * H); if (pm) { set_err(pm); hok = false; } }
        }
        hok = hok && upload_scorer(e, ht + GLC_TENSORS_HEAD);
        if (!hok) break;
        if (hipStreamSynchronize(e->stream) != hipSuccess) { set_err(std::string("engine_create: ") + hipGetErrorString(hipGetLastError())); break; }
        ok = true;
    } while (0);
    (void)hipFree(staging);
    return ok;
}

// the value a 16-bit MFMA operand carries for f (host side of glc_launch_convert)
inline float round_as_operand(float f, int dtype) {
    if (dtype == GLC_F16) return (float)(_Float16)f;
    if (dtype == GLC_BF16) return (float)(__bf16)f;
    return f;
}

bool check_shape(const glc_engine* e, int B, int S, int C) {
    if (B <= 0 || S <= 0 || C < 0) { set_err("forward: B and S must be positive, C non-negative"); return false; }
    if ((long long)B * round_up(S, 64) > (1ll << 30)) { set_err("forward: batch too large"); return false; }
    (void)e;
    return true;
}

}  // namespace

extern "C" {

const char* glc_last_error(void) { return g_err.c_str(); }

int glc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* modeling_deberta_v2.py:57-69 (make_log_bucket_position) + :318 clamp; float32 arithmetic like torch */
void glc_delta_table(int S, int bucket_size, int max_position, int32_t* out) {
    const int span = bucket_size > 0 ? bucket_size : max_position;
    const int mid = bucket_size / 2;
    const float den = (bucket_size > 0 && max_position > 0) ? logf((float)(((double)max_position - 1.0) / (double)mid)) : 1.f;
    for (int r = -(S - 1); r <= S - 1; ++r) {
        int bk = r;
        if (bucket_size > 0 && max_position > 0) {
            const int ar = r < 0 ? -r : r;
            if (ar > mid) {
                const float lp = ceilf(logf((float)ar / (float)mid) / den * (float)(mid - 1)) + (float)mid;
                bk = (int)(r < 0 ? -lp : lp);
            }
        }
        int v = bk + span;
        out[r + S - 1] = v < 0 ? 0 : (v > 2 * span - 1 ? 2 * span - 1 : v);
    }
}

glc_engine* glc_engine_create(const glc_model_config* cfg, const float* const* tensors, int n_tensors, int device, int dtype) {
    if (!cfg || !tensors) { set_err("engine_create: null argument"); return nullptr; }
    if (dtype != GLC_F32 && dtype != GLC_BF16 && dtype != GLC_F16) { set_err("engine_create: bad dtype"); return nullptr; }
    if (n_tensors != glc_num_tensors_cfg(cfg)) { set_err("engine_create: wrong tensor count"); return nullptr; }
    for (int i = 0; i < n_tensors; ++i) if (!tensors[i]) { set_err("engine_create: null tensor"); return nullptr; }
    const bool dec = cfg->backbone == GLC_BACKBONE_DECODER;
    if (cfg->backbone != GLC_BACKBONE_DEBERTA && !dec) { set_err("engine_create: unknown backbone"); return nullptr; }
    if (!dec && (cfg->head_dim != 64 || cfg->hidden != cfg->heads * 64)) { set_err("engine_create: head_dim must be 64 (all DeBERTa-v3 backbones)"); return nullptr; }
    if (dec && ((cfg->head_dim != 64 && cfg->head_dim != 128) || cfg->heads <= 0 || cfg->kv_heads < 0 ||
                cfg->heads % (cfg->kv_heads > 0 ? cfg->kv_heads : cfg->heads) || cfg->rope_theta <= 1.f)) {
        set_err("engine_create: decoder backbone needs head_dim 64 or 128, heads % kv_heads == 0 and rope_theta > 1"); return nullptr;
    }
    if (cfg->hidden % 128 || cfg->inter % 128) { set_err("engine_create: hidden and intermediate sizes must be multiples of 128"); return nullptr; }
    if (cfg->pooling < GLC_POOL_FIRST || cfg->pooling > GLC_POOL_LAST || cfg->scorer < GLC_SCORER_DOT || cfg->scorer > GLC_SCORER_MLP) {
        set_err("engine_create: pooling must be 'first', 'avg' or 'last' and the scorer 'simple', 'weighted-dot' or 'mlp'"); return nullptr;
    }
    int ndev = glc_device_count();
    if (ndev <= 0) { set_err("engine_create: no HIP device visible (this engine has no CPU path)"); return nullptr; }
    if (device < 0 || device >= ndev) { set_err("engine_create: bad device ordinal"); return nullptr; }
    HIPCHK(hipSetDevice(device), nullptr);

    glc_engine* e = new glc_engine();
    e->cfg = *cfg; e->dtype = dtype; e->device = device;
    if (e->cfg.kv_heads <= 0) e->cfg.kv_heads = e->cfg.heads;
    if (const char* pv = getenv("GLICLASS_PRUNE_LAST")) e->prune_last = atoi(pv) != 0;
    { const char* gv = getenv("GLICLASS_F32_GEMM"); e->w_presplit = !(gv && !strcmp(gv, "native")) ; }   // hidden and inter are multiples of 128 (checked above)
    { const char* av = getenv("GLICLASS_F32_ATTN"); e->dec_split = dtype == GLC_F32 && cfg->backbone == GLC_BACKBONE_DECODER && !(av && !strcmp(av, "native")); }
    { const char* av = getenv("GLICLASS_F32_ATTN"); e->attn_split = dtype == GLC_F32 && cfg->backbone != GLC_BACKBONE_DECODER && !(av && !strcmp(av, "native")); }
    if (const char* lv = glc_dev_env("GLC_LNF")) e->ln_fused = atoi(lv) != 0;      // developer A/B switch
    // MX cross-term pipeline (docs/LOG_r01-r05.md §3e) — the default arithmetic of the large forwards of the default mode since round 3: the
    // projections of the full layers run a_hi*w_hi in f16 MFMAs and both cross terms in one block-scaled fp8 MFMA (per-label
    // probabilities within 1e-4 of the split-f16 arithmetic, measured; the bar is 1e-3).  GLICLASS_MX=0: split-f16 projections
    // everywhere (three f16 MFMAs per product, ~1e-5); GLICLASS_MX=build: GX weight copies built, pipeline off until glc_debug_set_mx.
    {
        const char* mv = getenv("GLICLASS_MX");
        const bool eligible = dec ? (dtype == GLC_F32 && e->w_presplit && e->dec_split && e->ln_fused && cfg->hidden % 256 == 0 && (2 * cfg->inter) % 256 == 0 && cfg->inter % 32 == 0)
                                  : (dtype == GLC_F32 && e->w_presplit && e->attn_split && e->ln_fused && cfg->hidden % 256 == 0 && cfg->inter % 256 == 0 && cfg->layers >= 2);
        e->mx_built = eligible && !(mv && !strcmp(mv, "0"));
        e->mx = e->mx_built && !(mv && !strcmp(mv, "build"));
        if (const char* av = glc_dev_env("GLC_MX_ATTN")) e->mx_attn = atoi(av) != 0;      // developer A/B switch
        if (const char* av = glc_dev_env("GLC_DEC_ROPE_EPI")) e->dec_rope_epi = atoi(av) != 0;      // developer A/B switch
        if (const char* av = glc_dev_env("GLC_ATTN_MX2")) e->mx2 = atoi(av) != 0;         // developer A/B switch: 1 = the bucket-space kernel (attention_mx2.hip)
    }
    if (const char* gv = glc_dev_env("GLC_GS")) { const int g = atoi(gv); e->gs_mode = g < 0 ? 0 : (g > 2 ? 2 : g); }       // developer A/B switch
    if (const char* bv = getenv("GLICLASS_LENGTH_BUCKETS")) { const int g = atoi(bv); e->max_buckets = g < 1 ? 1 : (g > 64 ? 64 : g); }
    if (dec) {
        if (!create_decoder(e, tensors)) { glc_engine_destroy(e); return nullptr; }
        return e;
    }
    const int H = cfg->hidden, I = cfg->inter, L = cfg->layers, nh = cfg->heads;
    const int span = cfg->pos_buckets > 0 ? cfg->pos_buckets : cfg->max_rel_pos;
    const int P = 2 * span;
    e->P = P;
    const size_t es = esize(dtype);
    bool ok = false;
    do {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { set_err("stream create failed"); break; }
        if (hipEventCreate(&e->t0) != hipSuccess || hipEventCreate(&e->t1) != hipSuccess) { set_err("event create failed"); break; }
        size_t stage_n = (size_t)cfg->vocab * H;
        if ((size_t)I * H > stage_n) stage_n = (size_t)I * H;
        if (3 * (size_t)H * H > stage_n) stage_n = 3 * (size_t)H * H;     // the folded fused QKV weight goes up in one piece
        float* staging = nullptr;
        if (hipMalloc((void**)&staging, stage_n * sizeof(float)) != hipSuccess) { set_err("staging alloc failed"); break; }
        auto fail = [&]() { (void)hipFree(staging); };

        // embeddings (HF:518-562)
        e->emb = dmalloc(e, (size_t)cfg->vocab * H * es, false);
        if (!e->emb || !upload_as(e, tensors[0], (size_t)cfg->vocab * H, e->emb, staging)) { fail(); break; }
        e->eln_g = upload_f32(e, tensors[1], H); e->eln_b = upload_f32(e, tensors[2], H);
        if (!e->eln_g || !e->eln_b) { fail(); break; }

        // R = LayerNorm(rel_embeddings) in fp32 (HF:595-599), then to T, padded to 128 rows
        const int Ppad = round_up(P, 256);
        float* rel_f32 = upload_f32(e, tensors[3], (size_t)P * H);
        float* rg = upload_f32(e, tensors[4], H); float* rb = upload_f32(e, tensors[5], H);
        float* Rf = (float*)dmalloc(e, (size_t)Ppad * H * sizeof(float));
        void* Rt = dmalloc(e, (size_t)Ppad * H * es);
        void* vscratch = dmalloc(e, (size_t)nh * 64 * P * es);
        if (!rel_f32 || !rg || !rb || !Rf || !Rt || !vscratch) { fail(); break; }
        const char* m = glc_launch_layernorm(e->stream, GLC_F32, rel_f32, Rf, rg, rb, cfg->ln_eps, P, H);
        if (!m) m = glc_launch_convert(e->stream, dtype, Rf, Rt, (size_t)Ppad * H);
        if (m) { set_err(m); fail(); break; }

        // HF:237-242: scores / sqrt(d * 3); times log2(e) so that the attention kernels can use exp2 directly
        const float inv_scale = 1.4426950408889634f / sqrtf((float)cfg->head_dim * 3.0f);
        std::vector<float> wq((size_t)H * H), bq(H), bqkv(3 * (size_t)H);
        e->layers.resize(L);
        bool lok = true;
        float ln_bound = 0.f;
        for (int i = 0; i < H; ++i) ln_bound = fmaxf(ln_bound, fabsf(tensors[1][i]) * sqrtf((float)H) + fabsf(tensors[2][i]));      // the embedding LayerNorm
        for (int l = 0; l < L && lok; ++l) {
            const float* const* t = tensors + GLC_TENSORS_FIXED + GLC_TENSORS_PER_LAYER * l;
            LayerW& w = e->layers[l];
            w.Wqkv = dmalloc(e, 3 * (size_t)H * H * es, false);
            w.Wo = dmalloc(e, (size_t)H * H * es, false);
            w.W1 = dmalloc(e, (size_t)I * H * es, false);
            w.W2 = dmalloc(e, (size_t)H * I * es, false);
            w.PK = dmalloc(e, (size_t)nh * P * 64 * es);
            w.PQ = dmalloc(e, (size_t)nh * P * 64 * es);
            if (!w.Wqkv || !w.Wo || !w.W1 || !w.W2 || !w.PK || !w.PQ) { lok = false; break; }
            for (size_t i = 0; i < (size_t)H * H; ++i) wq[i] = t[0][i] * inv_scale;   // fold log2(e)/sqrt(3d) into the query projection
            for (int i = 0; i < H; ++i) { bqkv[i] = t[1][i] * inv_scale; bqkv[H + i] = t[3][i]; bqkv[2 * H + i] = t[5][i]; }
            lok = upload_as(e, wq.data(), (size_t)H * H, w.Wqkv, staging) &&
                  upload_as(e, t[2], (size_t)H * H, (char*)w.Wqkv + (size_t)H * H * es, staging) &&
                  upload_as(e, t[4], (size_t)H * H, (char*)w.Wqkv + 2 * (size_t)H * H * es, staging) &&
                  upload_as(e, t[6], (size_t)H * H, w.Wo, staging) && upload_as(e, t[10], (size_t)I * H, w.W1, staging) &&
                  upload_as(e, t[12], (size_t)H * I, w.W2, staging);
            if (!lok) break;
            w.bqkv = upload_f32(e, bqkv.data(), 3 * (size_t)H);
            w.bo = upload_f32(e, t[7], H); w.ln1g = upload_f32(e, t[8], H); w.ln1b = upload_f32(e, t[9], H);
            w.b1 = upload_f32(e, t[11], I); w.b2 = upload_f32(e, t[13], H);
            w.ln2g = upload_f32(e, t[14], H); w.ln2b = upload_f32(e, t[15], H);
            for (int i = 0; i < H; ++i) {       // (proactive fp8 range guard, below: the largest element a normalised row of this layer can hold)
                ln_bound = fmaxf(ln_bound, fmaxf(fabsf(t[8][i]) * sqrtf((float)H) + fabsf(t[9][i]), fabsf(t[14][i]) * sqrtf((float)H) + fabsf(t[15][i])));
            }
            if (!w.bqkv || !w.bo || !w.ln1g || !w.ln1b || !w.b1 || !w.b2 || !w.ln2g || !w.ln2b) { lok = false; break; }
            if (hipStreamSynchronize(e->stream) != hipSuccess) { set_err("sync failed"); lok = false; break; }   // bqkv host buffer is reused
            if (e->w_presplit && dtype == GLC_F32) {
                const char* pm = glc_launch_presplit(e->stream, w.Wqkv, 3 * (size_t)H * H);
                if (!pm) pm = glc_launch_presplit(e->stream, w.Wo, (size_t)H * H);
                if (!pm) pm = glc_launch_presplit(e->stream, w.W1, (size_t)I * H);
                if (!pm) pm = glc_launch_presplit(e->stream, w.W2, (size_t)H * I);
                if (pm) { set_err(pm); lok = false; break; }
            }
            // position projections (HF:296-302, share_att_key): PQ = query_proj(R)*log2e/sqrt(3d), PK = key_proj(R)
            GemmArgs g;
            g.A = Rt; g.W = w.Wqkv; g.bias = w.bqkv; g.Qh = w.PQ; g.Kh = w.PK; g.Vt = vscratch;
            g.Mpad = Ppad; g.N = 3 * H; g.K = H; g.Mvalid = P; g.Sp = P; g.nh = nh; g.H = H;
            g.w_presplit = e->w_presplit && dtype == GLC_F32;
            const char* gm = glc_launch_gemm_auto(e->stream, dtype, EPI_QKV, g);
            if (gm) { set_err(gm); lok = false; }
            if (lok && e->attn_split) {      // the same tables once more as split-f16 units for the fp32 band kernel
                w.PKs = dmalloc(e, (size_t)nh * P * 64 * es);
                w.PQs = dmalloc(e, (size_t)nh * P * 64 * es);
                if (!w.PKs || !w.PQs) { lok = false; break; }
                g.Qh = w.PQs; g.Kh = w.PKs; g.qkv_split = 1;
                gm = glc_launch_gemm_auto(e->stream, dtype, EPI_QKV, g);
                if (gm) { set_err(gm); lok = false; }
            }
            if (lok && e->ln_fused && ((e->w_presplit && dtype == GLC_F32) || dtype != GLC_F32)) {
                // LayerNorm folded into the consumer GEMMs of the group-split pipeline (GemmArgs::a_stats):
                //   W1' = W1 diag(gamma1), c1 = W1' 1, d1 = W1 beta1 + b1            (this layer's ln1 feeds its FFN1)
                //   Wqkv' = Wqkv diag(gamma2 of layer l - 1), cq, dq likewise          (the previous layer's ln2 feeds this QKV; layer 0
                //   reads the embedding LayerNorm's output, which stays a kernel of its own)
                std::vector<float> wf((size_t)(I > 3 * H ? I : 3 * H) * H), cv(I > 3 * H ? I : 3 * H), dv(cv.size());
                auto fold = [&](const float* Wsrc, int rows, const float* gam, const float* bet, const float* b0, int row0) {
                    for (int n = 0; n < rows; ++n) {
                        double cs = 0.0, ds = 0.0;
                        const float* wr = Wsrc + (size_t)n * H;
                        float* o = wf.data() + (size_t)(row0 + n) * H;
                        // c[n] sums the folded weights AS THE MFMA SEES THEM (rounded to T in the 16-bit modes; hi + lo in the fp32 mode is f to 2^-22):
                        // summed from the unrounded values, rstd (acc - mean c[n]) would keep a residue mean rstd sum(round(W') - W')
                        for (int k = 0; k < H; ++k) { const float f = wr[k] * gam[k]; o[k] = f; cs += (double)round_as_operand(f, dtype); ds += (double)bet[k] * (double)wr[k]; }
                        cv[row0 + n] = (float)cs; dv[row0 + n] = (float)(ds + (double)b0[row0 + n]);
                    }
                };
                fold(t[10], I, t[8], t[9], t[11], 0);
                w.W1f = dmalloc(e, (size_t)I * H * es, false);
                if (!w.W1f || !upload_as(e, wf.data(), (size_t)I * H, w.W1f, staging)) { lok = false; break; }
                const char* pm = dtype == GLC_F32 ? glc_launch_presplit(e->stream, w.W1f, (size_t)I * H) : nullptr;     // (16-bit modes: plain rows of T)
                if (pm) { set_err(pm); lok = false; break; }
                w.c1 = upload_f32(e, cv.data(), I); w.d1 = upload_f32(e, dv.data(), I);
                if (!w.c1 || !w.d1) { lok = false; break; }
                if (l > 0) {
                    const float* const* tp = tensors + GLC_TENSORS_FIXED + GLC_TENSORS_PER_LAYER * (l - 1);
                    fold(wq.data(), H, tp[14], tp[15], bqkv.data(), 0);       // wq / bqkv still hold this layer's scaled query projection
                    fold(t[2], H, tp[14], tp[15], bqkv.data(), H);
                    fold(t[4], H, tp[14], tp[15], bqkv.data(), 2 * H);
                    w.Wqkvf = dmalloc(e, 3 * (size_t)H * H * es, false);
                    if (!w.Wqkvf || !upload_as(e, wf.data(), 3 * (size_t)H * H, w.Wqkvf, staging)) { lok = false; break; }
                    pm = dtype == GLC_F32 ? glc_launch_presplit(e->stream, w.Wqkvf, 3 * (size_t)H * H) : nullptr;
                    if (pm) { set_err(pm); lok = false; break; }
                    w.cq = upload_f32(e, cv.data(), 3 * (size_t)H); w.dq = upload_f32(e, dv.data(), 3 * (size_t)H);
                    if (!w.cq || !w.dq) { lok = false; break; }
                }
                if (hipStreamSynchronize(e->stream) != hipSuccess) { set_err("sync failed"); lok = false; break; }   // host vectors are reused
            }
            if (lok && e->mx_built) {
                // MX pipeline: the GX copies of the projection weights are built by the first forward that takes the pipeline (build_mx_weights);
                // the position tables are converted here (small, and the range guard decides per layer at load)
                if (lok && w.PKs && w.PQs) {      // the position tables as MX tiles: PQ travels as (hi8 | lo8), PK as (lo8 | hi8)
                    // (each buffer = the tile images twice: [MX steps as 32 bytes per lane — the Q / K tile format | planar: MX steps as two 16-byte planes,
                    //  every piece of a row at the SAME per-lane offset, glc_layout.h] — the band loop's requests read the second copy, round 6)
                    w.PKm = dmalloc(e, 2 * (size_t)nh * P * 64 * es);
                    w.PQm = dmalloc(e, 2 * (size_t)nh * P * 64 * es);
                    // (fp8 range guard at load: a table value beyond the e4m3 range keeps this layer's attention on split units)
                    unsigned seen = 0;
                    const char* pm = !init_range_guard(e) ? "range guard: allocation failed" : (w.PKm && w.PQm) ? glc_launch_units_to_mxt(e->stream, w.PKs, w.PKm, nh * (P / 32), 0, e->d_gxsat) : "MX position tables: allocation failed";
                    if (!pm) pm = glc_launch_units_to_mxt(e->stream, w.PQs, w.PQm, nh * (P / 32), 1, e->d_gxsat);
                    if (!pm) pm = glc_launch_units_to_mxt(e->stream, w.PKs, (unsigned char*)w.PKm + (size_t)nh * P * 64 * es, nh * (P / 32), 0, nullptr, 1);
                    if (!pm) pm = glc_launch_units_to_mxt(e->stream, w.PQs, (unsigned char*)w.PQm + (size_t)nh * P * 64 * es, nh * (P / 32), 1, nullptr, 1);
                    if (!pm && (hipMemcpyAsync(&seen, e->d_gxsat, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)) pm = "MX position tables: readback failed";
                    if (pm) { set_err(pm); lok = false; }
                    else if (seen != e->gxsat_seen[0]) { e->gxsat_seen[0] = seen; dfree(e, w.PKm); dfree(e, w.PQm); w.PKm = w.PQm = nullptr; }
                }
                if (!lok) break;
            }
        }
        if (!lok) { fail(); break; }
        // Proactive fp8 range guard: the activation rows of the MX pipeline are LayerNorm outputs (and raw sums built from
        // them); an element of a normalised row is at most |gamma| sqrt(H) + |beta|.  A checkpoint whose gains put that bound beyond e4m3's
        // 448 (outlier channels) starts with the activation exponent the reactive guard would reach after one repeated forward — no request
        // pays the 2-3x repeat.  (The bound is the worst case of one spiked row; the reactive guard stays in place behind it.)
        if (e->mx_built && ln_bound > 448.0f && e->act_sc == 0) {
            e->act_sc = kActScLow;
            fprintf(stderr, "gliclass: LayerNorm gains of this checkpoint allow activations up to %.0f (beyond the fp8 range of the MX operand images, 448); "
                            "this engine's activation rows carry exponent %d (|x| up to %d) from the start\n", ln_bound, kActScLow, 448 << -kActScLow);
        }
        const float* const* ht = tensors + GLC_TENSORS_FIXED + GLC_TENSORS_PER_LAYER * L;
        bool hok = true;
        for (int i = 0; i < 8 && hok; ++i) {
            e->headw[i] = upload_f32(e, ht[i], (i % 2 == 0) ? (size_t)H * H : (size_t)H); hok = e->headw[i] != nullptr;
            if (hok && (i % 2 == 0) && e->w_presplit) { const char* pm = glc_launch_presplit(e->stream, e->headw[i], (size_t)H * H); if (pm) { set_err(pm); hok = false; } }
        }
        hok = hok && upload_scorer(e, ht + GLC_TENSORS_HEAD);
        if (!hok) { fail(); break; }
        if (hipStreamSynchronize(e->stream) != hipSuccess) { set_err(std::string("engine_create: ") + hipGetErrorString(hipGetLastError())); fail(); break; }
        (void)hipFree(staging);
        dfree(e, rel_f32); dfree(e, rg); dfree(e, rb); dfree(e, Rf); dfree(e, Rt); dfree(e, vscratch);
        ok = true;
    } while (0);
    if (!ok) { glc_engine_destroy(e); return nullptr; }
    return e;
}

void glc_engine_destroy(glc_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->h_gxsat) (void)hipHostFree(e->h_gxsat);
    for (auto& v : e->evs) { (void)hipEventDestroy(v.a); (void)hipEventDestroy(v.b); }
    if (e->t0) (void)hipEventDestroy(e->t0);
    if (e->t1) (void)hipEventDestroy(e->t1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// One padded batch: H2D, launch sequence, D2H of logits [B, c_alloc] and per-row class-token counts.  Caller holds e->mu.
// The stream's work is complete and a device-resident forward's counter copy sits in the pinned slot: compare, remember the verdict for
// glc_engine_sync / glc_engine_device_forward_valid, and give the guard's answer (rows only: exponent kActScLow; tiles, or rows again: split-f16).
static void settle_device_range_check(glc_engine* e) {
    e->fp8_device_pending = false;
    const bool rows = e->h_gxsat[0] != e->gxsat_seen[0], tiles = e->h_gxsat[1] != e->gxsat_seen[1];
    e->gxsat_seen[0] = e->h_gxsat[0]; e->gxsat_seen[1] = e->h_gxsat[1];
    if (!rows && !tiles) return;
    if (rows && !tiles && e->act_sc == 0) { e->act_sc = kActScLow; e->device_invalid = 1; }
    else { e->fp8_sticky_off = true; e->device_invalid = 2; }
}

static int forward_one(glc_engine* e, const int64_t* ids, const int64_t* mask, int B, int S, float* logits, int c_alloc, int* cnt) {
    if (!ensure_capacity(e, B, S, c_alloc)) return -1;
    const size_t nb = (size_t)B * S * sizeof(int64_t);
    HIPCHK(hipMemcpyAsync(e->d_ids, ids, nb, hipMemcpyHostToDevice, e->stream), -1);
    HIPCHK(hipMemcpyAsync(e->d_mask, mask, nb, hipMemcpyHostToDevice, e->stream), -1);
    // Every matrix product runs on f16 / bf16 MFMA operands (the fp32 mode as split-f16 pairs), so an activation beyond the operand
    // range (|x| > 65504 for f16) turns into inf / NaN silently.  A result the reference's fp32 graph would not produce must not be
    // returned as if it were one.  With the norm folded into the GEMMs (docs/LOG_r01-r05.md §3d) the RAW residual stream is such an operand — a
    // pre-norm decoder's massive-activation channels can leave the f16 range although every normalised row is tiny — so a non-finite
    // result of a folded forward is retried ONCE with the norms as kernels of their own (residual stream plain fp32, only normalised
    // rows split); what is still non-finite then fails the call.  (fp32 mode: GLICLASS_F32_GEMM=native GLICLASS_F32_ATTN=native run the fp32 MFMAs.)
    // fp8 range (round 4): the MX pipeline's operand images carry e4m3 parts with exponent 0; an activation beyond 448 has no image there (NaN
    // from 464 on: the producers do not clamp, glc_common.h gx_split8).  Every producer counts such elements (glc_common.h gx_range_note);
    // a forward that counted any is repeated ONCE on the split-f16 kernels (operands up to 65504, the LayerNorm fold kept).  After kFp8Sticky
    // consecutive forwards that needed it the engine leaves the MX pipeline for good: the model has outlier channels, paying twice per forward is pointless.
    struct Restore {      // every exit path puts the engine's switches back (a HIP error inside a retry must not leave it unfused / off MX)
        glc_engine* e; bool fused, mx;
        ~Restore() { e->ln_fused = fused; e->mx = mx; }
    } restore{e, e->ln_fused, e->mx};
    // (a device-resident forward that has not been through glc_engine_sync yet: settle its range check first — this forward's own
    //  counter readings would otherwise make the pinned copy look stale and glc_engine_sync report a range error that never happened;
    //  and BEFORE the sticky switch is read: a verdict that turns it on must already keep this forward off the MX pipeline)
    if (e->fp8_device_pending) {
        HIPCHK(hipStreamSynchronize(e->stream), -1);
        settle_device_range_check(e);
    }
    if (e->fp8_sticky_off) e->mx = false;
    unsigned sat_now[2] = {e->gxsat_seen[0], e->gxsat_seen[1]};
    bool tried_unfused = false, tried_split = false, tried_low = false;
    for (int attempt = 0; attempt < 4; ++attempt) {      // at most: MX, MX with exponent kActScLow, split-f16, norms unfused
        if (!run_forward(e, e->d_ids, e->d_mask, B, S, c_alloc, e->d_logits)) return -1;
        HIPCHK(hipMemcpyAsync(cnt, e->cls_cnt, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, e->stream), -1);
        if (c_alloc > 0) HIPCHK(hipMemcpyAsync(logits, e->d_logits, (size_t)B * c_alloc * sizeof(float), hipMemcpyDeviceToHost, e->stream), -1);
        HIPCHK(hipMemcpyAsync(sat_now, e->d_gxsat, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream), -1);
        HIPCHK(hipStreamSynchronize(e->stream), -1);
        bool finite = true;
        for (size_t i = 0, n = (size_t)B * (c_alloc > 0 ? c_alloc : 0); i < n && finite; ++i) finite = isfinite(logits[i]);
        // the two words of the guard: activation rows (their exponent can be lowered) and Q / K / V tiles (exponent 0: only the split kernels help)
        const bool sat_rows = sat_now[0] != e->gxsat_seen[0], sat_tiles = sat_now[1] != e->gxsat_seen[1];
        const bool sat = sat_rows || sat_tiles;
        e->gxsat_seen[0] = sat_now[0]; e->gxsat_seen[1] = sat_now[1];
        if (sat_rows && !sat_tiles && e->last_mx && e->act_sc == 0 && !tried_low) {      // first answer: activation rows with exponent kActScLow, still on the MX pipeline (kept for this engine)
            tried_low = true;
            e->act_sc = kActScLow;
            e->fp8_retries++;
            fprintf(stderr, "gliclass: activations beyond the fp8 range of the MX operand images (|x| > 448); this engine's activation rows now carry exponent %d (|x| up to %d)\n", kActScLow, 448 << -kActScLow);
            continue;
        }
        if (sat && e->last_mx && !tried_split) {      // (before the non-finite check: beyond 464 the unclamped e4m3 parts are NaN — glc_common.h gx_split8 — so such a forward may well be non-finite)
            tried_split = true;
            e->mx = false;                  // retry: three f16 MFMAs per product, operands up to 65504
            e->fp8_retries++;
            if (++e->fp8_streak >= kFp8Sticky && !e->fp8_sticky_off) {
                e->fp8_sticky_off = true;
                fprintf(stderr, "gliclass: activations beyond the fp8 range of the MX operand images (|x| > 448) in %d consecutive forwards; this engine now runs the split-f16 arithmetic (as GLICLASS_MX=0)\n", e->fp8_streak);
            }
            continue;
        }
        if (!finite && !tried_unfused && e->dtype == GLC_F32 && e->last_lnf) {
            tried_unfused = true;
            e->ln_fused = false;            // retry: norms unfused (this also leaves the MX pipeline)
            e->range_retries++;
            continue;
        }
        if (e->last_mx && !sat) e->fp8_streak = 0;
        break;
    }
    if (e->profile) prof_collect(e);        // (the attempt whose result is returned)
    for (size_t i = 0, n = (size_t)B * (c_alloc > 0 ? c_alloc : 0); i < n; ++i)
        if (!isfinite(logits[i])) {
            set_err("forward: non-finite logit (row " + std::to_string(i / c_alloc) + "): an activation left the range of the " +
                    (e->dtype == GLC_BF16 ? "bf16" : "f16") + " MFMA operands" +
                    (e->dtype == GLC_F32 ? "; set GLICLASS_F32_GEMM=native GLICLASS_F32_ATTN=native for the fp32-MFMA kernels" : "; use GLICLASS_DTYPE=f32 or bf16"));
            return -1;
        }
    return 0;
}

// Length bucketing: the reference pads every row of a batch to the longest one
// (/root/reference/src/tokenizer.c:44-54).  Rows are independent end to end and columns past a row's last attended token
// contribute nothing, so a ragged batch can run as a few groups of similar length, each padded only to ITS longest row.
// Plans the partition of the rows (sorted by length) into <= max_groups contiguous groups that minimises
// sum_g ( BucketCost(n_g * roundup(len_g, 64)) + kBucketOverheadRows ); returns the group boundaries in `cuts`
// (indices into `order`).  A coarse cost model in units of padded token rows: the 256-row GEMM tiles of the narrowest
// projection (N = hidden) fill the chip in whole waves of `quantum` = 256 * CUs / (hidden / 256) rows — measured: config c3's
// 65 536 rows are exactly three waves, and two forwards of 32 768 rows each take LONGER than one of 65 536 — and every
// extra forward costs about 1 k rows of fixed work (head, pruned last layer, launch sequence).
constexpr int kBucketOverheadRows = 1024;
// cost of a forward of `rows` padded token rows, in row units: its 256-row tiles of the N = hidden projections run in whole
// waves over the CUs (256 CUs assumed when no device is visible)
struct BucketCost {
    int ncu = 256, nt = 1;
    explicit BucketCost(int hidden) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
        nt = hidden / 256 > 0 ? hidden / 256 : 1;
    }
    long long operator()(long long rows) const {
        const long long blocks = (rows + 255) / 256 * nt, waves = (blocks + ncu - 1) / ncu;
        return waves * 256 * ncu / nt;
    }
};

static void plan_buckets(const std::vector<int>& len, int max_groups, const BucketCost& cost, std::vector<int>& order, std::vector<int>& cuts) {
    const int B = (int)len.size();
    order.resize(B);
    for (int i = 0; i < B; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len[a] > len[b]; });     // longest first
    auto sp = [&](int i) { return round_up(len[order[i]] > 0 ? len[order[i]] : 1, 64); };
    const int G = max_groups < 1 ? 1 : (max_groups > B ? B : max_groups);
    // best[g][i]: minimal cost of covering rows [i, B) with exactly g groups; a group [i, j) costs (j - i) * sp(i)
    const long long INF = 1ll << 60;
    std::vector<std::vector<long long>> best(G + 1, std::vector<long long>(B + 1, INF));
    std::vector<std::vector<int>> nxt(G + 1, std::vector<int>(B + 1, B));
    for (int g = 0; g <= G; ++g) best[g][B] = g == 0 ? 0 : INF;
    for (int g = 1; g <= G; ++g)
        for (int i = B - 1; i >= 0; --i)
            for (int j = i + 1; j <= B; ++j) {
                if (best[g - 1][j] >= INF) continue;
                const long long rows = (long long)(j - i) * sp(i);
                const long long c = cost(rows) + kBucketOverheadRows + best[g - 1][j];
                if (c < best[g][i]) { best[g][i] = c; nxt[g][i] = j; }
            }
    int gbest = 1;
    for (int g = 2; g <= G; ++g) if (best[g][0] < best[gbest][0]) gbest = g;
    cuts.clear();
    for (int g = gbest, i = 0; g >= 1 && i < B; --g) { cuts.push_back(i); i = nxt[g][i]; }
    cuts.push_back(B);
}

int glc_engine_forward(glc_engine* e, const int64_t* ids, const int64_t* mask, int B, int S, float* logits, int c_alloc, int* c_out) {
    if (!e || !ids || !mask || (!logits && c_alloc > 0)) { set_err("forward: null argument"); return -1; }
    if (!check_shape(e, B, S, c_alloc)) return -1;
    // A class token under attention_mask 0 is outside the contract: its hidden state would be a padding-QUERY row (HF: uniform
    // attention over every position), which this engine does not compute — refuse instead of returning a different number.  The
    // reference's tokenizer never produces it (mask 1 on every real token, /root/reference/src/tokenizer.c:77-79).
    for (size_t i = 0, n = (size_t)B * S; i < n; ++i)
        if (mask[i] == 0 && ids[i] == e->cfg.class_token_index) {
            set_err("forward: a class token (<<LABEL>>) lies under attention_mask 0 (row " + std::to_string(i / S) + ", position " + std::to_string(i % S) + "): not supported");
            return -1;
        }
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHK(hipSetDevice(e->device), -1);
    std::vector<int> cnt(B);
    std::vector<int> order, cuts;
    if (e->max_buckets > 1 && B > 1 && !e->keep_hidden) {
        // a row's effective length: one past its last attended token or class token
        std::vector<int> len(B, 0);
        for (int b = 0; b < B; ++b) {
            const int64_t* mk = mask + (size_t)b * S;
            const int64_t* id = ids + (size_t)b * S;
            int l = 0;
            for (int s = S - 1; s >= 0; --s) if (mk[s] != 0 || id[s] == e->cfg.class_token_index) { l = s + 1; break; }
            len[b] = l;
        }
        plan_buckets(len, e->max_buckets, BucketCost(e->cfg.hidden), order, cuts);
        if (cuts.size() > 2) {
            std::vector<int64_t> gi, gm;
            std::vector<float> gl;
            for (size_t g = 0; g + 1 < cuts.size(); ++g) {
                const int i0 = cuts[g], n = cuts[g + 1] - cuts[g];
                int Sg = len[order[i0]];
                Sg = Sg < 1 ? 1 : Sg;
                gi.assign((size_t)n * Sg, 0); gm.assign((size_t)n * Sg, 0); gl.assign((size_t)n * (c_alloc > 0 ? c_alloc : 1), 0.f);
                for (int r = 0; r < n; ++r) {
                    memcpy(gi.data() + (size_t)r * Sg, ids + (size_t)order[i0 + r] * S, (size_t)Sg * sizeof(int64_t));
                    memcpy(gm.data() + (size_t)r * Sg, mask + (size_t)order[i0 + r] * S, (size_t)Sg * sizeof(int64_t));
                }
                std::vector<int> gc(n);
                if (forward_one(e, gi.data(), gm.data(), n, Sg, gl.data(), c_alloc, gc.data()) != 0) return -1;
                for (int r = 0; r < n; ++r) {
                    cnt[order[i0 + r]] = gc[r];
                    if (c_alloc > 0) memcpy(logits + (size_t)order[i0 + r] * c_alloc, gl.data() + (size_t)r * c_alloc, (size_t)c_alloc * sizeof(float));
                }
            }
            int cmax = 0;
            for (int b = 0; b < B; ++b) cmax = cnt[b] > cmax ? cnt[b] : cmax;
            if (c_out) *c_out = cmax;
            e->last_groups = (int)cuts.size() - 1;
            return 0;
        }
    }
    e->last_groups = 1;
    if (forward_one(e, ids, mask, B, S, logits, c_alloc, cnt.data()) != 0) return -1;
    int cmax = 0;
    for (int b = 0; b < B; ++b) cmax = cnt[b] > cmax ? cnt[b] : cmax;
    if (c_out) *c_out = cmax;
    return 0;
}

int glc_plan_length_buckets(const int* lengths, int B, int max_groups, int hidden, int* order, int* cuts, int* n_groups) {
    if (!lengths || B <= 0 || hidden <= 0 || !order || !cuts || !n_groups) { set_err("plan_length_buckets: bad args"); return -1; }
    std::vector<int> len(lengths, lengths + B), ord, cu;
    plan_buckets(len, max_groups, BucketCost(hidden), ord, cu);
    for (int i = 0; i < B; ++i) order[i] = ord[i];
    for (size_t i = 0; i < cu.size(); ++i) cuts[i] = cu[i];
    *n_groups = (int)cu.size() - 1;
    return 0;
}

int glc_debug_last_forward_groups(const glc_engine* e) { return e ? e->last_groups : -1; }
int glc_debug_range_retries(const glc_engine* e) { return e ? e->range_retries : -1; }
int glc_debug_fp8_range_retries(const glc_engine* e) { return e ? e->fp8_retries : -1; }
int glc_debug_fp8_range_sticky(const glc_engine* e) { return e ? (e->fp8_sticky_off ? 1 : 0) : -1; }
int glc_debug_activation_exponent(const glc_engine* e) { return e ? e->act_sc : 1; }
long long glc_debug_mx_weight_bytes(const glc_engine* e) { return e ? (long long)e->mx_bytes : -1; }
int glc_debug_set_mx2(glc_engine* e, int on) {
    if (!e) return -1;
#ifndef GLC_DEVELOPER
    if (on) { set_err("set_mx2: the bucket-space attention kernel exists in developer builds only (make DEV=1)"); return -1; }
#endif
    std::lock_guard<std::mutex> lk(e->mu); e->mx2 = on != 0; return 0;
}

int glc_engine_set_length_buckets(glc_engine* e, int max_groups) {
    if (!e || max_groups < 1 || max_groups > 64) { set_err("set_length_buckets: 1..64 groups"); return -1; }
    e->max_buckets = max_groups;
    return 0;
}

int glc_engine_forward_device(glc_engine* e, const void* d_ids, const void* d_mask, int B, int S, int C, void* d_logits) {
    if (!e || !d_ids || !d_mask || (!d_logits && C > 0)) { set_err("forward_device: null argument"); return -1; }
    if (!check_shape(e, B, S, C)) return -1;
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHK(hipSetDevice(e->device), -1);
    if (!ensure_capacity(e, B, S, C)) return -1;
    const bool mx_saved = e->mx;
    if (e->fp8_sticky_off) e->mx = false;
    const bool ok = run_forward(e, (const int64_t*)d_ids, (const int64_t*)d_mask, B, S, C, (float*)d_logits);
    e->mx = mx_saved;
    if (!ok) return -1;
    // fp8 range guard (forward_one): a device-resident forward cannot be repeated behind the caller's back — the counter travels to a pinned
    // slot behind it, and glc_engine_sync reports a forward that left the range (and moves the engine to the split arithmetic)
    if (e->last_mx) { HIPCHK(hipMemcpyAsync(e->h_gxsat, e->d_gxsat, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream), -1); e->fp8_device_pending = true; }
    return 0;
}

int glc_engine_sync(glc_engine* e) {
    if (!e) { set_err("sync: null engine"); return -1; }
    HIPCHK(hipSetDevice(e->device), -1);
    HIPCHK(hipStreamSynchronize(e->stream), -1);
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->profile) prof_collect(e);
    if (e->fp8_device_pending) settle_device_range_check(e);
    if (e->device_invalid) {
        const int what = e->device_invalid;
        e->device_invalid = 0;
        if (what == 1)
            set_err("sync: a device-resident forward since the last sync had activations beyond the fp8 range of the MX operand images (|x| > 448): its logits are "
                    "not valid; this engine's activation rows now carry exponent -5 (|x| up to 14336) — run the forward again");
        else
            set_err("sync: a device-resident forward since the last sync had activations beyond the fp8 range of the MX operand images: its logits are "
                    "not valid; this engine now runs the split-f16 arithmetic (as GLICLASS_MX=0) — run the forward again");
        return -1;
    }
    return 0;
}

/* For callers of glc_engine_forward_device that wait for the engine's stream by other means than glc_engine_sync (hipStreamSynchronize,
 * an event, a consumer queued on the stream is NOT enough: it reads the logits before anybody has looked at the counter): once the stream's
 * work is complete, 1 = the logits of every device-resident forward since the last check are valid, 0 = one of them left the fp8 range of
 * the MX operand images — its logits are not valid (NaN-poisoned operand images), the engine has changed its arithmetic as glc_engine_sync
 * would have, run it again; -1 = error.  The stream must be idle: this call does not wait. */
int glc_engine_device_forward_valid(glc_engine* e) {
    if (!e) { set_err("device_forward_valid: null engine"); return -1; }
    std::lock_guard<std::mutex> lk(e->mu);
    if (hipStreamQuery(e->stream) != hipSuccess) { set_err("device_forward_valid: the engine's stream still has work queued — wait for it first (or call glc_engine_sync)"); return -1; }
    if (e->fp8_device_pending) settle_device_range_check(e);
    const int bad = e->device_invalid;
    e->device_invalid = 0;
    return bad ? 0 : 1;
}

void* glc_device_malloc(glc_engine* e, size_t bytes) {
    if (!e) { set_err("device_malloc: null engine"); return nullptr; }
    if (hipSetDevice(e->device) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { set_err("device_malloc failed"); return nullptr; }
    return p;
}
void glc_device_free(glc_engine* e, void* p) { if (e && p) { (void)hipSetDevice(e->device); (void)hipFree(p); } }
int glc_memcpy_h2d(glc_engine* e, void* dst, const void* src, size_t bytes) {
    if (!e) return -1;
    HIPCHK(hipSetDevice(e->device), -1);
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), -1);
    return 0;
}
int glc_memcpy_d2h(glc_engine* e, void* dst, const void* src, size_t bytes) {
    if (!e) return -1;
    HIPCHK(hipSetDevice(e->device), -1);
    HIPCHK(hipStreamSynchronize(e->stream), -1);
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), -1);
    return 0;
}

int glc_timer_start(glc_engine* e) {
    if (!e) return -1;
    HIPCHK(hipSetDevice(e->device), -1);
    HIPCHK(hipEventRecord(e->t0, e->stream), -1);
    return 0;
}
float glc_timer_stop_ms(glc_engine* e) {
    if (!e) return -1.f;
    HIPCHK(hipSetDevice(e->device), -1.f);
    HIPCHK(hipEventRecord(e->t1, e->stream), -1.f);
    HIPCHK(hipEventSynchronize(e->t1), -1.f);
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->t0, e->t1), -1.f);
    return ms;
}

int glc_profile_enable(glc_engine* e, int on) {
    if (!e) return -1;
    std::lock_guard<std::mutex> lk(e->mu);
    e->profile = on != 0;
    e->ev_used = 0;
    for (int i = 0; i < PC_N; ++i) { e->prof_ms[i] = 0.f; e->prof_n[i] = 0; }
    return 0;
}
int glc_profile_read(glc_engine* e, const char** names, float* total_ms, int* launches, int max_n) {
    if (!e) return -1;
    std::lock_guard<std::mutex> lk(e->mu);
    int n = PC_N < max_n ? PC_N : max_n;
    for (int i = 0; i < n; ++i) { if (names) names[i] = kProfNames[i]; if (total_ms) total_ms[i] = e->prof_ms[i]; if (launches) launches[i] = e->prof_n[i]; }
    return n;
}

int glc_debug_set_group_split(glc_engine* e, int mode) {
    if (!e || mode < 0 || mode > 2) { set_err("group_split: 0 off, 1 auto, 2 whenever the shapes allow"); return -1; }
    e->gs_mode = mode;
    return 0;
}
int glc_debug_last_forward_group_split(const glc_engine* e) { return e ? (e->last_gs ? 1 : 0) : -1; }
/* MX cross-term pipeline on / off (needs the GX weight copies: an engine created under GLICLASS_MX=1 or =build). */
int glc_debug_set_mx(glc_engine* e, int on) {
    if (!e) return -1;
    if (on && !e->mx_built) { set_err("set_mx: the MX pipeline is not available to this engine (shapes, dtype, or GLICLASS_MX=0 at creation)"); return -1; }
    std::lock_guard<std::mutex> lk(e->mu);
    e->mx = on != 0;
    if (on) { e->fp8_sticky_off = false; e->fp8_streak = 0; e->act_sc = 0; }      // (a developer switching MX back on also clears the range guard's verdict)
    return 0;
}
int glc_debug_last_forward_mx(const glc_engine* e) { return e ? (e->last_mx ? 1 : 0) : -1; }
int glc_debug_last_forward_mx_attention(const glc_engine* e) { return e ? (e->last_mx && e->last_mx_attn ? 1 : 0) : -1; }
/* MX pipeline: attention on MX tiles (1, default) or on split-f16 units (0). */
int glc_debug_set_mx_attention(glc_engine* e, int on) { if (!e) return -1; std::lock_guard<std::mutex> lk(e->mu); e->mx_attn = on != 0; return 0; }
/* Developer: stop the next forwards after stage 10 * layer + k (k = 0 QKV, 1 attention, 2 attention-output, 3 FFN1, 4 FFN2 + LayerNorm;
 * -1 = run to the end; logits are garbage when stopped) and read workspace rows as fp32: which = 0 X, 1 H1, 2 CTX, 3 FF (row formats
 * decoded: GX after an MX forward, GS after a group-split one), 4 T1 (plain fp32), 5 statsA, 6 statsB (2 floats per row). */
int glc_debug_set_stop(glc_engine* e, int stage) { if (!e) return -1; e->debug_stop = stage; return 0; }
int glc_debug_read_workspace(glc_engine* e, int which, int rows, float* out) {
    if (!e || !out || rows <= 0 || which < 0 || which > 9) { set_err("read_workspace: bad args"); return -1; }
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHK(hipSetDevice(e->device), -1);
    HIPCHK(hipStreamSynchronize(e->stream), -1);
    if (rows > e->capM) { set_err("read_workspace: more rows than the workspace holds"); return -1; }
    const int H = e->cfg.hidden, I = e->cfg.inter;
    if (which >= 7) { HIPCHK(hipMemcpy(out, which == 7 ? e->Qh : which == 8 ? e->Kh : e->Vt, (size_t)rows * e->cfg.hidden * 4, hipMemcpyDeviceToHost), -1); return 0; }   // raw units
    if (which >= 5) { HIPCHK(hipMemcpy(out, which == 5 ? e->statsA : e->statsB, (size_t)rows * 8, hipMemcpyDeviceToHost), -1); return 0; }
    const void* src = which == 0 ? e->X : which == 1 ? e->H1 : which == 2 ? e->CTX : which == 3 ? e->FF : e->T1;
    const int W = which == 3 ? I : H;
    std::vector<unsigned char> raw((size_t)rows * W * 4);
    HIPCHK(hipMemcpy(raw.data(), src, raw.size(), hipMemcpyDeviceToHost), -1);
    if (which == 4 || !e->last_gs) { memcpy(out, raw.data(), raw.size()); return 0; }
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < W; ++c) {
            const unsigned char* g = raw.data() + ((size_t)r * W + (c & ~31)) * 4;
            _Float16 hi, lo; memcpy(&hi, g + 2 * (c & 31), 2);
            float v = (float)hi;
            if (e->last_mx) {
                const unsigned char b = g[64 + 16 * ((c & 31) >> 3) + (c & 7)];
                const int sg = b >> 7, ex = (b >> 3) & 15, mn = b & 7;
                const float l8 = ex == 0 ? ldexpf((float)mn, -9) : ldexpf(1.0f + mn / 8.0f, ex - 7);
                v += (sg ? -l8 : l8) * ldexpf(1.0f, -GLC_GX_SHIFT - e->act_sc);
            } else { memcpy(&lo, g + 64 + 2 * (c & 31), 2); v += (float)lo; }
            out[(size_t)r * W + c] = v;
        }
    return 0;
}
/* Group-split pipeline with LayerNorm folded into the GEMMs (1, default) or as kernels of its own (0).  The folded weights are built
 * at load unless GLC_LNF=0 was set then; without them the switch has no effect. */
int glc_debug_last_forward_ln_folded(const glc_engine* e) { return e ? (e->last_lnf ? 1 : 0) : -1; }
int glc_debug_set_ln_fused(glc_engine* e, int on) {
    if (!e) return -1;
    std::lock_guard<std::mutex> lk(e->mu);
    e->ln_fused = on != 0;
    return 0;
}
/* Precision budget (developer): round operand groups of the default mode's group-split pipeline to f16 (PM_* bits of glc_kernels.h). */
int glc_debug_set_precision_mask(glc_engine* e, int mask) {
    if (!e || mask < 0 || mask >= (1 << 15)) { set_err("precision_mask: 15 bits"); return -1; }
    std::lock_guard<std::mutex> lk(e->mu);
    e->prec_mask = mask;
    return 0;
}
/* 256-tile GEMM: full-line ring stages on / off, process-wide (developer A/B switch; results are bit-identical either way). */
int glc_debug_set_gemm_full_lines(int on) { glc_gemm_set_full_lines(on); return 0; }
int glc_debug_keep_hidden(glc_engine* e, int on) { if (!e) return -1; e->keep_hidden = on != 0; return 0; }
int glc_engine_set_prune_last_layer(glc_engine* e, int on) { if (!e) return -1; e->prune_last = on != 0; return 0; }
int glc_debug_set_attention_impl(glc_engine* e, int impl) {
    if (!e || impl < 0 || impl > 3) { set_err("bad attention impl"); return -1; }
    e->attn_impl = impl;
    return 0;
}
int glc_debug_get_hidden(glc_engine* e, int which, float* out, size_t out_elems) {
    if (!e || !out) { set_err("get_hidden: null"); return -1; }
    std::lock_guard<std::mutex> lk(e->mu);
    const int B = e->lastB, S = e->lastS, Sp = e->lastSp, H = e->cfg.hidden;
    if (!e->hidden_dump || B == 0 || which < 0 || which > e->cfg.layers) { set_err("get_hidden: nothing recorded"); return -1; }
    if (out_elems < (size_t)B * S * H) { set_err("get_hidden: output too small"); return -1; }
    HIPCHK(hipSetDevice(e->device), -1);
    const size_t M = (size_t)B * Sp, es = esize(e->dtype);
    float* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, M * H * sizeof(float)), -1);
    const char* m = glc_launch_to_f32(e->stream, e->dtype, (char*)e->hidden_dump + (size_t)which * M * H * es, tmp, M * H);
    if (m) { (void)hipFree(tmp); set_err(m); return -1; }
    hipError_t r = hipMemcpy2DAsync(out, (size_t)S * H * sizeof(float), tmp, (size_t)Sp * H * sizeof(float), (size_t)S * H * sizeof(float), B,
                                    hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    (void)hipFree(tmp);
    if (r != hipSuccess) { set_err(std::string("get_hidden: ") + hipGetErrorString(r)); return -1; }
    return 0;
}

/* Developer microbenchmark: time `iters` launches of one GEMM shape on random 16-bit data (HIP events).
 * which: 0 = auto (256-tile when possible), 1 = force the 128x128 kernel.  Returns ms per launch or <0. */
float glc_debug_gemm_bench(glc_engine* e, int M, int N, int K, int epi, int iters, int which) {
    // which == 6: the group-split fp32-mode kernel (rows of [32 hi | 32 lo] f16 groups, 4 bytes per element; any engine dtype)
    const int epi_abl = which / 1000;          // which = 1000 abl + 100 (1 + prio) + 9: timing ablations of the GX-row epilogue (GemmArgs::epi_abl)
    which %= 1000;
    const int which_in = which;
    if (which >= 100) which %= 100;
    if (which == 14) { set_err("gemm_bench: the one-wave-per-SIMD 128 x 128 wave tile (which = 14) was deleted in round 5 (docs/LOG_r01-r05.md section 9)"); return -1.f; }
    const bool z16b = which == 13;                    // the GX kernel with the 16 x 16 MFMA shapes
    const bool gyb = which == 11 || which == 12;      // the same kernel on GY rows (e2m3 parts with block scales); 12: plus one stamped launch
    const bool mxb = which == 9 || which == 10 || gyb || z16b;      // the MX cross-term kernel on GX rows (gemm256x.hip); 10: plus one stamped launch
    const bool gsb = which == 6 || which == 8 || mxb;
    const int mx_ws = glc_gx_weight_exponent(0.5f);
    if (!e || M <= 0 || N <= 0 || K <= 0 || iters <= 0 || (e->dtype == GLC_F32 && !gsb) || epi < EPI_BIAS || epi > EPI_RESID) { set_err("gemm_bench: bad args"); return -1.f; }
    if (M % 256 || N % 256 || K % 64) { set_err("gemm_bench: M,N %256, K %64 required"); return -1.f; }
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHK(hipSetDevice(e->device), -1.f);
    const size_t es = gsb ? 4 : 2;
    void *A = nullptr, *W = nullptr, *C = nullptr, *R = nullptr; float *bias = nullptr, *tmp = nullptr;
    const size_t nA = (size_t)M * K, nW = (size_t)N * K, nC = (size_t)M * N;
    const size_t nmax = nA > nW ? (nA > nC ? nA : nC) : (nW > nC ? nW : nC);
    float ms = -1.f;
    do {
        if (hipMalloc(&A, nA * es) || hipMalloc(&W, nW * es) || hipMalloc(&C, nC * es) || hipMalloc(&R, nC * es) ||
            hipMalloc((void**)&bias, N * sizeof(float)) || hipMalloc((void**)&tmp, nmax * sizeof(float))) { set_err("gemm_bench: alloc failed"); break; }
        std::vector<float> h(nmax);
        unsigned s = 12345u;
        const char* zenv = glc_dev_env("GLC_BENCH_DATA");       // developer: "zero" = all-zero operands, "const" = one value everywhere (how much of the time is the power envelope?)
        for (size_t i = 0; i < nmax; ++i) { s = s * 1664525u + 1013904223u; h[i] = zenv && zenv[0] == 'z' ? 0.f : zenv && zenv[0] == 'c' ? 0.37f : ((float)(s >> 8) / 8388608.f - 1.f) * 0.5f; }
        if (hipMemcpy(tmp, h.data(), nmax * sizeof(float), hipMemcpyHostToDevice)) { set_err("gemm_bench: copy failed"); break; }
        if (gsb) {      // fp32 values, split in place into the group-split image (or the GX image)
            if (hipMemcpyAsync(A, tmp, nA * 4, hipMemcpyDeviceToDevice, e->stream) || hipMemcpyAsync(W, tmp, nW * 4, hipMemcpyDeviceToDevice, e->stream) ||
                hipMemcpyAsync(R, tmp, nC * 4, hipMemcpyDeviceToDevice, e->stream)) { set_err("gemm_bench: copy failed"); break; }
            if (gyb ? (glc_launch_to_gy(e->stream, tmp, A, M, K, 0) || glc_launch_to_gy(e->stream, tmp, W, N, K, 1) || glc_launch_to_gy(e->stream, tmp, R, M, N, 0)) :
                mxb ? (glc_launch_to_gx(e->stream, A, nA, 0, 0) || glc_launch_to_gx(e->stream, W, nW, mx_ws, 1) || glc_launch_to_gx(e->stream, R, nC, 0, 0))
                    : (glc_launch_presplit(e->stream, A, nA) || glc_launch_presplit(e->stream, W, nW) || glc_launch_presplit(e->stream, R, nC))) { set_err("gemm_bench: split failed"); break; }
        } else
        if (glc_launch_convert(e->stream, e->dtype, tmp, A, nA) || glc_launch_convert(e->stream, e->dtype, tmp, W, nW) ||
            glc_launch_convert(e->stream, e->dtype, tmp, R, nC)) { set_err("gemm_bench: convert failed"); break; }
        if (hipMemcpyAsync(bias, tmp, N * sizeof(float), hipMemcpyDeviceToDevice, e->stream)) break;
        GemmArgs g; g.A = A; g.W = W; g.bias = bias; g.C = C; g.resid = R; g.Mpad = M; g.N = N; g.K = K; g.mx_ws = mx_ws;
        if (mxb && which_in >= 100) g.prio_mode = which_in / 100 - 1;      // which = 100 (1 + prio) + 9 | 10
        g.epi_abl = mxb ? epi_abl : 0;
        g.gy = gyb ? 1 : 0;
        g.z16 = z16b ? 1 : 0;
        const char* m = nullptr;
        auto launch = [&]() -> const char* { return mxb ? glc_launch_gemm256x(e->stream, epi, g) : gsb ? glc_launch_gemm256s_gs(e->stream, epi, g) : which == 1 ? glc_launch_gemm(e->stream, e->dtype, epi, g) : (which == 5 || which == 7) ? glc_launch_gemm256s(e->stream, e->dtype, epi, g) : glc_launch_gemm_auto(e->stream, e->dtype, epi, g); };
        for (int i = 0; i < 2 && !m; ++i) m = launch();
        if (m) { set_err(m); break; }
        if (hipEventRecord(e->t0, e->stream)) break;
        for (int i = 0; i < iters; ++i) launch();
        if (hipEventRecord(e->t1, e->stream) || hipEventSynchronize(e->t1)) { set_err("gemm_bench: sync failed"); break; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, e->t0, e->t1)) break;
        ms = t / iters;
        if (which == 7 || which == 8 || which == 10 || which == 12) {     // diagnostic: one stamped launch of the full-line 256-tile kernel (7: 16-bit operands, 8: group-split), EPI_BIAS
            unsigned long long* dbuf = nullptr;
            const size_t ns = 64 * 8 * 14;
            if (hipMalloc((void**)&dbuf, ns * sizeof(unsigned long long)) == hipSuccess) {
                (void)hipMemsetAsync(dbuf, 0, ns * sizeof(unsigned long long), e->stream);
                GemmArgs gd = g; gd.stamps = dbuf;
                const char* dm = (which == 10 || which == 12) ? glc_launch_gemm256x(e->stream, EPI_BIAS, gd) : which == 8 ? glc_launch_gemm256s_gs(e->stream, EPI_BIAS, gd) : glc_launch_gemm256s(e->stream, e->dtype, EPI_BIAS, gd);
                (void)hipStreamSynchronize(e->stream);
                std::vector<unsigned long long> hs(ns);
                if (!dm && hipMemcpy(hs.data(), dbuf, ns * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                    for (int grp = 0; grp < 2; ++grp) {       // wave group 0 (waves 0-3) / the late group (waves 4-7)
                        double sg[12] = {0};
                        for (int b = 0; b < 64; ++b) for (int w = 4 * grp; w < 4 * grp + 4; ++w) for (int k = 0; k < 12; ++k) sg[k] += (double)hs[((size_t)b * 8 + w) * 12 + k];
                        const double n = 64 * 4, ng = sg[11] / n > 0 ? sg[11] / n : 1;
                        fprintf(stderr, "[gemm256s stamps M=%d N=%d K=%d %s, waves %d-%d] cycles per group and wave: E: dma %.0f reads+wait %.0f barrier %.0f mfma %.0f barrier %.0f | "
                                        "O: (dma %.0f) reads+wait %.0f barrier %.0f mfma %.0f barrier %.0f | total %.0f | clock %.0f MHz\n", M, N, K, which == 12 ? "MX on GY rows" : which == 10 ? "MX" : which == 8 ? "group-split" : "16-bit", 4 * grp, 4 * grp + 3,
                                sg[0] / n / ng, sg[1] / n / ng, sg[2] / n / ng, sg[3] / n / ng, sg[4] / n / ng, sg[5] / n / ng, sg[6] / n / ng, sg[7] / n / ng, sg[8] / n / ng, sg[9] / n / ng,
                                (sg[0] + sg[1] + sg[2] + sg[3] + sg[4] + sg[5] + sg[6] + sg[7] + sg[8] + sg[9]) / n / ng, sg[10] / n / 10.0);
                    }
                    double pro = 0, epi = 0;
                    for (size_t i = 0; i < 64 * 8; ++i) { pro += (double)hs[64 * 8 * 12 + 2 * i]; epi += (double)hs[64 * 8 * 12 + 2 * i + 1]; }
                    fprintf(stderr, "[gemm256s stamps] per tile and wave: entry -> loop %.0f cycles, loop end -> stores retired %.0f cycles\n", pro / (64 * 8), epi / (64 * 8));
                } else if (dm) fprintf(stderr, "[gemm256s stamps] %s\n", dm);
                (void)hipFree(dbuf);
            }
        }
    } while (0);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(R); (void)hipFree(bias); (void)hipFree(tmp);
    return ms;
}

/* Developer check: the MX cross-term GEMM (gemm256x.hip) against the split-f16 GEMM (gemm256s.hip, GS) on the same random fp32 operands
 * (A ~ U(-a_amp, a_amp), W ~ U(-w_amp, w_amp), bias), every epilogue path:
 *   mode 0  EPI_BIAS, plain fp32 outputs                      mode 1  EPI_GELU with the LayerNorm fold (a_stats, ln_c), GX / GS row outputs
 *   mode 2  EPI_RESID, raw residual rows + r_stats / gamma / beta, raw row outputs + ln_part          mode 3  EPI_RESID, plain fp32 out
 *   mode 4  EPI_QKV with the fold (N = 3 H, Sp = 256): Q, K, V^T split-f16 units
 * out[0] = max |mx - gs|, out[1] = max |gs|, out[2] = rms(mx - gs), out[3] = rms(gs) over the decoded outputs (mode 2: + the ln_part
 * sums in out[4] = max |diff|).  Returns 0 or < 0. */
int glc_debug_gemm_mx_check(glc_engine* e, int M, int N, int K, float a_amp, float w_amp, int mode, double* out) {
    if (mode >= 30) { set_err("gemm_mx_check: mode + 30 (the one-wave-per-SIMD 128 x 128 wave tile) was deleted in round 5 (docs/LOG_r01-r05.md section 9)"); return -1; }
    const bool z16 = mode >= 20;         // mode + 20: the MX leg's main loop on the 16 x 16 MFMA shapes (GemmArgs::z16)
    if (z16) mode -= 20;
    const bool gy = mode >= 10;          // mode + 10: the MX leg on GY rows (e2m3 parts with block scales) instead of GX rows
    if (gy) mode -= 10;
    if (gy && (K % 64 || N % 64)) { set_err("gemm_mx_check: GY rows need K, N % 64 == 0"); return -1; }
    if (!e || !out || M <= 0 || N <= 0 || K <= 0 || M % 256 || N % 256 || K % 32 || mode < 0 || mode > 4) { set_err("gemm_mx_check: bad args"); return -1; }
    if (mode == 4 && (N % 768 || M % 256)) { set_err("gemm_mx_check: the QKV mode needs N = 3 H, H % 256 == 0"); return -1; }
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHK(hipSetDevice(e->device), -1);
    const size_t nA = (size_t)M * K, nW = (size_t)N * K, nC = (size_t)M * N;
    float *A = nullptr, *W = nullptr, *A2 = nullptr, *W2 = nullptr, *C0 = nullptr, *C1 = nullptr, *R0 = nullptr, *R1 = nullptr, *bias = nullptr, *lnc = nullptr, *gam = nullptr, *bet = nullptr;
    float2 *st = nullptr, *lp0 = nullptr, *lp1 = nullptr;
    void *Ay = nullptr, *Wy = nullptr, *Ry = nullptr, *Cy = nullptr;
    int rc = -1;
    do {
        if (gy && (hipMalloc(&Ay, nA * 4) || hipMalloc(&Wy, nW * 4) || hipMalloc(&Ry, nC * 4) || hipMalloc(&Cy, nC * 4))) { set_err("gemm_mx_check: alloc failed"); break; }
        if (hipMalloc((void**)&A, nA * 4) || hipMalloc((void**)&W, nW * 4) || hipMalloc((void**)&A2, nA * 4) || hipMalloc((void**)&W2, nW * 4) ||
            hipMalloc((void**)&C0, nC * 4) || hipMalloc((void**)&C1, nC * 4) || hipMalloc((void**)&R0, nC * 4) || hipMalloc((void**)&R1, nC * 4) ||
            hipMalloc((void**)&bias, (size_t)N * 4) || hipMalloc((void**)&lnc, (size_t)N * 4) || hipMalloc((void**)&gam, (size_t)N * 4) || hipMalloc((void**)&bet, (size_t)N * 4) ||
            hipMalloc((void**)&st, (size_t)M * 8) || hipMalloc((void**)&lp0, (size_t)M * (N / 64) * 8) || hipMalloc((void**)&lp1, (size_t)M * (N / 64) * 8)) { set_err("gemm_mx_check: alloc failed"); break; }
        std::vector<float> ha(nA), hw(nW), hb(N), hc(N), hg(N), hbe(N), hr(nC);
        std::vector<float2> hst(M);
        unsigned s = 777u + 13u * mode;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.f - 1.f; };
        const bool special = a_amp < 0.f;      // developer: a = 1 + 2^-12 (hi 1, lo 2^-12), w = 1, no bias: every output = K (1 + 2^-12) iff the a_lo w_hi terms arrive
        const int spec = special ? (int)(-a_amp + 0.5f) : 0;      // 1: a = 1 + 2^-12, w = 1; 2: a = 1, w = 1 + 2^-12; 3: both; 4: a = 3 + 3 2^-12 (hi 3, lo), w = 1
        if (special) a_amp = 1.f;
        for (auto& v : ha) v = rnd() * a_amp;
        for (auto& v : hw) v = rnd() * w_amp;
        for (auto& v : hr) v = rnd() * a_amp;
        for (int n = 0; n < N; ++n) { hb[n] = rnd() * 0.1f; hc[n] = rnd() * 0.05f; hg[n] = 1.f + 0.3f * rnd(); hbe[n] = 0.2f * rnd(); }
        for (int m = 0; m < M; ++m) hst[m] = make_float2(0.1f * rnd() * a_amp, (0.5f + 0.4f * rnd()) / a_amp);
        if (special) { const float d = ldexpf(1.f, -12); for (auto& v : ha) v = spec == 2 ? 1.0f : spec == 4 ? 3.0f + 3.0f * d : 1.0f + d; for (auto& v : hw) v = spec >= 2 && spec <= 3 ? 1.0f + d : 1.0f; for (auto& v : hb) v = 0.f; }
        if (hipMemcpy(A, ha.data(), nA * 4, hipMemcpyHostToDevice) || hipMemcpy(W, hw.data(), nW * 4, hipMemcpyHostToDevice) ||
            hipMemcpy(A2, ha.data(), nA * 4, hipMemcpyHostToDevice) || hipMemcpy(W2, hw.data(), nW * 4, hipMemcpyHostToDevice) ||
            hipMemcpy(R0, hr.data(), nC * 4, hipMemcpyHostToDevice) || hipMemcpy(R1, hr.data(), nC * 4, hipMemcpyHostToDevice) ||
            hipMemcpy(bias, hb.data(), (size_t)N * 4, hipMemcpyHostToDevice) || hipMemcpy(lnc, hc.data(), (size_t)N * 4, hipMemcpyHostToDevice) ||
            hipMemcpy(gam, hg.data(), (size_t)N * 4, hipMemcpyHostToDevice) || hipMemcpy(bet, hbe.data(), (size_t)N * 4, hipMemcpyHostToDevice) ||
            hipMemcpy(st, hst.data(), (size_t)M * 8, hipMemcpyHostToDevice) || hipMemset(C0, 0, nC * 4) || hipMemset(C1, 0, nC * 4)) { set_err("gemm_mx_check: copy failed"); break; }
        const int ws = glc_gx_weight_exponent(w_amp);
        const char* m = glc_launch_presplit(e->stream, A, nA);
        if (!m) m = glc_launch_presplit(e->stream, W, nW);
        if (!m) m = glc_launch_presplit(e->stream, R0, nC);
        if (gy) {
            if (!m) m = glc_launch_to_gy(e->stream, A2, Ay, M, K, 0);
            if (!m) m = glc_launch_to_gy(e->stream, W2, Wy, N, K, 1);
            if (!m) m = glc_launch_to_gy(e->stream, R1, Ry, M, N, 0);
        } else {
        if (!m) m = glc_launch_to_gx(e->stream, A2, nA, 0, 0);
        if (!m) m = glc_launch_to_gx(e->stream, W2, nW, ws, 1);
        if (!m) m = glc_launch_to_gx(e->stream, R1, nC, 0, 0);
        }
        GemmArgs g; g.bias = bias; g.Mpad = M; g.N = N; g.K = K;
        int epi = EPI_BIAS;
        if (mode == 0) g.gs_c_plain = 1;
        if (mode == 1) { epi = EPI_GELU; g.a_stats = st; g.ln_c = lnc; }
        if (mode == 2 || mode == 3) { epi = EPI_RESID; if (mode == 2) { g.r_stats = st; g.r_gamma = gam; g.r_beta = bet; } }
        if (mode == 4) { epi = EPI_QKV; g.a_stats = st; g.ln_c = lnc; g.H = N / 3; g.nh = g.H / 64; g.Sp = 256; g.Mvalid = M; g.qkv_split = 1; }
        const size_t third = nC / 3;
        g.A = A; g.W = W; g.C = C0; g.resid = R0; if (mode == 2) g.ln_part = lp0;
        if (mode == 4) { g.Qh = C0; g.Kh = C0 + third; g.Vt = C0 + 2 * third; }
        if (!m) m = glc_launch_gemm256s_gs(e->stream, epi, g);
        g.A = A2; g.W = W2; g.C = C1; g.resid = R1; g.mx_ws = ws; if (mode == 2) g.ln_part = lp1;
        if (mode == 4) { g.Qh = C1; g.Kh = C1 + third; g.Vt = C1 + 2 * third; }
        const bool rows_out = mode == 1 || mode == 2;
        if (gy) { g.gy = 1; g.A = Ay; g.W = Wy; g.resid = Ry; if (rows_out) g.C = Cy; }
        g.z16 = z16 ? 1 : 0;
        if (!m) m = glc_launch_gemm256x(e->stream, epi, g);
        if (gy && rows_out && !m) m = glc_launch_gy_to_f32(e->stream, Cy, C1, M, N);        // (the GY output rows decoded on the device)
        if (m) { set_err(m); break; }
        std::vector<float> c0(nC), c1(nC);
        if (hipStreamSynchronize(e->stream) || hipMemcpy(c0.data(), C0, nC * 4, hipMemcpyDeviceToHost) || hipMemcpy(c1.data(), C1, nC * 4, hipMemcpyDeviceToHost)) { set_err("gemm_mx_check: readback failed"); break; }
        // decode the row formats on the host: GS group = [32 hi halves | 32 lo halves]; GX group = [32 hi halves | 32 lo8 | 32 hi8]; QKV units = [8 hi | 8 lo] halves
        auto half_at = [](const float* base, size_t hidx) { _Float16 hv; memcpy(&hv, reinterpret_cast<const unsigned char*>(base) + 2 * hidx, 2); return (float)hv; };
        auto fp8_at = [](const float* base, size_t bidx) {
            const unsigned char b = reinterpret_cast<const unsigned char*>(base)[bidx];
            const int sg = b >> 7, ex = (b >> 3) & 15, mn = b & 7;
            const float v = ex == 0 ? ldexpf((float)mn, -9) : ldexpf(1.0f + mn / 8.0f, ex - 7);
            return sg ? -v : v;
        };
        const bool rows_gs = mode == 1 || mode == 2, units = mode == 4;
        double md = 0, mr = 0, sd = 0, sr = 0;
        for (size_t i = 0; i < nC; ++i) {
            double v0, v1;
            if (rows_gs) {
                const size_t row = i / N, col = i % N, grp = col >> 5, e5 = col & 31;
                v0 = half_at(c0.data(), (row * N + grp * 32) * 2 + e5) + half_at(c0.data(), (row * N + grp * 32) * 2 + 32 + e5);
                if (gy) v1 = c1[i];
                else
                v1 = half_at(c1.data(), (row * N + grp * 32) * 2 + e5) + fp8_at(c1.data(), (row * N + grp * 32) * 4 + 64 + 16 * (e5 >> 3) + (e5 & 7)) * ldexp(1.0, -GLC_GX_SHIFT);
            } else if (units) {
                const size_t u = i >> 3, j = i & 7;
                v0 = half_at(c0.data(), u * 16 + j) + half_at(c0.data(), u * 16 + 8 + j);
                v1 = half_at(c1.data(), u * 16 + j) + half_at(c1.data(), u * 16 + 8 + j);
            } else { v0 = c0[i]; v1 = c1[i]; }
            const double d = v1 - v0;
            if (!(fabs(d) <= md)) md = fabs(d);
            if (fabs(v0) > mr) mr = fabs(v0);
            sd += d * d; sr += v0 * v0;
        }
        out[0] = md; out[1] = mr; out[2] = sqrt(sd / nC); out[3] = sqrt(sr / nC); out[4] = 0;
        if (special && gy) {
            unsigned char ra[128], rw[128];
            (void)hipMemcpy(ra, (const unsigned char*)Ay + 64, 48, hipMemcpyDeviceToHost); (void)hipMemcpy(rw, (const unsigned char*)Wy + 64, 48, hipMemcpyDeviceToHost);
            fprintf(stderr, "  Ay row 0 fp6 area:"); for (int i = 0; i < 48; ++i) fprintf(stderr, " %02x", ra[i]);
            fprintf(stderr, "\n  Wy row 0 fp6 area:"); for (int i = 0; i < 48; ++i) fprintf(stderr, " %02x", rw[i]);
            (void)hipMemcpy(ra, (const unsigned char*)Ay + (size_t)(K / 32) * 112, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(rw, (const unsigned char*)Wy + (size_t)(K / 32) * 112, 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "\n  scale bytes A %d %d %d %d  W %d %d %d %d\n", ra[0], ra[1], ra[2], ra[3], rw[0], rw[1], rw[2], rw[3]);
        }
        if (special) fprintf(stderr, "[gemm_mx_check special] K = %d: split-f16 C[0] = %.6f, MX leg C[0] = %.6f C[1] = %.6f C[N+5] = %.6f; (case %d)\n", K, c0[0], c1[0], c1[1], c1[N + 5], spec);
        if (mode == 2) {
            std::vector<float2> p0((size_t)M * (N / 64)), p1(p0.size());
            if (hipMemcpy(p0.data(), lp0, p0.size() * 8, hipMemcpyDeviceToHost) || hipMemcpy(p1.data(), lp1, p1.size() * 8, hipMemcpyDeviceToHost)) { set_err("gemm_mx_check: readback failed"); break; }
            double pd = 0;
            for (size_t i = 0; i < p0.size(); ++i) { pd = fmax(pd, fabs((double)p0[i].x - p1[i].x)); pd = fmax(pd, fabs((double)p0[i].y - p1[i].y) / (1.0 + fabs(p0[i].y))); }
            out[4] = pd;
        }
        rc = 0;
    } while (0);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(A2); (void)hipFree(W2); (void)hipFree(C0); (void)hipFree(C1); (void)hipFree(R0); (void)hipFree(R1);
    (void)hipFree(bias); (void)hipFree(lnc); (void)hipFree(gam); (void)hipFree(bet); (void)hipFree(st); (void)hipFree(lp0); (void)hipFree(lp1);
    (void)hipFree(Ay); (void)hipFree(Wy); (void)hipFree(Ry); (void)hipFree(Cy);
    return rc;
}

/* Developer microbenchmark: re-run the band attention kernel `iters` times on the Q/K/V^T that the last forward left in the
 * workspace (layer-0 position tables), HIP-event timed.  checksum[0..1] = sum and sum of squares of the context output;
 * variant is passed through to the kernel; stamps != 0 adds one launch of the s_memtime-instrumented build and prints
 * the per-tile segment cycles.  Returns ms per launch or < 0. */
float glc_debug_attn_bench(glc_engine* e, int iters, int variant, int stamps, double* checksum) {
    if (!e || iters <= 0 || (e->dtype == GLC_F32 && !e->attn_split) || e->lastB <= 0 || e->cfg.backbone != GLC_BACKBONE_DEBERTA) {
        set_err("attn_bench: needs a DeBERTa engine (16-bit, or fp32 with split-f16 attention) and a previous forward"); return -1.f;
    }
#ifndef GLC_DEVELOPER
    if (stamps || (variant & (256 | 512 | 4096 | 8192 | 16384 | 32768 | 65536 | 131072 | 262144 | 524288 | 1048576 | 2097152))) {
        set_err("attn_bench: stamped builds, timing-only builds (wrong results) and the rejected attention kernels exist in developer builds only (make DEV=1)"); return -1.f;
    }
#endif
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHK(hipSetDevice(e->device), -1.f);
    const int B = e->lastB, Sp = e->lastSp, H = e->cfg.hidden, nh = e->cfg.heads;
    const LayerW& w = e->layers[0];
    const bool sp = e->dtype == GLC_F32;
    AttnArgs a{e->Qh, e->Kh, e->Vt, sp ? w.PKs : w.PK, sp ? w.PQs : w.PQ, e->dtabs[Sp], e->kbias, e->klen, e->kfirst, e->CTX, B, nh, Sp, H, e->P};
    a.rsat_pos = e->dsat[Sp].first; a.rsat_neg = e->dsat[Sp].second; a.variant = variant & 123; a.otab = e->otabs[Sp]; a.mtab = e->mtabs.count(Sp) ? e->mtabs[Sp] : nullptr; a.split = sp;    // bits 0-1: per-wave kernel diagnostics; bit 3: wg kernel without the K/V ring; bits 4 / 5: wg kernel with / without the half-tile stagger
    hipStream_t st = e->stream;
    const bool wg = (variant & 4) != 0;                       // bit 2: the workgroup-shared kernel (attention_wg.hip)
    const bool mxk = (variant & 128) != 0;                    // bit 7: the MX-tile kernel (attention_mx.hip) on the MX tiles the last (MX) forward left; bits 8 / 9: its timing-only builds
    if (mxk) {
        if (!(sp && e->last_mx && e->mx_attn && w.PKm && w.PQm)) { set_err("attn_bench: the MX kernel needs a previous forward of the MX pipeline with MX attention"); return -1.f; }
        a.PK = w.PKm; a.PQ = w.PQm; a.ctx_gs = 2; a.variant = variant & (256 | 512 | 1024 | 2048 | 4096 | 16384 | 65536 | 131072 | 262144 | 1048576 | 2097152);
        a.idx16 = e->mx2tabs[Sp].first; a.tinfo = e->mx2tabs[Sp].second;
    }
    const bool mxk2 = mxk && (variant & 8192) != 0;           // bit 13: the bucket-space MX kernel (attention_mx2.hip)
    if (mxk2 && !(a.idx16 && a.tinfo)) { set_err("attn_bench: no mx2 tables for this length"); return -1.f; }
    const bool mxd = mxk && (variant & 524288) != 0;         // bit 19: two query tiles per wave, one wave per SIMD (csrc/dev/attention_mxd.hip; developer builds)
    auto launch = [&]() -> const char* { return mxd ? glc_launch_attention_mxd(st, a) : mxk2 ? glc_launch_attention_mx2(st, a) : mxk ? glc_launch_attention_mx(st, a) : wg ? glc_launch_attention_wg(st, e->dtype, a) : glc_launch_attention(st, e->dtype, 2, a); };
    for (int i = 0; i < 2; ++i) KCHK(launch(), -1.f);
    HIPCHK(hipEventRecord(e->t0, st), -1.f);
    for (int i = 0; i < iters; ++i) launch();
    HIPCHK(hipEventRecord(e->t1, st), -1.f);
    HIPCHK(hipEventSynchronize(e->t1), -1.f);
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e->t0, e->t1), -1.f);
    if (checksum) {
        const size_t n = (size_t)B * Sp * H;
        float* tmp = nullptr;
        HIPCHK(hipMalloc((void**)&tmp, n * sizeof(float)), -1.f);
        std::vector<float> h(n);
        const char* m = mxk ? nullptr : glc_launch_to_f32(st, e->dtype, e->CTX, tmp, n);
        if (mxk) { HIPCHK(hipMemsetAsync(tmp, 0, n * sizeof(float), st), -1.f); }        // (GX rows: no checksum)
        hipError_t r = m ? hipErrorUnknown : hipMemcpyAsync(h.data(), tmp, n * sizeof(float), hipMemcpyDeviceToHost, st);
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        (void)hipFree(tmp);
        if (r != hipSuccess) { set_err("attn_bench: readback failed"); return -1.f; }
        double s1 = 0, s2 = 0;
        for (size_t i = 0; i < n; ++i) { s1 += h[i]; s2 += (double)h[i] * h[i]; }
        checksum[0] = s1; checksum[1] = s2;
    }
    if (stamps && wg && sp && !(variant & 121)) {       // the split-f16 workgroup kernel, stamped build: 64 workgroups x 8 waves x 8 counters
        unsigned long long* dbuf = nullptr;
        const size_t ns = 64 * 8 * 8;
        if (hipMalloc((void**)&dbuf, ns * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemsetAsync(dbuf, 0, ns * sizeof(unsigned long long), st);
            AttnArgs as = a; as.stamps = dbuf;
            const char* m = glc_launch_attention_wg(st, e->dtype, as);
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> hs(ns);
            if (!m && hipMemcpy(hs.data(), dbuf, ns * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                double s[8] = {0};
                for (size_t i = 0; i < 64 * 8; ++i) for (int k = 0; k < 8; ++k) s[k] += (double)hs[i * 8 + k];
                const double nt = s[7] > 0 ? s[7] : 1;
                fprintf(stderr, "[attn_wg stamps] per band tile per wave (s_memtime ticks), %.0f tiles: request wait %.0f | K+gather+p2c/S issue %.0f | barrier X %.0f | "
                                "image stores + barrier Y %.0f | DMA, image gather, c2p issue %.0f | softmax + P.V + c2p store %.0f | total %.0f | s_memtime clock %.0f MHz\n",
                        nt, s[0] / nt, s[1] / nt, s[2] / nt, s[3] / nt, s[4] / nt, s[5] / nt, (s[0] + s[1] + s[2] + s[3] + s[4] + s[5]) / nt, s[6] / (64 * 8) / 10.0);
            } else if (m) fprintf(stderr, "[attn_wg stamps] %s\n", m);
            (void)hipFree(dbuf);
        }
    }
    if (stamps && mxk) {       // the MX-tile kernel, stamped build: 64 workgroups x 8 waves x 10 counters
        unsigned long long* dbuf = nullptr;
        const size_t ns = 64 * 8 * 10;
        if (hipMalloc((void**)&dbuf, ns * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemsetAsync(dbuf, 0, ns * sizeof(unsigned long long), st);
            AttnArgs as = a; as.stamps = dbuf;
            const char* m = mxd ? glc_launch_attention_mxd(st, as) : mxk2 ? glc_launch_attention_mx2(st, as) : glc_launch_attention_mx(st, as);
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> hs(ns);
            if (!m && hipMemcpy(hs.data(), dbuf, ns * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                double s[10] = {0};
                for (size_t i = 0; i < 64 * 8; ++i) for (int k = 0; k < 10; ++k) s[k] += (double)hs[i * 10 + k];
                const double nt = s[9] > 0 ? s[9] : 1;
                double tot = 0;
                for (int k = 0; k < 8; ++k) tot += s[k];
                if (mxk2)
                    fprintf(stderr, "[attn_mx2 stamps] per generic tile per wave (s_memtime ticks), %.0f tiles: K fragments + c2p store / gather + p2c / S issue %.0f | request wait + ring barrier %.0f | "
                                    "DMA + row requests %.0f | image stores + gathers %.0f | softmax + P.V %.0f | c2p issue %.0f | total %.0f | s_memtime clock %.0f MHz\n",
                            nt, (s[0] + s[1]) / nt, s[2] / nt, s[3] / nt, s[4] / nt, s[5] / nt, s[6] / nt, tot / nt, s[8] / (64 * 8) / 10.0);
                else if (mxd)
                    fprintf(stderr, "[attn_mxd stamps] per band step (two query tiles) per wave (s_memtime ticks), %.0f steps: DMA + row requests %.0f | gathers, bias, maxima %.0f | barrier X' %.0f | "
                                    "M first half + softmax A %.0f | M second half + P.V A + softmax B %.0f | P.V B, c2p, ring stores %.0f | requests landed + barrier Y' %.0f | total %.0f\n",
                            nt, s[0] / nt, s[1] / nt, s[2] / nt, s[3] / nt, s[4] / nt, s[5] / nt, s[6] / nt, tot / nt);
                else
                fprintf(stderr, "[attn_mx stamps] per band tile per wave (s_memtime ticks), %.0f tiles: request wait %.0f | K + c2p gather + p2c/S issue %.0f | row requests %.0f | "
                                "barrier X %.0f | image stores + barrier Y %.0f | DMA + image gather %.0f | c2p issue + softmax + P.V %.0f | c2p store %.0f | total %.0f | s_memtime clock %.0f MHz\n",
                        nt, s[0] / nt, s[1] / nt, s[2] / nt, s[3] / nt, s[4] / nt, s[5] / nt, s[6] / nt, s[7] / nt, tot / nt, s[8] / (64 * 8) / 10.0);
            } else if (m) fprintf(stderr, "[attn_mx stamps] %s\n", m);
            (void)hipFree(dbuf);
        }
    }
    if (stamps && !wg && !sp) {
        unsigned long long* dbuf = nullptr;
        const size_t ns = 64 * 4 * 8;
        if (hipMalloc((void**)&dbuf, ns * sizeof(unsigned long long)) == hipSuccess) {
            (void)hipMemsetAsync(dbuf, 0, ns * sizeof(unsigned long long), st);
            AttnArgs as = a; as.stamps = dbuf;
            glc_launch_attention(st, e->dtype, 2, as);
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> hs(ns);
            if (hipMemcpy(hs.data(), dbuf, ns * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                double s[8] = {0};
                for (size_t i = 0; i < 64 * 4; ++i) for (int k = 0; k < 8; ++k) s[k] += (double)hs[i * 8 + k];
                const double nt = s[7] > 0 ? s[7] : 1;
                fprintf(stderr, "[attn stamps] per band tile per wave (s_memtime ticks), %0.f tiles: mfma_qk_p2c+stores %.0f | lds_sync %.0f | gather %.0f | "
                                "max+xchg %.0f | exp+sum %.0f | cvt+pv %.0f | c2p_next %.0f | total %.0f\n",
                        nt, s[0] / nt, s[1] / nt, s[2] / nt, s[3] / nt, s[4] / nt, s[5] / nt, s[6] / nt,
                        (s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6]) / nt);
            }
            (void)hipFree(dbuf);
        }
    }
    return t / iters;
}

/* 1: this library was built with make DEV=1 (developer kernels, stamps and the GLC_* environment switches compiled in); 0: the product library. */
int glc_debug_is_developer_build(void) {
#ifdef GLC_DEVELOPER
    return 1;
#else
    return 0;
#endif
}

const glc_model_config* glc_engine_config(const glc_engine* e) { return e ? &e->cfg : nullptr; }
int glc_engine_dtype(const glc_engine* e) { return e ? e->dtype : -1; }

}  // extern "C"
