/*
 * gliclass_hip.h — C-ABI of the MI355X (gfx950) GLiClass forward engine ("the HIP shim").
 *
 * This is the boundary the pure-C host code (gliclass/c_amd/host/model.c — the drop-in for
 * /root/reference/src/model.c) binds to.  Plain pointers and sizes only; no C++/torch types.
 * Each entry point names the reference interface it stands in for:
 *
 *   glc_engine_create      <- g_ort->CreateSession(env, model_path, opts, &session)
 *                             /root/reference/src/model.c:269 (graph load + optimise, once)
 *   glc_engine_forward     <- g_ort->Run(session, ..., input_ids, attention_mask -> logits)
 *                             /root/reference/src/model.c:173-182  (the whole hot path)
 *   glc_engine_destroy     <- g_ort->ReleaseSession          /root/reference/main.c:186
 *   glc_delta_table        <- make_log_bucket_position in the exported graph
 *   glc_last_error         <- g_ort->GetErrorMessage(status) /root/reference/src/model.c:194
 *
 * Error convention (mirrors /root/reference/src/model.c): pointer-returning calls return NULL,
 * int-returning calls return non-zero; the message is available from glc_last_error().
 */
#ifndef GLICLASS_HIP_H
#define GLICLASS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* arithmetic type of the GEMM/attention operands (accumulation, LayerNorm statistics, softmax and
 * the scorer head are always fp32) */
enum { GLC_F32 = 0, GLC_BF16 = 1, GLC_F16 = 2 };
enum { GLC_POOL_FIRST = 0, GLC_POOL_AVG = 1, GLC_POOL_LAST = 2 /* last attended token (decoder backbones) */ };
/* scorer of the GLiClass head (upstream `scorer_type`; SURVEY.md §8a row a12 — restated, parity unpinned): 'simple' = dot product of the
 * projected text and class features; 'weighted-dot' = out_mlp([t1, c1, t2 * c2]) on (t1|t2) = proj_text(text), (c1|c2) = proj_label(class),
 * out_mlp = Linear(3H,4H) -> ReLU -> Linear(4H,1); 'mlp' = Linear(2H,256) -> ReLU -> Linear(256,128) -> ReLU -> Linear(128,1) on [text, class] */
enum { GLC_SCORER_DOT = 0, GLC_SCORER_WEIGHTED_DOT = 1, GLC_SCORER_MLP = 2 };
#define GLC_SCORER_MLP_HIDDEN 256
/* backbone family: DeBERTa-v2/v3 disentangled encoder, or a decoder-style stack with Qwen2 arithmetic (RMSNorm, RoPE,
 * grouped-query attention, SwiGLU; SURVEY.md §8a row a16, BASELINE.json configs[4]) */
enum { GLC_BACKBONE_DEBERTA = 0, GLC_BACKBONE_DECODER = 1 };

/* Same int/float slots, same order, as the .glcw blob header (gliclass/c_amd/weights.py). */
typedef struct glc_model_config {
    int32_t vocab, hidden, layers, heads, head_dim, inter, pos_buckets, max_rel_pos;
    int32_t pad_id, cls_id, sep_id, class_token_index, text_token_index;
    int32_t pooling, scorer, embed_class_token, normalize_features;
    int32_t backbone, kv_heads, causal;    /* decoder backbone: key/value heads (0 = heads), causal mask on/off */
    float ln_eps, logit_scale;             /* ln_eps is rms_norm_eps for the decoder backbone */
    float rope_theta;
} glc_model_config;

/* Tensor order expected in `tensors[]` (all fp32, row-major, nn.Linear weights are [out,in]):
 *   0 embeddings.word_embeddings.weight [vocab,H]      1,2 embeddings.LayerNorm.{weight,bias}
 *   3 encoder.rel_embeddings.weight [2*span,H]         4,5 encoder.LayerNorm.{weight,bias}
 *   6+16*l .. : layer l: q.w q.b k.w k.b v.w v.b  attn.out.w attn.out.b attn.LN.w attn.LN.b
 *                        inter.w inter.b  out.w out.b out.LN.w out.LN.b
 *   then text_projector.linear_1.{w,b} linear_2.{w,b}, classes_projector.linear_1.{w,b} linear_2.{w,b}
 *   then the scorer's tensors (none for 'simple'):
 *     weighted-dot: scorer.proj_text.{weight [2H,H], bias [2H]}  scorer.proj_label.{weight [2H,H], bias}
 *                   scorer.out_mlp.0.{weight [4H,3H], bias [4H]}  scorer.out_mlp.3.{weight [1,4H], bias [1]}
 *     mlp:          scorer.mlp.0.{weight [256,2H], bias}  scorer.mlp.2.{weight [128,256], bias}  scorer.mlp.4.{weight [1,128], bias [1]} */
#define GLC_TENSORS_FIXED 6
#define GLC_TENSORS_PER_LAYER 16
#define GLC_TENSORS_HEAD 8
static inline int glc_num_tensors(int layers) { return GLC_TENSORS_FIXED + GLC_TENSORS_PER_LAYER * layers + GLC_TENSORS_HEAD; }
static inline int glc_num_scorer_tensors(int scorer) { return scorer == GLC_SCORER_WEIGHTED_DOT ? 8 : scorer == GLC_SCORER_MLP ? 6 : 0; }
/* Decoder backbone (names of HF Qwen2Model.state_dict()):
 *   0 embed_tokens.weight [vocab,H]
 *   1+12*l .. : layer l: input_layernorm.weight  q_proj.{w,b} [nq*d,H]  k_proj.{w,b} [nkv*d,H]  v_proj.{w,b}
 *                        o_proj.weight [H,nq*d]  post_attention_layernorm.weight
 *                        mlp.gate_proj.weight [I,H]  mlp.up_proj.weight [I,H]  mlp.down_proj.weight [H,I]
 *   then norm.weight [H], then the same 8 head tensors */
#define GLC_DEC_TENSORS_PER_LAYER 12
static inline int glc_num_tensors_cfg(const glc_model_config* c) {
    return (c->backbone == GLC_BACKBONE_DECODER ? 2 + GLC_DEC_TENSORS_PER_LAYER * c->layers + GLC_TENSORS_HEAD : glc_num_tensors(c->layers)) +
           glc_num_scorer_tensors(c->scorer);
}

typedef struct glc_engine glc_engine;

int glc_device_count(void);
const char* glc_last_error(void);

/* Uploads + converts weights, precomputes LayerNorm(rel_embeddings) and the per-layer position
 * projections (batch independent).  `device` = HIP ordinal.  Host tensors may be freed afterwards. */
glc_engine* glc_engine_create(const glc_model_config* cfg, const float* const* tensors, int n_tensors,
                              int device, int dtype);
void glc_engine_destroy(glc_engine* e);

/* Host-buffer forward: ids/mask int64 [B,S] row-major (what create_tensor() wraps,
 * /root/reference/src/model.c:39-71).  Writes logits[b*c_alloc + j], j < *c_out, where
 * *c_out = max over rows of the number of class tokens (the ONNX graph's dynamic C).
 * Blocking; thread-safe per engine (internally serialised, cf. /root/reference/main.c:143-146). */
int glc_engine_forward(glc_engine* e, const int64_t* ids, const int64_t* mask, int B, int S,
                       float* logits, int c_alloc, int* c_out);

/* Device-resident forward (bench / pipelined callers): d_ids, d_mask int64 [B,S] and d_logits
 * f32 [B,C] are device pointers on this engine's device; C = number of class-token slots to score.
 * Enqueues on the engine stream and returns; call glc_engine_sync() to wait.
 *
 * CONTRACT (fp8 range guard of the MX pipeline, the default arithmetic of large fp32-mode forwards): the operand images of that pipeline
 * hold e4m3 parts; an activation beyond their range (|x| > 448) has no image, and from 464 on its parts are NaN.  A host-buffer forward
 * (glc_engine_forward) notices and repeats itself behind the caller's back; a device-resident forward cannot.  Its d_logits are valid
 * only once glc_engine_sync() has returned 0 — or glc_engine_device_forward_valid() has returned 1 for callers that wait by other means.
 * glc_engine_sync() returning -1 with "... run the forward again" means: the logits of the forward(s) since the last sync are NOT valid,
 * the engine has already changed its arithmetic (first answer: activation rows at exponent -5, |x| up to 14336, still on the MX pipeline;
 * second answer, or outliers in Q / K / V: the split-f16 kernels for good, as GLICLASS_MX=0), and the same forward has to be enqueued again.
 * At most two such repeats per engine lifetime.  Consumers queued on the stream BEHIND the forward (an RCCL all-gather, a D2H copy) read
 * whatever the forward wrote: check before trusting them.  Engines created with GLICLASS_MX=0, 16-bit engines and small forwards never
 * take the MX pipeline and never report this. */
int glc_engine_forward_device(glc_engine* e, const void* d_ids, const void* d_mask, int B, int S, int C,
                              void* d_logits);
int glc_engine_sync(glc_engine* e);
int glc_engine_device_forward_valid(glc_engine* e);      /* 1 valid / 0 repeat the forward / -1 error; the stream must be idle (see above) */

/* Exact last-layer pruning (default on; env GLICLASS_PRUNE_LAST=0 disables): the final encoder layer computes
 * Q / attention output / FFN only for the rows the head reads ([CLS] + class tokens).  Logits are unchanged. */
int glc_engine_set_prune_last_layer(glc_engine* e, int on);

/* Length bucketing of glc_engine_forward (host buffers): the reference pads every row of a batch to the longest one
 * (/root/reference/src/tokenizer.c:44-54); rows are independent, so a ragged batch is run as up to `max_groups` groups of
 * similar length, each padded to its own longest row (results per row are unchanged; only padding work is saved).
 * Default 4 (env GLICLASS_LENGTH_BUCKETS), 1 = off.  The device-resident forward is never bucketed. */
int glc_engine_set_length_buckets(glc_engine* e, int max_groups);
/* The planner on its own (host only, no GPU): rows sorted longest first into `order` [B]; group g = order[cuts[g] .. cuts[g+1]);
 * cuts has *n_groups + 1 entries (caller provides B + 1).  Cost model (engine.hip): padded token rows rounded up to whole waves of
 * 256-row GEMM tiles over the CUs for the N = hidden projections, plus 1024 rows per group. */
int glc_plan_length_buckets(const int* lengths, int B, int max_groups, int hidden, int* order, int* cuts, int* n_groups);
int glc_debug_last_forward_groups(const glc_engine* e);   /* how many length groups the last glc_engine_forward ran as */

/* Device memory helpers so a host language can stage buffers without linking HIP itself. */
void* glc_device_malloc(glc_engine* e, size_t bytes);
void glc_device_free(glc_engine* e, void* p);
int glc_memcpy_h2d(glc_engine* e, void* dst, const void* src, size_t bytes);
int glc_memcpy_d2h(glc_engine* e, void* dst, const void* src, size_t bytes);

/* HIP-event timing on the engine stream (the stream every kernel is launched on). */
int glc_timer_start(glc_engine* e);
float glc_timer_stop_ms(glc_engine* e); /* records, synchronises, returns elapsed ms (<0 on error) */

/* Per-kernel-class profile: when enabled every launch is bracketed by HIP events.
 * glc_profile_read returns the number of classes and fills name/total_ms/launch counts. */
#define GLC_PROFILE_MAX 16
int glc_profile_enable(glc_engine* e, int on);
int glc_profile_read(glc_engine* e, const char** names, float* total_ms, int* launches, int max_n);

/* Diagnostics for parity tests: copy a hidden state of the LAST forward to host as fp32
 * [B,S,H]; which = 0 embeddings output, l+1 = output of layer l.  Only valid after
 * glc_debug_keep_hidden(e,1) was set before the forward. */
int glc_debug_keep_hidden(glc_engine* e, int on);
int glc_debug_get_hidden(glc_engine* e, int which, float* out, size_t out_elems);
/* Attention kernel choice: 0 auto, 1 straightforward (non-MFMA), 2 per-wave MFMA band kernel, 3 workgroup-shared band kernel. */
int glc_debug_set_attention_impl(glc_engine* e, int impl);

/* fp32 mode: the group-split pipeline (activations kept as [32 hi | 32 lo] f16 groups, every projection on the 256-tile LDS-DMA
 * kernel).  mode 0 off, 1 auto (default: forwards large enough to fill the chip), 2 whenever the shapes allow (tests). */
int glc_debug_set_group_split(glc_engine* e, int mode);
int glc_debug_last_forward_group_split(const glc_engine* e);
/* MX cross-term pipeline (docs/LOG_r01-r05.md §3e): every projection of the full layers as a_hi*w_hi in f16 MFMAs + both cross terms in ONE
 * block-scaled fp8 MFMA, on "GX" rows.  Needs the GX weight copies, i.e. an engine created under GLICLASS_MX=1 (selected) or =build. */
int glc_debug_set_mx(glc_engine* e, int on);
int glc_debug_last_forward_mx(const glc_engine* e);
int glc_debug_last_forward_mx_attention(const glc_engine* e);   /* 1: the last forward's attention ran on MX tiles (two MFMA times per product) */
int glc_debug_set_mx_attention(glc_engine* e, int on);
long long glc_debug_mx_weight_bytes(const glc_engine* e);  /* bytes of the GX weight copies (0 until a forward has taken the MX pipeline: they are built then) */
int glc_debug_set_mx2(glc_engine* e, int on);              /* developer builds (make DEV=1): bucket-space kernel (csrc/dev/attention_mx2.hip) / band kernel (default); -1 for on != 0 in the product library */     /* MX pipeline: attention on MX tiles (default) / on split-f16 units */
/* Developer: stop forwards after a stage and read workspace rows decoded to fp32 (engine.hip). */
int glc_debug_set_stop(glc_engine* e, int stage);
int glc_debug_read_workspace(glc_engine* e, int which, int rows, float* out);
/* Group-split pipeline: LayerNorm folded into the GEMMs around it (1, default: the producer writes raw rows + row statistics, the consumer
 * runs on weights with gamma folded in and finishes (LN(x) W^T + b) in its epilogue) or as kernels of its own (0). */
int glc_debug_set_ln_fused(glc_engine* e, int on);
int glc_debug_last_forward_ln_folded(const glc_engine* e);      /* 1: the last forward ran with the norm folded into its GEMMs (every mode) */
/* Precision budget of the default mode (developer; scripts/precision_budget.py): a set bit rounds one operand group of the group-split
 * pipeline to f16 by dropping its lo halves — numerically the cheaper kernel that never fetches them.  Bits 0-7: (A, W) of the QKV,
 * attention-output, FFN1, FFN2 projections; 8-13: attention Q, K, V^T, P, PQ rows, PK rows; 14: the residual rows. */
int glc_debug_set_precision_mask(glc_engine* e, int mask);
/* Host-buffer forwards that were repeated with the norms unfused because the folded forward came out non-finite (a raw residual
 * stream beyond the f16 operand range; engine.hip forward_one). */
int glc_debug_range_retries(const glc_engine* e);
/* fp8 range guard of the MX pipeline: host-buffer forwards repeated on the split-f16 kernels because an activation left the e4m3 range of the
 * operand images (|x| > 448); 1 once the engine has left the MX pipeline for good (two such host-buffer forwards in a row; a device-resident forward
 * whose Q / K / V tiles left the range, or a second one after the activation exponent was already lowered — glc_engine_sync's second answer) */
int glc_debug_fp8_range_retries(const glc_engine* e);
int glc_debug_fp8_range_sticky(const glc_engine* e);
/* ... and the exponent the engine's activation rows carry: 0, or -5 once a forward left the range (the guard's first answer: rows that hold
 * |x| up to 14336, the forward repeated on the MX pipeline; only what still leaves the range goes to the split-f16 kernels) */
int glc_debug_activation_exponent(const glc_engine* e);
/* 256-tile GEMM ring: full-line (operand-major) stages on / off, process-wide developer A/B switch; bit-identical results. */
int glc_debug_set_gemm_full_lines(int on);

/* clamp(bucket(q-k)+span, 0, 2span-1) for q-k in [-(S-1), S-1] at out[q-k+S-1] (float32 math as
 * torch).  Pure host function (no GPU needed). */
void glc_delta_table(int S, int bucket_size, int max_position, int32_t* out);

/* Developer microbenchmark of one GEMM shape (16-bit engines): ms per launch, <0 on error. */
float glc_debug_gemm_bench(glc_engine* e, int M, int N, int K, int epi, int iters, int which);
/* Developer check: MX cross-term GEMM against the split-f16 GEMM on the same random operands (engine.hip). */
int glc_debug_gemm_mx_check(glc_engine* e, int M, int N, int K, float a_amp, float w_amp, int mode, double* out);
/* Developer microbenchmark of the band attention kernel on the workspace of the last forward (see engine.hip). */
float glc_debug_attn_bench(glc_engine* e, int iters, int variant, int stamps, double* checksum);
int glc_debug_is_developer_build(void);                  /* 1: built with make DEV=1 (developer kernels, stamps, GLC_* switches); 0: the product library */

const glc_model_config* glc_engine_config(const glc_engine* e);
int glc_engine_dtype(const glc_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* GLICLASS_HIP_H */
