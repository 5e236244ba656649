/*
 * ORACLE — test infrastructure only.  Never linked, imported or executed by the product path
 * (the gliclass/c_amd package and include/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / CPU baseline.
 *
 * What it is: a plain C11 + OpenMP fp32 restatement of what `g_ort->Run()` computes at
 * /root/reference/src/model.c:173-182 for a GLiClass uni-encoder graph, i.e. the ONNX export
 * (/root/reference/ONNX_CONVERTING/convert_to_onnx.py:71-79) of HF DeBERTa-v2/v3 + the GLiClass
 * head.  The arithmetic lives in third-party code that is NOT under /root/reference:
 *   - transformers (unpinned in the reference; 5.15.0 in the build image):
 *     models/deberta_v2/modeling_deberta_v2.py — cited below as HF:<line>
 *   - gliclass (PyPI, unpinned; unavailable offline) — head restated per SURVEY.md §8a row a12.
 *
 * PARITY STATUS: encoder pinned against transformers' DebertaV2Model through the committed
 * fixtures in tests/golden (made by oracle/gen_golden.py); the head is "parity unpinned"
 * (no gliclass package / no golden logits available offline — SURVEY.md §8c).
 *
 * glo_forward_decoder restates the decoder-style backbone of BASELINE.json configs[4] after
 * transformers' models/qwen2/modeling_qwen2.py — cited below as Q2:<line>; pinned against Qwen2Model through
 * tests/golden/dec_*.npz (causal mask).  The bidirectional variant (causal = 0) and the head on top of a decoder are
 * "parity unpinned" (upstream gliclass wraps decoders itself; package unavailable).
 *
 * The formulation is deliberately the LITERAL one of the HF code (dense c2p/p2c score matrices,
 * gather by clamp(rel+span), masked_fill(finfo.min), division by sqrt(3d)) so that it is an
 * independent check of the re-derived Toeplitz/band formulation the HIP kernels use.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t vocab, hidden, layers, heads, head_dim, inter, pos_buckets, max_rel_pos;
    int32_t pad_id, class_token_index, embed_class_token, pooling, normalize_features;
    float ln_eps, logit_scale;
    /* decoder backbone (glo_forward_decoder): key/value heads, causal flag, RoPE base; ln_eps = rms_norm_eps */
    int32_t backbone, kv_heads, causal;
    float rope_theta;
    int32_t scorer;    /* 0 'simple' (dot), 1 'weighted-dot', 2 'mlp' (upstream gliclass scorers; restated, parity unpinned) */
} glo_config;

/* tensor order = gliclass/c_amd/weights.py::tensor_specs */
enum { T_WORD = 0, T_ELN_W, T_ELN_B, T_REL, T_RLN_W, T_RLN_B, T_LAYER0 };
enum { L_QW = 0, L_QB, L_KW, L_KB, L_VW, L_VB, L_OW, L_OB, L_LN1W, L_LN1B, L_IW, L_IB, L_DW, L_DB, L_LN2W, L_LN2B, L_N };
enum { H_T1W = 0, H_T1B, H_T2W, H_T2B, H_C1W, H_C1B, H_C2W, H_C2B, H_SCORER };
#define GLO_SCORER_MLP_HIDDEN 256

/* ---- HF:57-69 make_log_bucket_position, float32 arithmetic exactly as torch does it ---- */
static int glo_bucket(int rel, int bucket_size, int max_position) {
    if (bucket_size <= 0 || max_position <= 0) return rel;
    int mid = bucket_size / 2;
    int sign = (rel > 0) - (rel < 0);
    int abs_pos = (rel < mid && rel > -mid) ? mid - 1 : abs(rel);
    if (abs_pos <= mid) return rel;
    float x = (float)abs_pos / (float)mid;
    float den = logf((float)(((double)max_position - 1.0) / (double)mid));
    float log_pos = ceilf(logf(x) / den * (float)(mid - 1)) + (float)mid;
    return (int)(log_pos * (float)sign);
}

/* table over rel = q-k in [-(S-1), S-1], index rel+S-1 : clamp(bucket+span, 0, 2span-1)  (HF:318) */
void glo_delta_table(int S, int bucket_size, int max_position, int32_t* out) {
    int span = bucket_size > 0 ? bucket_size : max_position;
    for (int r = -(S - 1); r <= S - 1; ++r) {
        int v = glo_bucket(r, bucket_size, max_position) + span;
        if (v < 0) v = 0;
        if (v > 2 * span - 1) v = 2 * span - 1;
        out[r + S - 1] = v;
    }
}

/* C[M,N] = A[M,K] * W[N,K]^T + bias   (nn.Linear) */
static void linear(const float* A, int M, int K, const float* W, const float* bias, int N, float* C) {
    float* Wt = (float*)malloc((size_t)K * N * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) Wt[(size_t)k * N + n] = W[(size_t)n * K + k];
#pragma omp parallel for schedule(dynamic, 1)
    for (int m0 = 0; m0 < M; m0 += 4) {
        int mr = M - m0 < 4 ? M - m0 : 4;
        float* c[4];
        const float* a[4];
        for (int i = 0; i < 4; ++i) {
            int mi = m0 + (i < mr ? i : 0);
            c[i] = C + (size_t)mi * N;
            a[i] = A + (size_t)mi * K;
        }
        for (int i = 0; i < mr; ++i)
            for (int n = 0; n < N; ++n) c[i][n] = bias ? bias[n] : 0.f;
        if (mr == 4) {
            for (int k = 0; k < K; ++k) {
                const float* w = Wt + (size_t)k * N;
                float a0 = a[0][k], a1 = a[1][k], a2 = a[2][k], a3 = a[3][k];
                float *c0 = c[0], *c1 = c[1], *c2 = c[2], *c3 = c[3];
                for (int n = 0; n < N; ++n) {
                    float wv = w[n];
                    c0[n] += a0 * wv; c1[n] += a1 * wv; c2[n] += a2 * wv; c3[n] += a3 * wv;
                }
            }
        } else {
            for (int i = 0; i < mr; ++i)
                for (int k = 0; k < K; ++k) {
                    const float* w = Wt + (size_t)k * N;
                    float av = a[i][k];
                    for (int n = 0; n < N; ++n) c[i][n] += av * w[n];
                }
        }
    }
    free(Wt);
}

/* torch.nn.LayerNorm: biased variance, eps inside sqrt (HF:550, :52, :411) */
static void layernorm_rows(const float* X, int M, int H, const float* g, const float* b, float eps, float* Y) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* x = X + (size_t)m * H;
        float* y = Y + (size_t)m * H;
        double s = 0, ss = 0;
        for (int i = 0; i < H; ++i) s += x[i];
        float mean = (float)(s / H);
        for (int i = 0; i < H; ++i) { double d = x[i] - mean; ss += d * d; }
        float rstd = 1.0f / sqrtf((float)(ss / H) + eps);
        for (int i = 0; i < H; ++i) y[i] = (x[i] - mean) * rstd * g[i] + b[i];
    }
}

static inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

/* ---- GLiClass head (parity unpinned): X [B,S,H] final hidden states -> logits ---- */
static void glo_head(const glo_config* cfg, const float* const* hw, const float* X, const int64_t* ids, const int64_t* mask,
                     int B, int S, float* logits, int c_alloc, int* c_out) {
    const int H = cfg->hidden;
    int C = 0;
    for (int b = 0; b < B; ++b) {
        int c = 0;
        for (int s = 0; s < S; ++s) c += ids[(size_t)b * S + s] == cfg->class_token_index;
        if (c > C) C = c;
    }
    if (c_out) *c_out = C;
    if (C > c_alloc) C = c_alloc;
    int rows = B * (1 + C);
    float* G = (float*)calloc((size_t)rows * H, sizeof(float));
    float* G1 = (float*)malloc((size_t)rows * H * sizeof(float));
    float* G2 = (float*)malloc((size_t)rows * H * sizeof(float));
    for (int b = 0; b < B; ++b) {
        float* pooled = G + (size_t)b * H;
        if (cfg->pooling == 0) memcpy(pooled, X + (size_t)b * S * H, (size_t)H * sizeof(float));
        else if (cfg->pooling == 2) {          /* last attended token (decoder backbones) */
            int last = 0;
            for (int s = 0; s < S; ++s) if (mask[(size_t)b * S + s]) last = s;
            memcpy(pooled, X + ((size_t)b * S + last) * H, (size_t)H * sizeof(float));
        } else {                                /* 'avg': mean over the attended positions */
            int n = 0;
            for (int s = 0; s < S; ++s) {
                if (!mask[(size_t)b * S + s]) continue;
                ++n;
                for (int i = 0; i < H; ++i) pooled[i] += X[((size_t)b * S + s) * H + i];
            }
            for (int i = 0; i < H; ++i) pooled[i] = n ? pooled[i] / (float)n : 0.f;
        }
        int j = 0;
        for (int s = 0; s < S && j < C; ++s)
            if (ids[(size_t)b * S + s] == cfg->class_token_index) {
                int pos = cfg->embed_class_token ? s : (s + 1 < S ? s + 1 : s);
                memcpy(G + ((size_t)B + (size_t)b * C + j) * H, X + ((size_t)b * S + pos) * H, (size_t)H * sizeof(float));
                ++j;
            }
    }
    /* text projector on rows [0,B), classes projector on rows [B, B+B*C) */
    linear(G, B, H, hw[H_T1W], hw[H_T1B], H, G1);
    for (size_t i = 0; i < (size_t)B * H; ++i) G1[i] = gelu_erf(G1[i]);
    linear(G1, B, H, hw[H_T2W], hw[H_T2B], H, G2);
    if (C > 0) {
        linear(G + (size_t)B * H, B * C, H, hw[H_C1W], hw[H_C1B], H, G1 + (size_t)B * H);
        for (size_t i = (size_t)B * H; i < (size_t)rows * H; ++i) G1[i] = gelu_erf(G1[i]);
        linear(G1 + (size_t)B * H, B * C, H, hw[H_C2W], hw[H_C2B], H, G2 + (size_t)B * H);
    }
    if (cfg->normalize_features)
        for (int r = 0; r < rows; ++r) {
            float nn = 0;
            for (int i = 0; i < H; ++i) nn += G2[(size_t)r * H + i] * G2[(size_t)r * H + i];
            nn = sqrtf(nn) + 1e-8f;
            for (int i = 0; i < H; ++i) G2[(size_t)r * H + i] /= nn;
        }
    const float* const* sw = hw + H_SCORER;
    if (cfg->scorer == 1 && C > 0) {
        /* ScorerWeightedDot (upstream gliclass scorers.py, restated): (t1|t2) = proj_text(text) [2H], (c1|c2) = proj_label(class) [2H];
         * score = out_mlp([t1, c1, t2 * c2]), out_mlp = Linear(3H, 4H) -> (Dropout) -> ReLU -> Linear(4H, 1) */
        float* T = (float*)malloc((size_t)B * 2 * H * sizeof(float));
        float* L = (float*)malloc((size_t)B * C * 2 * H * sizeof(float));
        float* cat = (float*)malloc((size_t)B * C * 3 * H * sizeof(float));
        float* hid = (float*)malloc((size_t)B * C * 4 * H * sizeof(float));
        linear(G2, B, H, sw[0], sw[1], 2 * H, T);
        linear(G2 + (size_t)B * H, B * C, H, sw[2], sw[3], 2 * H, L);
        for (int b = 0; b < B; ++b)
            for (int j = 0; j < C; ++j) {
                const float* t = T + (size_t)b * 2 * H;
                const float* l = L + ((size_t)b * C + j) * 2 * H;
                float* o = cat + ((size_t)b * C + j) * 3 * H;
                for (int i = 0; i < H; ++i) { o[i] = t[i]; o[H + i] = l[i]; o[2 * H + i] = t[H + i] * l[H + i]; }
            }
        linear(cat, B * C, 3 * H, sw[4], sw[5], 4 * H, hid);
        for (int b = 0; b < B; ++b)
            for (int j = 0; j < C; ++j) {
                const float* hrow = hid + ((size_t)b * C + j) * 4 * H;
                float a = 0;
                for (int i = 0; i < 4 * H; ++i) a += (hrow[i] > 0.f ? hrow[i] : 0.f) * sw[6][i];
                logits[(size_t)b * c_alloc + j] = (a + sw[7][0]) * (cfg->normalize_features ? cfg->logit_scale : 1.0f);
            }
        free(T); free(L); free(cat); free(hid);
    } else if (cfg->scorer == 2 && C > 0) {
        /* MLPScorer (restated): Linear(2H, 256) -> ReLU -> Linear(256, 128) -> ReLU -> Linear(128, 1) on [text, class] */
        const int Mh = GLO_SCORER_MLP_HIDDEN;
        float* cat = (float*)malloc((size_t)B * C * 2 * H * sizeof(float));
        float* h1 = (float*)malloc((size_t)B * C * Mh * sizeof(float));
        float* h2 = (float*)malloc((size_t)B * C * (Mh / 2) * sizeof(float));
        for (int b = 0; b < B; ++b)
            for (int j = 0; j < C; ++j) {
                float* o = cat + ((size_t)b * C + j) * 2 * H;
                memcpy(o, G2 + (size_t)b * H, (size_t)H * sizeof(float));
                memcpy(o + H, G2 + ((size_t)B + (size_t)b * C + j) * H, (size_t)H * sizeof(float));
            }
        linear(cat, B * C, 2 * H, sw[0], sw[1], Mh, h1);
        for (size_t i = 0; i < (size_t)B * C * Mh; ++i) h1[i] = h1[i] > 0.f ? h1[i] : 0.f;
        linear(h1, B * C, Mh, sw[2], sw[3], Mh / 2, h2);
        for (int r = 0; r < B * C; ++r) {
            float a = 0;
            for (int i = 0; i < Mh / 2; ++i) { const float v = h2[(size_t)r * (Mh / 2) + i]; a += (v > 0.f ? v : 0.f) * sw[4][i]; }
            logits[(size_t)(r / C) * c_alloc + (r % C)] = (a + sw[5][0]) * (cfg->normalize_features ? cfg->logit_scale : 1.0f);
        }
        free(cat); free(h1); free(h2);
    } else
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < C; ++j) {
            float a = 0;
            for (int i = 0; i < H; ++i) a += G2[(size_t)b * H + i] * G2[((size_t)B + (size_t)b * C + j) * H + i];
            if (cfg->normalize_features) a *= cfg->logit_scale;
            logits[(size_t)b * c_alloc + j] = a;
        }
    free(G); free(G1); free(G2);
}

/*
 * One forward.  tensors[] in tensor_specs order.  logits: [B, c_alloc] row-major with stride
 * c_alloc (only first *c_out columns valid).  hidden_dump (optional): [(L+1), B, S, H].
 * scores_dump (optional): pre-softmax masked scores of layer 0, batch 0, head 0: [S,S].
 * returns 0 on success.
 */
int glo_forward(const glo_config* cfg, const float* const* tensors, const int64_t* ids, const int64_t* mask,
                int B, int S, float* logits, int c_alloc, int* c_out, float* hidden_dump, float* scores_dump) {
    const int H = cfg->hidden, nh = cfg->heads, d = cfg->head_dim, I = cfg->inter, L = cfg->layers;
    const int span = cfg->pos_buckets > 0 ? cfg->pos_buckets : cfg->max_rel_pos, P = 2 * span;
    const int M = B * S;
    if (H != nh * d) return -1;
    float* X = (float*)malloc((size_t)M * H * sizeof(float));
    float* Q = (float*)malloc((size_t)M * H * sizeof(float));
    float* Kb = (float*)malloc((size_t)M * H * sizeof(float));
    float* V = (float*)malloc((size_t)M * H * sizeof(float));
    float* CTX = (float*)malloc((size_t)M * H * sizeof(float));
    float* T1 = (float*)malloc((size_t)M * H * sizeof(float));
    float* H1 = (float*)malloc((size_t)M * H * sizeof(float));
    float* FF = (float*)malloc((size_t)M * I * sizeof(float));
    float* R = (float*)malloc((size_t)P * H * sizeof(float));
    float* PK = (float*)malloc((size_t)P * H * sizeof(float));
    float* PQ = (float*)malloc((size_t)P * H * sizeof(float));
    int32_t* dtab = (int32_t*)malloc((size_t)(2 * S - 1) * sizeof(int32_t));
    if (!X || !Q || !Kb || !V || !CTX || !T1 || !H1 || !FF || !R || !PK || !PQ || !dtab) return -2;

    /* embeddings: HF:533 gather, :550 LayerNorm, :552-559 * mask (position_biased_input=false, no token types) */
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        int64_t id = ids[m];
        if (id < 0 || id >= cfg->vocab) id = cfg->pad_id;
        memcpy(T1 + (size_t)m * H, tensors[T_WORD] + (size_t)id * H, (size_t)H * sizeof(float));
    }
    layernorm_rows(T1, M, H, tensors[T_ELN_W], tensors[T_ELN_B], cfg->ln_eps, X);
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float mk = (float)mask[m];
        for (int i = 0; i < H; ++i) X[(size_t)m * H + i] *= mk;
    }
    if (hidden_dump) memcpy(hidden_dump, X, (size_t)M * H * sizeof(float));

    /* HF:595-599 rel_embeddings -> LayerNorm ; HF:72-101 relative positions (depend on q-k only) */
    layernorm_rows(tensors[T_REL], P, H, tensors[T_RLN_W], tensors[T_RLN_B], cfg->ln_eps, R);
    glo_delta_table(S, cfg->pos_buckets, cfg->max_rel_pos, dtab);
    const float scale = sqrtf((float)d * 3.0f); /* HF:237-242, scale_factor = 1 + c2p + p2c */

    for (int l = 0; l < L; ++l) {
        const float* const* w = tensors + T_LAYER0 + l * L_N;
        linear(X, M, H, w[L_QW], w[L_QB], H, Q);   /* HF:231-233 */
        linear(X, M, H, w[L_KW], w[L_KB], H, Kb);
        linear(X, M, H, w[L_VW], w[L_VB], H, V);
        linear(R, P, H, w[L_KW], w[L_KB], H, PK);  /* HF:296-302 share_att_key: pos_key = key_proj(rel) */
        linear(R, P, H, w[L_QW], w[L_QB], H, PQ);  /*                          pos_query = query_proj(rel) */
#pragma omp parallel
        {
            float* sc = (float*)malloc((size_t)S * S * sizeof(float));
            float* c2p = (float*)malloc((size_t)S * P * sizeof(float));
            float* p2c = (float*)malloc((size_t)S * P * sizeof(float));
#pragma omp for schedule(dynamic, 1) collapse(2)
            for (int b = 0; b < B; ++b)
                for (int h = 0; h < nh; ++h) {
                    const float* q = Q + (size_t)b * S * H + h * d;
                    const float* k = Kb + (size_t)b * S * H + h * d;
                    const float* v = V + (size_t)b * S * H + h * d;
                    const int64_t* mk = mask + (size_t)b * S;
                    /* c2p_att[q,p] = Q_q . PK_p (HF:317) ; p2c_att[k,p] = K_k . PQ_p (HF:337) */
                    for (int i = 0; i < S; ++i)
                        for (int p = 0; p < P; ++p) {
                            const float* pk = PK + (size_t)p * H + h * d;
                            const float* pq = PQ + (size_t)p * H + h * d;
                            float a = 0, c = 0;
                            for (int e = 0; e < d; ++e) { a += q[(size_t)i * H + e] * pk[e]; c += k[(size_t)i * H + e] * pq[e]; }
                            c2p[(size_t)i * P + p] = a;
                            p2c[(size_t)i * P + p] = c;
                        }
                    for (int i = 0; i < S; ++i) {
                        float* row = sc + (size_t)i * S;
                        float mx = -FLT_MAX;
                        for (int j = 0; j < S; ++j) {
                            float a = 0;
                            for (int e = 0; e < d; ++e) a += q[(size_t)i * H + e] * (k[(size_t)j * H + e] / scale); /* HF:243 */
                            /* c2p: gather index clamp(rel[i,j]+span) (HF:318-323) */
                            int ci = dtab[i - j + S - 1];
                            /* p2c: index clamp(-rel[j,i]+span) on row j, then transposed (HF:336-342) */
                            int pj = -glo_bucket(j - i, cfg->pos_buckets, cfg->max_rel_pos) + span;
                            if (pj < 0) pj = 0;
                            if (pj > P - 1) pj = P - 1;
                            a += c2p[(size_t)i * P + ci] / scale + p2c[(size_t)j * P + pj] / scale; /* HF:251 */
                            if (!(mk[i] && mk[j])) a = -FLT_MAX;                                    /* HF:256-257 */
                            row[j] = a;
                            if (a > mx) mx = a;
                        }
                        if (scores_dump && l == 0 && b == 0 && h == 0) memcpy(scores_dump + (size_t)i * S, row, (size_t)S * sizeof(float));
                        float sum = 0;
                        for (int j = 0; j < S; ++j) { row[j] = expf(row[j] - mx); sum += row[j]; } /* HF:259 */
                        float inv = 1.0f / sum;
                        float* o = CTX + ((size_t)b * S + i) * H + h * d;
                        for (int e = 0; e < d; ++e) o[e] = 0;
                        for (int j = 0; j < S; ++j) {
                            float pv = row[j] * inv;
                            const float* vr = v + (size_t)j * H;
                            for (int e = 0; e < d; ++e) o[e] += pv * vr[e];                         /* HF:262 */
                        }
                    }
                }
            free(sc); free(c2p); free(p2c);
        }
        /* HF:49-53 SelfOutput: LN(dense(ctx) + X) */
        linear(CTX, M, H, w[L_OW], w[L_OB], H, T1);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)M * H; ++i) T1[i] += X[i];
        layernorm_rows(T1, M, H, w[L_LN1W], w[L_LN1B], cfg->ln_eps, H1);
        /* HF:393-396 Intermediate (gelu erf), HF:408-412 Output: LN(dense(f) + h1) */
        linear(H1, M, H, w[L_IW], w[L_IB], I, FF);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)M * I; ++i) FF[i] = gelu_erf(FF[i]);
        linear(FF, M, I, w[L_DW], w[L_DB], H, T1);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)M * H; ++i) T1[i] += H1[i];
        layernorm_rows(T1, M, H, w[L_LN2W], w[L_LN2B], cfg->ln_eps, X);
        if (hidden_dump) memcpy(hidden_dump + (size_t)(l + 1) * M * H, X, (size_t)M * H * sizeof(float));
    }

    glo_head(cfg, tensors + T_LAYER0 + L * L_N, X, ids, mask, B, S, logits, c_alloc, c_out);
    free(X); free(Q); free(Kb); free(V); free(CTX); free(T1); free(H1); free(FF); free(R); free(PK); free(PQ); free(dtab);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Decoder-style backbone (Qwen2 arithmetic).  tensors[]: embed_tokens, per layer {input_layernorm, q.w q.b k.w k.b v.w v.b,
 * o.w, post_attention_layernorm, gate.w up.w down.w}, norm, then the 8 head tensors (weights.py::tensor_specs).
 * hidden_dump (optional): [(L+1),B,S,H] = embeddings, output of layers 0..L-2, and norm(output of layer L-1) — the
 * tuple HF returns with output_hidden_states=True.
 * ------------------------------------------------------------------------------------------------------------------ */
enum { D_LN1 = 0, D_QW, D_QB, D_KW, D_KB, D_VW, D_VB, D_OW, D_LN2, D_GW, D_UW, D_DW, D_N };

/* Q2:247-252 */
static void rmsnorm_rows(const float* X, int M, int H, const float* w, float eps, float* Y) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* x = X + (size_t)m * H;
        float* y = Y + (size_t)m * H;
        double ss = 0;
        for (int i = 0; i < H; ++i) ss += (double)x[i] * x[i];
        float r = 1.0f / sqrtf((float)(ss / H) + eps);
        for (int i = 0; i < H; ++i) y[i] = w[i] * (x[i] * r);
    }
}

int glo_forward_decoder(const glo_config* cfg, const float* const* tensors, const int64_t* ids, const int64_t* mask,
                        int B, int S, float* logits, int c_alloc, int* c_out, float* hidden_dump) {
    const int H = cfg->hidden, nq = cfg->heads, d = cfg->head_dim, I = cfg->inter, L = cfg->layers;
    const int nkv = cfg->kv_heads > 0 ? cfg->kv_heads : nq, grp = nq / nkv, NQ = nq * d, NKV = nkv * d, hd2 = d / 2;
    const int M = B * S;
    if (NQ != H && 0) return -1;
    if (nq % nkv || d % 2) return -1;
    float* X = (float*)malloc((size_t)M * H * sizeof(float));
    float* Hn = (float*)malloc((size_t)M * H * sizeof(float));
    float* Q = (float*)malloc((size_t)M * NQ * sizeof(float));
    float* Kb = (float*)malloc((size_t)M * NKV * sizeof(float));
    float* V = (float*)malloc((size_t)M * NKV * sizeof(float));
    float* CTX = (float*)malloc((size_t)M * NQ * sizeof(float));
    float* T1 = (float*)malloc((size_t)M * H * sizeof(float));
    float* G = (float*)malloc((size_t)M * I * sizeof(float));
    float* U = (float*)malloc((size_t)M * I * sizeof(float));
    float* cs = (float*)malloc((size_t)S * hd2 * 2 * sizeof(float));
    if (!X || !Hn || !Q || !Kb || !V || !CTX || !T1 || !G || !U || !cs) return -2;

    /* Q2:384 inputs_embeds = embed_tokens(input_ids) — no mask multiply, no norm */
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        int64_t id = ids[m];
        if (id < 0 || id >= cfg->vocab) id = cfg->pad_id;
        memcpy(X + (size_t)m * H, tensors[0] + (size_t)id * H, (size_t)H * sizeof(float));
    }
    if (hidden_dump) memcpy(hidden_dump, X, (size_t)M * H * sizeof(float));

    /* Q2:86 inv_freq = 1 / base^(arange(0,d,2)/d) in float32; Q2:97-100 freqs = inv_freq * position, cos/sin (position_ids =
     * arange(S), independent of padding) */
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < hd2; ++i) {
            float inv = 1.0f / powf(cfg->rope_theta, (float)(2 * i) / (float)d);
            float f = inv * (float)s;
            cs[((size_t)s * hd2 + i) * 2] = cosf(f);
            cs[((size_t)s * hd2 + i) * 2 + 1] = sinf(f);
        }
    const float scaling = 1.0f / sqrtf((float)d);   /* Q2:186 */

    for (int l = 0; l < L; ++l) {
        const float* const* w = tensors + 1 + l * D_N;
        rmsnorm_rows(X, M, H, w[D_LN1], cfg->ln_eps, Hn);            /* Q2:280 */
        linear(Hn, M, H, w[D_QW], w[D_QB], NQ, Q);                   /* Q2:206-208 */
        linear(Hn, M, H, w[D_KW], w[D_KB], NKV, Kb);
        linear(Hn, M, H, w[D_VW], w[D_VB], NKV, V);
        /* Q2:105-109,133-134 rotate_half form: out = x*cos + rotate_half(x)*sin, cos/sin duplicated over the two halves */
#pragma omp parallel for schedule(static)
        for (int m = 0; m < M; ++m) {
            const int s = m % S;
            for (int pass = 0; pass < 2; ++pass) {
                float* base = pass == 0 ? Q + (size_t)m * NQ : Kb + (size_t)m * NKV;
                const int nh_ = pass == 0 ? nq : nkv;
                for (int h = 0; h < nh_; ++h)
                    for (int i = 0; i < hd2; ++i) {
                        float c = cs[((size_t)s * hd2 + i) * 2], sn = cs[((size_t)s * hd2 + i) * 2 + 1];
                        float x1 = base[h * d + i], x2 = base[h * d + i + hd2];
                        base[h * d + i] = x1 * c - x2 * sn;
                        base[h * d + i + hd2] = x2 * c + x1 * sn;
                    }
            }
        }
        /* Q2:160-170 repeat_kv, scores * scaling + mask, softmax fp32, * V; mask = causal AND key-padding (create_causal_mask) */
#pragma omp parallel
        {
            float* row = (float*)malloc((size_t)S * sizeof(float));
#pragma omp for schedule(dynamic, 1) collapse(2)
            for (int b = 0; b < B; ++b)
                for (int h = 0; h < nq; ++h) {
                    const int kvh = h / grp;
                    const int64_t* mk = mask + (size_t)b * S;
                    for (int i = 0; i < S; ++i) {
                        const float* q = Q + ((size_t)b * S + i) * NQ + h * d;
                        float mx = -FLT_MAX;
                        for (int j = 0; j < S; ++j) {
                            float a;
                            if ((cfg->causal && j > i) || !mk[j]) a = -FLT_MAX;
                            else {
                                const float* k = Kb + ((size_t)b * S + j) * NKV + kvh * d;
                                a = 0;
                                for (int e = 0; e < d; ++e) a += q[e] * k[e];
                                a *= scaling;
                            }
                            row[j] = a;
                            if (a > mx) mx = a;
                        }
                        float sum = 0;
                        for (int j = 0; j < S; ++j) { row[j] = expf(row[j] - mx); sum += row[j]; }
                        float inv = 1.0f / sum;
                        float* o = CTX + ((size_t)b * S + i) * NQ + h * d;
                        for (int e = 0; e < d; ++e) o[e] = 0;
                        for (int j = 0; j < S; ++j) {
                            float pv = row[j] * inv;
                            if (pv == 0.f) continue;
                            const float* vr = V + ((size_t)b * S + j) * NKV + kvh * d;
                            for (int e = 0; e < d; ++e) o[e] += pv * vr[e];
                        }
                    }
                }
            free(row);
        }
        linear(CTX, M, NQ, w[D_OW], NULL, H, T1);                    /* Q2:233 o_proj (no bias) */
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)M * H; ++i) X[i] += T1[i];    /* Q2:291 */
        rmsnorm_rows(X, M, H, w[D_LN2], cfg->ln_eps, Hn);            /* Q2:295 */
        linear(Hn, M, H, w[D_GW], NULL, I, G);                       /* Q2:47 down(silu(gate(x)) * up(x)) */
        linear(Hn, M, H, w[D_UW], NULL, I, U);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)M * I; ++i) { float g = G[i]; G[i] = g / (1.0f + expf(-g)) * U[i]; }
        linear(G, M, I, w[D_DW], NULL, H, T1);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < (size_t)M * H; ++i) X[i] += T1[i];
        if (hidden_dump && l + 1 < L) memcpy(hidden_dump + (size_t)(l + 1) * M * H, X, (size_t)M * H * sizeof(float));
    }
    rmsnorm_rows(X, M, H, tensors[1 + L * D_N], cfg->ln_eps, Hn);    /* Q2:398 final norm */
    if (hidden_dump) memcpy(hidden_dump + (size_t)L * M * H, Hn, (size_t)M * H * sizeof(float));
    glo_head(cfg, tensors + 2 + L * D_N, Hn, ids, mask, B, S, logits, c_alloc, c_out);
    free(X); free(Hn); free(Q); free(Kb); free(V); free(CTX); free(T1); free(G); free(U); free(cs);
    return 0;
}


/* /root/reference/src/postprocessor.c:14-16 */
float glo_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

/* Caps the OpenMP team (oracle_c.py passes the CPUs this process may really use: affinity and cgroup quota). */
void glo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int glo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
