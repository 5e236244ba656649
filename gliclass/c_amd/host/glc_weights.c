/*
 * Weight source for create_ort_session(): reads a .glcw blob (written by gliclass/c_amd/weights.py)
 * or synthesises a model from "synthetic:<config>[:seed[:scorer]]".  Stands in for the ONNX file load inside
 * g_ort->CreateSession (/root/reference/src/model.c:269, path from /root/reference/include/paths.h:5).
 */
#include "glc_weights.h"

#include <fcntl.h>
#include <math.h>
#include "glc_cpus.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define GLCW_HEADER_BYTES 256
#define GLCW_REC_BYTES 160

uint64_t glc_fnv1a64(const char* s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 0x100000001B3ull; }
    return h;
}

void glc_prng_fill(uint64_t seed, const char* name, size_t n, double amp, double mean, float* out) {
    const uint64_t base = glc_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15ull);
    /* team sized to the CPUs this process may really use (affinity + cgroup quota): the OpenMP default — every core of the host —
     * took 25 s for the base model on a 16-CPU share of a large box (oversubscribed spin-waiting teams), 0.2 s with the clamp */
#pragma omp parallel for schedule(static) num_threads(glc_host_cpus()) if (n > 65536)
    for (size_t i = 0; i < n; ++i) {
        uint64_t z = base + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
        out[i] = (float)((2.0 * u - 1.0) * amp + mean);
    }
}

int glc_named_config(const char* name, glc_model_config* c) {
    static const struct { const char* n; int vocab, hidden, layers, heads, inter; } T[] = {
        {"tiny", 515, 128, 2, 2, 256},         {"mini", 1027, 256, 3, 4, 512},
        {"small", 128003, 768, 6, 12, 3072},   {"base", 128003, 768, 12, 12, 3072},
        {"large", 128003, 1024, 24, 16, 4096},
    };
    for (size_t i = 0; i < sizeof(T) / sizeof(T[0]); ++i)
        if (strcmp(name, T[i].n) == 0) {
            memset(c, 0, sizeof(*c));
            c->vocab = T[i].vocab; c->hidden = T[i].hidden; c->layers = T[i].layers; c->heads = T[i].heads;
            c->head_dim = 64; c->inter = T[i].inter; c->pos_buckets = 256; c->max_rel_pos = 512;
            c->pad_id = 0; c->cls_id = 1; c->sep_id = 2;
            c->class_token_index = c->vocab - 2; c->text_token_index = c->vocab - 1;
            c->pooling = GLC_POOL_FIRST; c->scorer = GLC_SCORER_DOT; c->embed_class_token = 1; c->normalize_features = 0;
            c->backbone = GLC_BACKBONE_DEBERTA; c->kv_heads = c->heads; c->causal = 1; c->rope_theta = 1.0e6f;
            c->ln_eps = 1e-7f; c->logit_scale = 1.0f;
            return 0;
        }
    /* decoder-style backbones (config.py CONFIGS "dec-tiny", "dec-mini", "qwen-1.5b") */
    static const struct { const char* n; int vocab, hidden, layers, heads, kv, inter; } D[] = {
        {"dec-tiny", 515, 256, 2, 2, 1, 512}, {"dec-mini", 1027, 512, 3, 4, 2, 768}, {"qwen-1.5b", 151648, 1536, 28, 12, 2, 8960},
    };
    for (size_t i = 0; i < sizeof(D) / sizeof(D[0]); ++i)
        if (strcmp(name, D[i].n) == 0) {
            memset(c, 0, sizeof(*c));
            c->vocab = D[i].vocab; c->hidden = D[i].hidden; c->layers = D[i].layers; c->heads = D[i].heads;
            c->head_dim = 128; c->inter = D[i].inter; c->pos_buckets = 256; c->max_rel_pos = 512;
            c->pad_id = 0; c->cls_id = 1; c->sep_id = 2;
            c->class_token_index = c->vocab - 2; c->text_token_index = c->vocab - 1;
            c->pooling = GLC_POOL_LAST; c->scorer = GLC_SCORER_DOT; c->embed_class_token = 1; c->normalize_features = 0;
            c->backbone = GLC_BACKBONE_DECODER; c->kv_heads = D[i].kv; c->causal = 1; c->rope_theta = 1.0e6f;
            c->ln_eps = 1e-6f; c->logit_scale = 1.0f;
            return 0;
        }
    return -1;
}

static double lin_amp(double t, double fan_in) { return sqrt(3.0) * t / sqrt(fan_in); }

/* the head's tensors (index k inside the head): the two FeaturesProjectors, then the scorer's own (include/gliclass_hip.h) */
static int head_spec(const glc_model_config* c, int k, char* name, uint64_t shape[4], double* amp, double* mean) {
    const uint64_t H = (uint64_t)c->hidden;
    char buf[96];
#define SPEC1(nm, n0, a, m) do { snprintf(name, 96, "%s", nm); shape[0] = (n0); *amp = (a); *mean = (m); return 1; } while (0)
#define SPEC2(nm, n0, n1, a) do { snprintf(name, 96, "%s", nm); shape[0] = (n0); shape[1] = (n1); *amp = (a); return 2; } while (0)
    if (k < 0 || k >= GLC_TENSORS_HEAD + glc_num_scorer_tensors(c->scorer)) return -1;
    if (k < GLC_TENSORS_HEAD) {
        const double t2 = sqrt(1.5 / sqrt((double)H)) / 0.7;
        snprintf(buf, sizeof buf, "%s.linear_%d.%s", k < 4 ? "text_projector" : "classes_projector", (k % 4) / 2 + 1,
                 (k % 2) ? "bias" : "weight");
        switch (k % 4) {
            case 0: SPEC2(buf, H, H, lin_amp(1.0, (double)H));
            case 1: SPEC1(buf, H, 0.1, 0.0);
            case 2: SPEC2(buf, H, H, lin_amp(t2, (double)H));
            default: SPEC1(buf, H, 0.02, 0.0);
        }
    }
    const int j = k - GLC_TENSORS_HEAD;
    if (c->scorer == GLC_SCORER_WEIGHTED_DOT) {
        switch (j) {
            case 0: SPEC2("scorer.proj_text.weight", 2 * H, H, lin_amp(1.0, (double)H));
            case 1: SPEC1("scorer.proj_text.bias", 2 * H, 0.05, 0.0);
            case 2: SPEC2("scorer.proj_label.weight", 2 * H, H, lin_amp(1.0, (double)H));
            case 3: SPEC1("scorer.proj_label.bias", 2 * H, 0.05, 0.0);
            case 4: SPEC2("scorer.out_mlp.0.weight", 4 * H, 3 * H, lin_amp(1.0, (double)(3 * H)));
            case 5: SPEC1("scorer.out_mlp.0.bias", 4 * H, 0.05, 0.0);
            case 6: SPEC2("scorer.out_mlp.3.weight", 1, 4 * H, lin_amp(2.0, (double)(4 * H)));
            default: SPEC1("scorer.out_mlp.3.bias", 1, 0.1, 0.0);
        }
    }
    const uint64_t Mh = GLC_SCORER_MLP_HIDDEN;
    switch (j) {
        case 0: SPEC2("scorer.mlp.0.weight", Mh, 2 * H, lin_amp(1.0, (double)(2 * H)));
        case 1: SPEC1("scorer.mlp.0.bias", Mh, 0.05, 0.0);
        case 2: SPEC2("scorer.mlp.2.weight", Mh / 2, Mh, lin_amp(1.5, (double)Mh));
        case 3: SPEC1("scorer.mlp.2.bias", Mh / 2, 0.05, 0.0);
        case 4: SPEC2("scorer.mlp.4.weight", 1, Mh / 2, lin_amp(3.0, (double)(Mh / 2)));
        default: SPEC1("scorer.mlp.4.bias", 1, 0.1, 0.0);
    }
#undef SPEC1
#undef SPEC2
}

int glc_tensor_spec(const glc_model_config* c, int i, char* name, uint64_t shape[4], double* amp, double* mean) {
    const uint64_t H = (uint64_t)c->hidden, I = (uint64_t)c->inter;
    const uint64_t P = 2ull * (uint64_t)(c->pos_buckets > 0 ? c->pos_buckets : c->max_rel_pos);
    shape[0] = shape[1] = shape[2] = shape[3] = 0;
    *mean = 0.0;
#define SPEC1(nm, n0, a, m) do { snprintf(name, 96, "%s", nm); shape[0] = (n0); *amp = (a); *mean = (m); return 1; } while (0)
#define SPEC2(nm, n0, n1, a) do { snprintf(name, 96, "%s", nm); shape[0] = (n0); shape[1] = (n1); *amp = (a); return 2; } while (0)
    if (c->backbone == GLC_BACKBONE_DECODER) {
        const uint64_t nqd = (uint64_t)c->heads * (uint64_t)c->head_dim;
        const uint64_t nkvd = (uint64_t)(c->kv_heads > 0 ? c->kv_heads : c->heads) * (uint64_t)c->head_dim;
        const int nl = GLC_DEC_TENSORS_PER_LAYER * c->layers;
        char buf[96];
        if (i == 0) SPEC2("embed_tokens.weight", (uint64_t)c->vocab, H, 1.0);
        if (i >= 1 && i < 1 + nl) {
            const int l = (i - 1) / GLC_DEC_TENSORS_PER_LAYER, k = (i - 1) % GLC_DEC_TENSORS_PER_LAYER;
            static const char* sfx[12] = {
                "input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight",
                "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias", "self_attn.o_proj.weight",
                "post_attention_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"};
            snprintf(buf, sizeof buf, "layers.%d.%s", l, sfx[k]);
            switch (k) {
                case 0: case 8: SPEC1(buf, H, 0.2, 1.0);
                case 1: SPEC2(buf, nqd, H, lin_amp(1.6, (double)H));
                case 2: SPEC1(buf, nqd, 0.1, 0.0);
                case 3: SPEC2(buf, nkvd, H, lin_amp(1.6, (double)H));
                case 4: case 6: SPEC1(buf, nkvd, 0.1, 0.0);
                case 5: SPEC2(buf, nkvd, H, lin_amp(1.0, (double)H));
                case 7: SPEC2(buf, H, nqd, lin_amp(0.7, (double)nqd));
                case 9: case 10: SPEC2(buf, I, H, lin_amp(1.0, (double)H));
                default: SPEC2(buf, H, I, lin_amp(0.7, (double)I));
            }
        }
        if (i == 1 + nl) SPEC1("norm.weight", H, 0.2, 1.0);
        return head_spec(c, i - 2 - nl, name, shape, amp, mean);
    }
    switch (i) {
        case 0: SPEC2("embeddings.word_embeddings.weight", (uint64_t)c->vocab, H, 1.0);
        case 1: SPEC1("embeddings.LayerNorm.weight", H, 0.2, 1.0);
        case 2: SPEC1("embeddings.LayerNorm.bias", H, 0.1, 0.0);
        case 3: SPEC2("encoder.rel_embeddings.weight", P, H, 1.0);
        case 4: SPEC1("encoder.LayerNorm.weight", H, 0.2, 1.0);
        case 5: SPEC1("encoder.LayerNorm.bias", H, 0.1, 0.0);
        default: break;
    }
    const int nl = GLC_TENSORS_PER_LAYER * c->layers;
    char buf[96];
    if (i < GLC_TENSORS_FIXED + nl) {
        const int l = (i - GLC_TENSORS_FIXED) / GLC_TENSORS_PER_LAYER, k = (i - GLC_TENSORS_FIXED) % GLC_TENSORS_PER_LAYER;
        static const char* sfx[16] = {
            "attention.self.query_proj.weight", "attention.self.query_proj.bias", "attention.self.key_proj.weight",
            "attention.self.key_proj.bias", "attention.self.value_proj.weight", "attention.self.value_proj.bias",
            "attention.output.dense.weight", "attention.output.dense.bias", "attention.output.LayerNorm.weight",
            "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
            "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"};
        snprintf(buf, sizeof buf, "encoder.layer.%d.%s", l, sfx[k]);
        switch (k) {
            case 0: case 2: SPEC2(buf, H, H, lin_amp(2.0, (double)H));
            case 4: case 6: SPEC2(buf, H, H, lin_amp(1.0, (double)H));
            case 10: SPEC2(buf, I, H, lin_amp(1.0, (double)H));
            case 12: SPEC2(buf, H, I, lin_amp(1.0, (double)I));
            case 11: SPEC1(buf, I, 0.1, 0.0);
            case 8: case 14: SPEC1(buf, H, 0.2, 1.0);
            default: SPEC1(buf, H, 0.1, 0.0);
        }
    }
    return head_spec(c, i - GLC_TENSORS_FIXED - nl, name, shape, amp, mean);
#undef SPEC1
#undef SPEC2
}

static int load_synthetic(const char* spec, glc_weights* w) {
    char name[64];
    unsigned long long seed = 42;
    const char* p = spec + strlen("synthetic:");
    const char* colon = strchr(p, ':');
    size_t nlen = colon ? (size_t)(colon - p) : strlen(p);
    if (nlen == 0 || nlen >= sizeof name) { fprintf(stderr, "Error: bad synthetic model spec '%s'\n", spec); return -1; }
    memcpy(name, p, nlen);
    name[nlen] = 0;
    const char* sc = NULL;                       /* "synthetic:<config>[:seed[:scorer]]", scorer = simple | weighted-dot | mlp */
    if (colon) { char* end = NULL; seed = strtoull(colon + 1, &end, 10); if (end && *end == ':') sc = end + 1; }
    if (glc_named_config(name, &w->cfg) != 0) { fprintf(stderr, "Error: unknown synthetic config '%s'\n", name); return -1; }
    if (sc) {
        if (!strcmp(sc, "simple")) w->cfg.scorer = GLC_SCORER_DOT;
        else if (!strcmp(sc, "weighted-dot")) w->cfg.scorer = GLC_SCORER_WEIGHTED_DOT;
        else if (!strcmp(sc, "mlp")) w->cfg.scorer = GLC_SCORER_MLP;
        else { fprintf(stderr, "Error: unknown scorer '%s' in '%s' (simple, weighted-dot, mlp)\n", sc, spec); return -1; }
    }
    w->n_tensors = glc_num_tensors_cfg(&w->cfg);
    w->tensors = (const float**)calloc((size_t)w->n_tensors, sizeof(float*));
    if (!w->tensors) return -1;
    size_t total = 0;
    char tn[96];
    uint64_t shp[4];
    double amp, mean;
    for (int i = 0; i < w->n_tensors; ++i) {
        int nd = glc_tensor_spec(&w->cfg, i, tn, shp, &amp, &mean);
        if (nd < 0) return -1;
        size_t n = (size_t)shp[0] * (nd > 1 ? (size_t)shp[1] : 1);
        total += (n + 15) / 16 * 16;
    }
    w->_owned = (float*)malloc(total * sizeof(float));
    if (!w->_owned) { fprintf(stderr, "Error: cannot allocate %zu bytes for synthetic weights\n", total * sizeof(float)); return -1; }
    size_t off = 0;
    for (int i = 0; i < w->n_tensors; ++i) {
        int nd = glc_tensor_spec(&w->cfg, i, tn, shp, &amp, &mean);
        size_t n = (size_t)shp[0] * (nd > 1 ? (size_t)shp[1] : 1);
        glc_prng_fill(seed, tn, n, amp, mean, w->_owned + off);
        w->tensors[i] = w->_owned + off;
        off += (n + 15) / 16 * 16;
    }
    return 0;
}

static int load_blob(const char* path, glc_weights* w) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { fprintf(stderr, "Error: cannot open model file '%s'\n", path); return -1; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < GLCW_HEADER_BYTES) { close(fd); fprintf(stderr, "Error: '%s' is not a GLCW blob\n", path); return -1; }
    void* m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { fprintf(stderr, "Error: mmap of '%s' failed\n", path); return -1; }
    w->_map = m;
    w->_map_len = (size_t)st.st_size;
    const unsigned char* b = (const unsigned char*)m;
    static const unsigned char magic[8] = {'G', 'L', 'C', 'W', 0, 1, 0, 0};
    if (memcmp(b, magic, 8) != 0) { fprintf(stderr, "Error: '%s' has no GLCW magic\n", path); return -1; }
    uint32_t ver, nt;
    memcpy(&ver, b + 8, 4);
    memcpy(&nt, b + 12, 4);
    int32_t ints[20];
    float fl[3];
    memcpy(ints, b + 16, sizeof ints);
    memcpy(fl, b + 16 + sizeof ints, sizeof fl);
    glc_model_config* c = &w->cfg;
    c->vocab = ints[0]; c->hidden = ints[1]; c->layers = ints[2]; c->heads = ints[3]; c->head_dim = ints[4]; c->inter = ints[5];
    c->pos_buckets = ints[6]; c->max_rel_pos = ints[7]; c->pad_id = ints[8]; c->cls_id = ints[9]; c->sep_id = ints[10];
    c->class_token_index = ints[11]; c->text_token_index = ints[12]; c->pooling = ints[13]; c->scorer = ints[14];
    c->embed_class_token = ints[15]; c->normalize_features = ints[16];
    c->backbone = ints[17]; c->kv_heads = ints[18]; c->causal = ints[19];
    c->ln_eps = fl[0]; c->logit_scale = fl[1]; c->rope_theta = fl[2];
    if (ver != 2 || c->layers <= 0 || c->layers > 4096 || (int)nt != glc_num_tensors_cfg(c)) {
        fprintf(stderr, "Error: '%s': unsupported GLCW header (version %u, %u tensors)\n", path, ver, nt);
        return -1;
    }
    w->n_tensors = (int)nt;
    w->tensors = (const float**)calloc(nt, sizeof(float*));
    if (!w->tensors) return -1;
    char want[96];
    uint64_t shp[4];
    double amp, mean;
    for (uint32_t i = 0; i < nt; ++i) {
        const unsigned char* r = b + GLCW_HEADER_BYTES + (size_t)i * GLCW_REC_BYTES;
        if ((size_t)(r - b) + GLCW_REC_BYTES > w->_map_len) { fprintf(stderr, "Error: '%s' truncated\n", path); return -1; }
        uint32_t dt, nd;
        uint64_t shape[4], off, nb;
        memcpy(&dt, r + 96, 4); memcpy(&nd, r + 100, 4); memcpy(shape, r + 104, 32); memcpy(&off, r + 136, 8); memcpy(&nb, r + 144, 8);
        int wnd = glc_tensor_spec(c, (int)i, want, shp, &amp, &mean);
        if (wnd < 0 || strncmp((const char*)r, want, 96) != 0 || dt != 0 || (int)nd != wnd || shape[0] != shp[0] ||
            (wnd > 1 && shape[1] != shp[1]) || off % 4 || off + nb > w->_map_len || nb != 4 * shp[0] * (wnd > 1 ? shp[1] : 1)) {
            fprintf(stderr, "Error: '%s': tensor %u ('%.95s') does not match the expected '%s'\n", path, i, (const char*)r, want);
            return -1;
        }
        w->tensors[i] = (const float*)(b + off);
    }
    return 0;
}

int glc_weights_load(const char* path, glc_weights* out) {
    if (!path || !out) return -1;
    memset(out, 0, sizeof(*out));
    int rc;
    struct stat sb;
    const size_t pl = strlen(path);
    if (strncmp(path, "synthetic:", 10) == 0) rc = load_synthetic(path, out);
    else if ((stat(path, &sb) == 0 && S_ISDIR(sb.st_mode)) || (pl > 12 && !strcmp(path + pl - 12, ".safetensors")))
        rc = glc_load_hf_checkpoint(path, out);                 /* HF directory (config.json + model.safetensors) */
    else rc = load_blob(path, out);
    if (rc != 0) glc_weights_free(out);
    return rc;
}

void glc_weights_free(glc_weights* w) {
    if (!w) return;
    free((void*)w->tensors);
    free(w->_owned);
    if (w->_map) munmap(w->_map, w->_map_len);
    memset(w, 0, sizeof(*w));
}
