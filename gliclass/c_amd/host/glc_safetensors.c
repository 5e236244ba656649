/*
 * Native checkpoint importer: an HF model directory (config.json + model.safetensors) or a
 * .safetensors file with config.json beside it -> glc_weights, so create_ort_session() can open what the reference's
 * launcher downloads (/root/reference/run_GLiClass.sh:34-36) without the ONNX export step
 * (/root/reference/ONNX_CONVERTING/convert_to_onnx.py:48-78).  Pure C: the host layer's JSON reader parses both
 * config.json and the safetensors header (8-byte little-endian length, JSON {"name": {"dtype","shape","data_offsets"}},
 * raw little-endian tensor data); F32 / F16 / BF16 tensors are widened to fp32.
 *
 * Tensor names are HF's (`DebertaV2Model` / `Qwen2Model` state_dict) under any of the prefixes GLiClass checkpoints use;
 * configuration fields follow transformers' DebertaV2Config / Qwen2Config inside `encoder_config`, and the GLiClass
 * fields as restated in SURVEY.md §8a row a12 (class_token_index, text_token_index, pooling_strategy, scorer_type,
 * embed_class_token, normalize_features ...).  The GLiClass field names come from the upstream python package, which
 * is not available offline: that part is UNPINNED, so anything the engine does not implement is rejected loudly instead
 * of being guessed (other scorers, LSTM, bi-encoder architectures, conv layer, absolute positions ...).
 */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "glc_json.h"
#include "glc_weights.h"

static const char* const kPrefixes[] = {"", "model.", "deberta.", "encoder_model.model.", "model.encoder_model.model.",
                                        "encoder_model.", "model.encoder_model.", "decoder_model.model.", "model.decoder_model.model.",
                                        "encoder_model.deberta.", "model.encoder_model.deberta."};
#define N_PREFIXES (sizeof(kPrefixes) / sizeof(kPrefixes[0]))

static double jnum(const gj_value* o, const char* k, double dflt) {
    const gj_value* v = gj_get(o, k);
    return gj_is(v, GJ_NUM) ? v->u.num : dflt;
}
static int jflag(const gj_value* o, const char* k, int dflt) {
    const gj_value* v = gj_get(o, k);
    if (gj_is(v, GJ_BOOL)) return v->u.boolean;
    if (gj_is(v, GJ_NUM)) return v->u.num != 0;
    return dflt;
}
static const char* jtext(const gj_value* o, const char* k) {
    const gj_value* v = gj_get(o, k);
    return gj_is(v, GJ_STR) ? v->u.str.s : NULL;
}
/* "p2c|c2p" or ["p2c","c2p"] */
static int list_has(const gj_value* v, const char* item) {
    if (gj_is(v, GJ_STR)) return strstr(v->u.str.s, item) != NULL;
    if (gj_is(v, GJ_ARR)) for (size_t i = 0; i < v->u.arr.n; ++i) if (gj_is(v->u.arr.items[i], GJ_STR) && !strcmp(v->u.arr.items[i]->u.str.s, item)) return 1;
    return 0;
}

static char* slurp(const char* path, size_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* b = (char*)malloc((size_t)(n > 0 ? n : 0) + 1);
    if (!b) { fclose(f); return NULL; }
    size_t got = fread(b, 1, (size_t)(n > 0 ? n : 0), f);
    fclose(f);
    b[got] = 0; *len = got;
    return b;
}

#define REJECT(...) do { fprintf(stderr, "Error: checkpoint config: " __VA_ARGS__); fputc('\n', stderr); return -1; } while (0)

static int parse_config(const gj_value* root, glc_model_config* c) {
    memset(c, 0, sizeof(*c));
    const gj_value* enc = gj_get(root, "encoder_config");
    if (!gj_is(enc, GJ_OBJ)) enc = root;                                  /* a bare backbone config */
    const char* mt = jtext(enc, "model_type");
    if (!mt) REJECT("no model_type");
    const char* arch = jtext(root, "architecture_type");
    if (arch && strcmp(arch, "uni-encoder")) REJECT("architecture_type '%s' is not implemented (only uni-encoder)", arch);
    if (jflag(root, "use_lstm", 0)) REJECT("use_lstm=true is not implemented");
    const char* scorer = jtext(root, "scorer_type");
    int scorer_id = GLC_SCORER_DOT;
    if (scorer) {
        if (!strcmp(scorer, "simple")) scorer_id = GLC_SCORER_DOT;
        else if (!strcmp(scorer, "weighted-dot")) scorer_id = GLC_SCORER_WEIGHTED_DOT;
        else if (!strcmp(scorer, "mlp")) scorer_id = GLC_SCORER_MLP;
        else REJECT("scorer_type '%s' is not implemented (simple, weighted-dot, mlp)", scorer);
    }
    const char* pool = jtext(root, "pooling_strategy");
    c->pooling = GLC_POOL_FIRST;
    if (pool) {
        if (!strcmp(pool, "first")) c->pooling = GLC_POOL_FIRST;
        else if (!strcmp(pool, "avg")) c->pooling = GLC_POOL_AVG;
        else if (!strcmp(pool, "last")) c->pooling = GLC_POOL_LAST;
        else REJECT("pooling_strategy '%s' is not implemented (first, avg, last)", pool);
    }
    c->scorer = scorer_id;
    c->embed_class_token = jflag(root, "embed_class_token", 1);
    c->normalize_features = jflag(root, "normalize_features", 0);
    c->logit_scale = (float)jnum(root, "logit_scale", 1.0);
    c->class_token_index = (int32_t)jnum(root, "class_token_index", -1);
    c->text_token_index = (int32_t)jnum(root, "text_token_index", -1);
    c->pad_id = (int32_t)jnum(enc, "pad_token_id", 0);
    c->cls_id = (int32_t)jnum(enc, "cls_token_id", jnum(enc, "bos_token_id", 1));
    c->sep_id = (int32_t)jnum(enc, "sep_token_id", jnum(enc, "eos_token_id", 2));
    c->hidden = (int32_t)jnum(enc, "hidden_size", 0);
    c->layers = (int32_t)jnum(enc, "num_hidden_layers", 0);
    c->heads = (int32_t)jnum(enc, "num_attention_heads", 0);
    c->inter = (int32_t)jnum(enc, "intermediate_size", 0);
    c->vocab = (int32_t)jnum(root, "vocab_size", jnum(enc, "vocab_size", 0));
    if (c->hidden <= 0 || c->layers <= 0 || c->heads <= 0 || c->inter <= 0 || c->hidden % c->heads) REJECT("missing or inconsistent backbone dimensions");
    c->head_dim = c->hidden / c->heads;
    if (!strcmp(mt, "deberta-v2")) {
        c->backbone = GLC_BACKBONE_DEBERTA;
        c->kv_heads = c->heads; c->causal = 1; c->rope_theta = 1.0e6f;   /* unused by this backbone; the blob header's defaults */
        c->ln_eps = (float)jnum(enc, "layer_norm_eps", 1e-7);
        if (!jflag(enc, "relative_attention", 0)) REJECT("relative_attention=false is not implemented");
        const gj_value* pat = gj_get(enc, "pos_att_type");
        if (!list_has(pat, "c2p") || !list_has(pat, "p2c")) REJECT("pos_att_type must contain c2p and p2c");
        if (!jflag(enc, "share_att_key", 0)) REJECT("share_att_key=false is not implemented");
        const char* nre = jtext(enc, "norm_rel_ebd");
        if (!nre || !strstr(nre, "layer_norm")) REJECT("norm_rel_ebd must be layer_norm");
        if (jflag(enc, "position_biased_input", 1)) REJECT("position_biased_input=true is not implemented");
        if (jnum(enc, "type_vocab_size", 0) != 0) REJECT("type_vocab_size != 0 is not implemented");
        if (jnum(enc, "conv_kernel_size", 0) > 0) REJECT("conv_kernel_size > 0 (ConvLayer) is not implemented");
        int mrp = (int)jnum(enc, "max_relative_positions", -1);
        if (mrp < 1) mrp = (int)jnum(enc, "max_position_embeddings", 512);        /* modeling_deberta_v2.py:586-588 */
        c->max_rel_pos = mrp;
        c->pos_buckets = (int32_t)jnum(enc, "position_buckets", -1);
        if (c->pos_buckets < 1) c->pos_buckets = 0;
    } else if (!strcmp(mt, "qwen2")) {
        c->backbone = GLC_BACKBONE_DECODER;
        c->kv_heads = (int32_t)jnum(enc, "num_key_value_heads", c->heads);
        c->causal = jflag(root, "causal", 1);                                      /* BASELINE.json: causal; upstream wrapping unpinned */
        c->ln_eps = (float)jnum(enc, "rms_norm_eps", 1e-6);
        c->rope_theta = (float)jnum(enc, "rope_theta", 1.0e6);
        c->pos_buckets = 0; c->max_rel_pos = 0;
        if (!pool) c->pooling = GLC_POOL_LAST;
        if (c->kv_heads <= 0 || c->heads % c->kv_heads) REJECT("num_key_value_heads does not divide num_attention_heads");
    } else REJECT("backbone model_type '%s' is not implemented (deberta-v2, qwen2)", mt);
    return 0;
}

typedef struct { const gj_value* hdr; const unsigned char* data; size_t data_len; } st_file;

static const gj_value* st_find(const st_file* st, const char* name, char* found, size_t fcap) {
    for (size_t p = 0; p < N_PREFIXES; ++p) {
        snprintf(found, fcap, "%s%s", kPrefixes[p], name);
        const gj_value* v = gj_get(st->hdr, found);
        if (gj_is(v, GJ_OBJ)) return v;
    }
    return NULL;
}

static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (!man) bits = sign;
        else { int e = -1; do { man <<= 1; ++e; } while (!(man & 0x400u)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13); }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

int glc_load_hf_checkpoint(const char* path, glc_weights* w) {
    char dir[3072], stp[4096], cfp[4096], err[200];
    struct stat sb;
    if (stat(path, &sb) != 0) { fprintf(stderr, "Error: cannot open model '%s'\n", path); return -1; }
    if (S_ISDIR(sb.st_mode)) snprintf(dir, sizeof dir, "%s", path);
    else {
        snprintf(dir, sizeof dir, "%s", path);
        char* slash = strrchr(dir, '/');
        if (slash) *slash = 0; else snprintf(dir, sizeof dir, ".");
    }
    if (S_ISDIR(sb.st_mode)) snprintf(stp, sizeof stp, "%s/model.safetensors", dir); else snprintf(stp, sizeof stp, "%s", path);
    snprintf(cfp, sizeof cfp, "%s/config.json", dir);

    size_t clen = 0;
    char* ctext = slurp(cfp, &clen);
    if (!ctext) { fprintf(stderr, "Error: cannot read '%s'\n", cfp); return -1; }
    gj_doc* cdoc = gj_parse(ctext, clen, 0, err, sizeof err);
    free(ctext);
    if (!cdoc) { fprintf(stderr, "Error: %s: %s\n", cfp, err); return -1; }
    int rc = parse_config(gj_root(cdoc), &w->cfg);
    gj_free(cdoc);
    if (rc) return -1;

    int fd = open(stp, O_RDONLY);
    if (fd < 0) {
        char idx[4200]; snprintf(idx, sizeof idx, "%s/model.safetensors.index.json", dir);
        if (access(idx, R_OK) == 0) fprintf(stderr, "Error: '%s' is a sharded checkpoint; merge it into one model.safetensors first\n", dir);
        else fprintf(stderr, "Error: cannot open '%s'\n", stp);
        return -1;
    }
    if (fstat(fd, &sb) != 0 || sb.st_size < 8) { close(fd); fprintf(stderr, "Error: '%s' is not a safetensors file\n", stp); return -1; }
    void* m = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { fprintf(stderr, "Error: mmap of '%s' failed\n", stp); return -1; }
    const unsigned char* b = (const unsigned char*)m;
    uint64_t hlen = 0;
    for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | b[i];
    gj_doc* hdoc = NULL;
    rc = -1;
    if (hlen > (uint64_t)sb.st_size - 8) { fprintf(stderr, "Error: '%s': header length beyond the file\n", stp); goto done; }
    hdoc = gj_parse((const char*)b + 8, (size_t)hlen, 0, err, sizeof err);
    if (!hdoc || !gj_is(gj_root(hdoc), GJ_OBJ)) { fprintf(stderr, "Error: '%s': %s\n", stp, hdoc ? "header is not an object" : err); goto done; }
    st_file st = {gj_root(hdoc), b + 8 + hlen, (size_t)sb.st_size - 8 - (size_t)hlen};

    glc_model_config* c = &w->cfg;
    char found[200];
    /* the embedding matrix decides the vocabulary size (tokens were added after the backbone config was written) */
    const gj_value* emb = st_find(&st, c->backbone == GLC_BACKBONE_DECODER ? "embed_tokens.weight" : "embeddings.word_embeddings.weight", found, sizeof found);
    const gj_value* eshape = gj_get(emb, "shape");
    if (!emb || !gj_is(eshape, GJ_ARR) || eshape->u.arr.n != 2) { fprintf(stderr, "Error: '%s': no word-embedding tensor under any known prefix\n", stp); goto done; }
    c->vocab = (int32_t)eshape->u.arr.items[0]->u.num;
    if (c->class_token_index < 0) c->class_token_index = c->vocab - 2;
    if (c->text_token_index < 0) c->text_token_index = c->vocab - 1;

    w->n_tensors = glc_num_tensors_cfg(c);
    w->tensors = (const float**)calloc((size_t)w->n_tensors, sizeof(float*));
    if (!w->tensors) goto done;
    char tn[96]; uint64_t shp[4]; double amp, mean; size_t total = 0;
    for (int i = 0; i < w->n_tensors; ++i) {
        int nd = glc_tensor_spec(c, i, tn, shp, &amp, &mean);
        if (nd < 0) goto done;
        total += ((size_t)shp[0] * (nd > 1 ? (size_t)shp[1] : 1) + 15) / 16 * 16;
    }
    w->_owned = (float*)malloc(total * sizeof(float));
    if (!w->_owned) { fprintf(stderr, "Error: cannot allocate %zu bytes for the checkpoint\n", total * sizeof(float)); goto done; }
    size_t off = 0;
    for (int i = 0; i < w->n_tensors; ++i) {
        int nd = glc_tensor_spec(c, i, tn, shp, &amp, &mean);
        size_t n = (size_t)shp[0] * (nd > 1 ? (size_t)shp[1] : 1);
        const gj_value* t = st_find(&st, tn, found, sizeof found);
        if (!t) { fprintf(stderr, "Error: '%s': tensor '%s' not found under any known prefix\n", stp, tn); goto done; }
        const char* dt = jtext(t, "dtype");
        const gj_value* shape = gj_get(t, "shape");
        const gj_value* offs = gj_get(t, "data_offsets");
        if (!dt || !gj_is(shape, GJ_ARR) || !gj_is(offs, GJ_ARR) || offs->u.arr.n != 2) { fprintf(stderr, "Error: '%s': malformed entry '%s'\n", stp, found); goto done; }
        int shape_ok = (int)shape->u.arr.n == nd;
        for (int d = 0; shape_ok && d < nd; ++d) shape_ok = (uint64_t)shape->u.arr.items[d]->u.num == shp[d];
        if (!shape_ok) { fprintf(stderr, "Error: '%s': tensor '%s' has an unexpected shape (want %llu x %llu)\n", stp, found, (unsigned long long)shp[0], (unsigned long long)(nd > 1 ? shp[1] : 1)); goto done; }
        size_t b0 = (size_t)offs->u.arr.items[0]->u.num, b1 = (size_t)offs->u.arr.items[1]->u.num;
        size_t esz = !strcmp(dt, "F32") ? 4 : (!strcmp(dt, "F16") || !strcmp(dt, "BF16")) ? 2 : 0;
        if (!esz) { fprintf(stderr, "Error: '%s': tensor '%s' has dtype %s (F32, F16, BF16 are read)\n", stp, found, dt); goto done; }
        if (b1 < b0 || b1 > st.data_len || b1 - b0 != n * esz) { fprintf(stderr, "Error: '%s': tensor '%s' has bad data offsets\n", stp, found); goto done; }
        float* dst = w->_owned + off;
        const unsigned char* src = st.data + b0;
        if (esz == 4) memcpy(dst, src, n * 4);
        else if (dt[0] == 'B') for (size_t k = 0; k < n; ++k) { uint32_t u = ((uint32_t)src[2 * k] | ((uint32_t)src[2 * k + 1] << 8)) << 16; memcpy(&dst[k], &u, 4); }
        else for (size_t k = 0; k < n; ++k) dst[k] = half_to_float((uint16_t)(src[2 * k] | (src[2 * k + 1] << 8)));
        w->tensors[i] = dst;
        off += (n + 15) / 16 * 16;
    }
    rc = 0;
done:
    if (hdoc) gj_free(hdoc);
    munmap(m, (size_t)sb.st_size);
    return rc;
}
