/*
 * Native tokenizer of the host layer.  Two interfaces in one file:
 *
 *  (1) the reference's own src/tokenizer.c functions, same names / semantics / messages
 *      (/root/reference/src/tokenizer.c:19-91 tokenize_inputs, :98-120 print_tokenized_inputs, :127-136
 *      free_tokenized_inputs, :145-184 create_tokenizer), and
 *  (2) the tokenizers-cpp C API they call (include/tokenizers_c.h), implemented here in plain C instead of the Rust
 *      `tokenizers` library behind mlc-ai/tokenizers-cpp.
 *
 * What is restated is the published algorithm of HF `tokenizers` (0.22; the reference pins no version) for the pieces a
 * DeBERTa-v3 `tokenizer.json` uses:
 *   - AddedVocabulary::extract_and_normalize: tokens with normalized=false are cut out of the raw text, every remaining
 *     piece is normalised on its own, then tokens with normalized=true are cut out of the normalised pieces
 *     (leftmost-longest matching; lstrip / rstrip / single_word honoured);
 *   - normalizers Sequence / Strip / Precompiled (SentencePiece's precompiled_charsmap: darts-clone double array +
 *     replacement blob, applied per extended grapheme cluster when the cluster is shorter than 6 bytes, else per
 *     character, taking the FIRST = shortest prefix match, exactly like the spm_precompiled crate) / Replace (literal
 *     patterns and the "X{n,}" / "X+" run regexes) / Lowercase / Prepend;
 *   - pre-tokenizer Metaspace (' ' -> replacement, prepend_scheme, split before every replacement character);
 *   - model Unigram: Viterbi over a byte trie of the vocabulary (`encode_optimized`): strict '>' on path scores, unknown
 *     characters cost min_score - 10 and consecutive unknowns fuse into one unk token; optional byte_fallback;
 *   - post-processor TemplateProcessing (single sequence), Bert/RobertaProcessing; optional `truncation`.
 * Parity is pinned by tests/golden/tokenizer_golden.json.gz (ids produced by the python `tokenizers` wheel -- the same
 * Rust code -- on a DeBERTa-v3-structured tokenizer.json, see oracle/gen_tokenizer_fixture.py) and, where the wheel is
 * importable, by a live randomised comparison (tests/test_tokenizer.py).
 * Known approximations (none of them reachable with the DeBERTa-v3 file): prepend_scheme "first" is decided from "first piece
 * of the text AND the normalisers removed nothing at its start" (equal to the Rust library's original-offset test on every live
 * probe, tests/test_tokenizer.py); the `padding` section is ignored (the reference pads itself, src/tokenizer.c:77-81).
 */
#include "tokenizer.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "glc_cpus.h"
#include "glc_json.h"
#include "glc_unicode_tables.h"
#include "glc_bpe_tables.h"
#include "tokenizers_c.h"

/* ---------------------------------------------------------------- small utilities */

typedef struct { char* s; size_t len, cap; } sbuf;

static int sb_reserve(sbuf* b, size_t extra) {
    if (b->len + extra + 1 <= b->cap) return 1;
    size_t nc = b->cap ? b->cap * 2 : 256;
    while (nc < b->len + extra + 1) nc *= 2;
    char* ns = (char*)realloc(b->s, nc);
    if (!ns) return 0;
    b->s = ns; b->cap = nc;
    return 1;
}
static void sb_put(sbuf* b, const char* s, size_t n) {
    if (!sb_reserve(b, n)) return;
    memcpy(b->s + b->len, s, n); b->len += n; b->s[b->len] = 0;
}
static void sb_clear(sbuf* b) { b->len = 0; if (b->s) b->s[0] = 0; }
static void sb_free(sbuf* b) { free(b->s); b->s = NULL; b->len = b->cap = 0; }

typedef struct { int32_t* v; size_t n, cap; } ivec;
static void iv_push(ivec* a, int32_t x) {
    if (a->n == a->cap) {
        size_t nc = a->cap ? a->cap * 2 : 64;
        int32_t* nv = (int32_t*)realloc(a->v, nc * sizeof(int32_t));
        if (!nv) return;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = x;
}

/* Decodes one UTF-8 scalar (input has been sanitised, so sequences are well formed). */
static inline uint32_t u8_decode(const unsigned char* s, size_t n, size_t* adv) {
    unsigned char c = s[0];
    if (c < 0x80 || n < 2) { *adv = 1; return c; }
    if (c < 0xE0) { *adv = 2; return ((uint32_t)(c & 0x1F) << 6) | (s[1] & 0x3F); }
    if (c < 0xF0 || n < 4) { if (n < 3) { *adv = 1; return 0xFFFD; } *adv = 3; return ((uint32_t)(c & 0x0F) << 12) | ((uint32_t)(s[1] & 0x3F) << 6) | (s[2] & 0x3F); }
    *adv = 4;
    return ((uint32_t)(c & 0x07) << 18) | ((uint32_t)(s[1] & 0x3F) << 12) | ((uint32_t)(s[2] & 0x3F) << 6) | (s[3] & 0x3F);
}
static inline size_t u8_len(unsigned char c) { return c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4; }
static size_t u8_encode(char* o, uint32_t cp) {
    if (cp < 0x80) { o[0] = (char)cp; return 1; }
    if (cp < 0x800) { o[0] = (char)(0xC0 | (cp >> 6)); o[1] = (char)(0x80 | (cp & 0x3F)); return 2; }
    if (cp < 0x10000) { o[0] = (char)(0xE0 | (cp >> 12)); o[1] = (char)(0x80 | ((cp >> 6) & 0x3F)); o[2] = (char)(0x80 | (cp & 0x3F)); return 3; }
    o[0] = (char)(0xF0 | (cp >> 18)); o[1] = (char)(0x80 | ((cp >> 12) & 0x3F)); o[2] = (char)(0x80 | ((cp >> 6) & 0x3F)); o[3] = (char)(0x80 | (cp & 0x3F));
    return 4;
}

/* Copies `s` replacing every ill-formed UTF-8 sequence by U+FFFD (Rust's from_utf8_lossy policy: maximal subparts). */
static void u8_sanitize(const char* s, size_t n, sbuf* out) {
    const unsigned char* p = (const unsigned char*)s;
    size_t i = 0, run = 0;
    while (i < n) {
        unsigned char c = p[i];
        size_t need = 0; uint32_t lo = 0x80, hi = 0xBF;
        if (c < 0x80) { ++i; continue; }
        else if (c >= 0xC2 && c <= 0xDF) need = 1;
        else if (c == 0xE0) { need = 2; lo = 0xA0; }
        else if (c >= 0xE1 && c <= 0xEC) need = 2;
        else if (c == 0xED) { need = 2; hi = 0x9F; }
        else if (c >= 0xEE && c <= 0xEF) need = 2;
        else if (c == 0xF0) { need = 3; lo = 0x90; }
        else if (c >= 0xF1 && c <= 0xF3) need = 3;
        else if (c == 0xF4) { need = 3; hi = 0x8F; }
        size_t k = 0; int ok = need > 0;
        while (ok && k < need) {
            if (i + 1 + k >= n) { ok = 0; break; }
            unsigned char d = p[i + 1 + k];
            if (k == 0 ? (d < lo || d > hi) : (d < 0x80 || d > 0xBF)) { ok = 0; break; }
            ++k;
        }
        if (ok) { i += 1 + need; continue; }
        sb_put(out, s + run, i - run);
        sb_put(out, "\xEF\xBF\xBD", 3);
        i += 1 + k;                                   /* the valid part of the broken sequence is consumed with it */
        run = i;
    }
    sb_put(out, s + run, n - run);
}

static int is_white_space(uint32_t c) {              /* Unicode White_Space = Rust char::is_whitespace = regex \s */
    return (c >= 9 && c <= 13) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
           c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

static int range_lookup(const glc_urange* t, size_t n, uint32_t c) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (c < t[mid].lo) hi = mid; else if (c > t[mid].hi) lo = mid + 1; else return t[mid].v;
    }
    return 0;
}
static int gcb_of(uint32_t c) {
    if (c < 0x300) {                                                                 /* fast path: Latin */
        if (c == 0x0D) return GCB_CR;
        if (c == 0x0A) return GCB_LF;
        if (c < 0x20 || (c >= 0x7F && c < 0xA0) || c == 0xAD) return GCB_CONTROL;
        return GCB_OTHER;
    }
    if (c >= 0xAC00 && c <= 0xD7A3) return ((c - 0xAC00) % 28 == 0) ? GCB_LV : GCB_LVT;
    return range_lookup(glc_gcb_ranges, sizeof(glc_gcb_ranges) / sizeof(glc_gcb_ranges[0]), c);
}
static int extpict_of(uint32_t c) { return c >= 0xA9 && range_lookup(glc_extpict_ranges, sizeof(glc_extpict_ranges) / sizeof(glc_extpict_ranges[0]), c); }
static int incb_of(uint32_t c) { return c < 0x300 ? INCB_NONE : range_lookup(glc_incb_ranges, sizeof(glc_incb_ranges) / sizeof(glc_incb_ranges[0]), c); }

/* End (byte offset) of the extended grapheme cluster that starts at byte i (UAX #29, rules GB3-GB13 + GB9c). */
static size_t grapheme_end(const unsigned char* s, size_t i, size_t n) {
    size_t adv;
    uint32_t c = u8_decode(s + i, n - i, &adv);
    size_t j = i + adv;
    int p = gcb_of(c);
    int emoji = extpict_of(c) ? 1 : 0;                 /* 1: ExtPict Extend*   2: ExtPict Extend* ZWJ */
    int conj = incb_of(c) == INCB_CONSONANT ? 1 : 0;   /* 1: consonant seen   2: ... and a linker after it */
    int ri = p == GCB_RI ? 1 : 0;
    while (j < n) {
        c = u8_decode(s + j, n - j, &adv);
        int q = gcb_of(c);
        int join;
        if (p == GCB_CR && q == GCB_LF) join = 1;
        else if (p == GCB_CONTROL || p == GCB_CR || p == GCB_LF || q == GCB_CONTROL || q == GCB_CR || q == GCB_LF) join = 0;
        else if (p == GCB_L && (q == GCB_L || q == GCB_V || q == GCB_LV || q == GCB_LVT)) join = 1;
        else if ((p == GCB_LV || p == GCB_V) && (q == GCB_V || q == GCB_T)) join = 1;
        else if ((p == GCB_LVT || p == GCB_T) && q == GCB_T) join = 1;
        else if (q == GCB_EXTEND || q == GCB_ZWJ || q == GCB_SPACINGMARK) join = 1;
        else if (p == GCB_PREPEND) join = 1;
        else if (conj == 2 && incb_of(c) == INCB_CONSONANT) join = 1;
        else if (emoji == 2 && extpict_of(c)) join = 1;
        else if (p == GCB_RI && q == GCB_RI && (ri & 1)) join = 1;
        else join = 0;
        if (!join) break;
        if (extpict_of(c)) emoji = 1;
        else if (emoji == 1 && q == GCB_EXTEND) emoji = 1;
        else if (emoji == 1 && q == GCB_ZWJ) emoji = 2;
        else emoji = 0;
        int ic = incb_of(c);
        if (ic == INCB_CONSONANT) conj = 1;
        else if (conj && ic == INCB_LINKER) conj = 2;
        else if (conj && ic == INCB_EXTEND) { /* keep */ }
        else conj = 0;
        ri = q == GCB_RI ? ri + 1 : 0;
        p = q;
        j += adv;
    }
    return j;
}

/* ---------------------------------------------------------------- tokenizer object */

enum { N_STRIP = 1, N_PRECOMPILED, N_REPLACE_LIT, N_REPLACE_RUN, N_LOWERCASE, N_PREPEND, N_NFC };
enum { MODEL_UNIGRAM = 0, MODEL_BPE = 1 };
enum { BLRE_NONE = 0, BLRE_GPT2 = 1, BLRE_QWEN = 2 };        /* which split regex runs before the byte-level mapping */
typedef struct { uint64_t key; int32_t rank, merged; } bpe_merge;   /* open-addressing table: (left id << 32 | right id) -> rank, merged id */

typedef struct {
    int kind;
    int left, right;                          /* Strip */
    uint32_t* trie; size_t ntrie;             /* Precompiled: double array */
    char* blob; size_t nblob;                 /* Precompiled: NUL-separated replacement strings */
    char* pat; size_t patlen;                 /* Replace literal pattern / Prepend text */
    char* rep; size_t replen;                 /* Replace content */
    int atom_ws; uint32_t atom_cp; int min_rep; /* Replace run regex: atom (\s or one character) repeated >= min_rep */
} norm_step;

typedef struct { uint32_t first_edge; uint16_t n_edges; int32_t id; } tnode;

typedef struct {
    int32_t id;
    char* content; size_t len;                /* as written in tokenizer.json */
    char* pattern; size_t plen;               /* what is searched: content, or normalize(content) when normalized=true */
    int single_word, lstrip, rstrip, normalized, special;
} added_tok;

typedef struct { int is_special; int32_t* ids; size_t n; } tmpl_item;

typedef struct glc_tokenizer {
    size_t nvocab;                            /* model vocabulary */
    char** tok; uint32_t* toklen; double* score;
    size_t id_space;                          /* 1 + largest id incl. added tokens */
    tnode* nodes; size_t nnodes; uint8_t* edge_byte; uint32_t* edge_node; size_t nedges;
    int32_t root_child[256];
    int32_t unk_id; int byte_fallback; double unk_score;
    int32_t byte_ids[256];                    /* <0xXX> pieces when byte_fallback */
    norm_step* steps; size_t nsteps;
    int has_metaspace; char ms_rep[8]; size_t ms_replen; int ms_scheme /* 0 always 1 first 2 never */, ms_split;
    added_tok* added; size_t nadded;
    tmpl_item* tmpl; size_t ntmpl; size_t tmpl_specials;
    size_t trunc_max; int trunc_left;
    int dec_metaspace;
    /* byte-level BPE (GPT-2 / Qwen2 family) */
    int model_kind;                           /* MODEL_UNIGRAM | MODEL_BPE */
    bpe_merge* merges; size_t merge_cap;      /* power of two, key 0 = empty (id pair (0,0) is stored with key | 1<<63) */
    int bpe_ignore_merges;
    int pre_bytelevel, bl_regex, bl_prefix_space, dec_bytelevel;
    uint16_t bl_cp[256];                      /* byte -> code point of its printable stand-in */
    int32_t bl_char_id[256];                  /* vocabulary id of that one-character token (-1: absent) */
    int16_t bl_byte_of[512];                  /* code point -> byte (-1: not a stand-in) */
    sbuf decoded;
} glc_tokenizer;

/* ---------------------------------------------------------------- vocabulary trie */

typedef struct { const char* s; uint32_t len; int32_t id; } sort_tok;

static int cmp_sort_tok(const void* a, const void* b) {
    const sort_tok* x = (const sort_tok*)a; const sort_tok* y = (const sort_tok*)b;
    size_t m = x->len < y->len ? x->len : y->len;
    int c = memcmp(x->s, y->s, m);
    if (c) return c;
    if (x->len != y->len) return x->len < y->len ? -1 : 1;
    return x->id < y->id ? -1 : x->id > y->id;
}

typedef struct { glc_tokenizer* tk; const sort_tok* st; size_t node_cap, edge_cap; int oom; } trie_builder;

static uint32_t tb_new_node(trie_builder* b) {
    glc_tokenizer* tk = b->tk;
    if (tk->nnodes == b->node_cap) {
        size_t nc = b->node_cap ? b->node_cap * 2 : 4096;
        tnode* nn = (tnode*)realloc(tk->nodes, nc * sizeof(tnode));
        if (!nn) { b->oom = 1; return 0; }
        tk->nodes = nn; b->node_cap = nc;
    }
    tnode* nd = &tk->nodes[tk->nnodes];
    nd->first_edge = 0; nd->n_edges = 0; nd->id = -1;
    return (uint32_t)tk->nnodes++;
}

static void tb_build(trie_builder* b, uint32_t node, size_t lo, size_t hi, uint32_t depth) {
    glc_tokenizer* tk = b->tk;
    const sort_tok* st = b->st;
    while (lo < hi && st[lo].len == depth) { tk->nodes[node].id = st[lo].id; ++lo; }   /* duplicates: the last id wins (HashMap insert) */
    if (lo >= hi || b->oom) return;
    size_t groups = 0;
    for (size_t i = lo; i < hi;) {
        unsigned char c = (unsigned char)st[i].s[depth];
        size_t j = i + 1;
        while (j < hi && (unsigned char)st[j].s[depth] == c) ++j;
        ++groups; i = j;
    }
    if (tk->nedges + groups > b->edge_cap) {
        size_t nc = b->edge_cap ? b->edge_cap * 2 : 8192;
        while (nc < tk->nedges + groups) nc *= 2;
        uint8_t* nb = (uint8_t*)realloc(tk->edge_byte, nc);
        uint32_t* nn = (uint32_t*)realloc(tk->edge_node, nc * sizeof(uint32_t));
        if (nb) tk->edge_byte = nb;
        if (nn) tk->edge_node = nn;
        if (!nb || !nn) { b->oom = 1; return; }
        b->edge_cap = nc;
    }
    uint32_t first = (uint32_t)tk->nedges;
    tk->nedges += groups;
    tk->nodes[node].first_edge = first;
    tk->nodes[node].n_edges = (uint16_t)groups;
    size_t g = 0;
    for (size_t i = lo; i < hi;) {
        unsigned char c = (unsigned char)st[i].s[depth];
        size_t j = i + 1;
        while (j < hi && (unsigned char)st[j].s[depth] == c) ++j;
        uint32_t child = tb_new_node(b);
        if (b->oom) return;
        tk->edge_byte[first + g] = c;
        tk->edge_node[first + g] = child;
        tb_build(b, child, i, j, depth + 1);
        if (b->oom) return;
        ++g; i = j;
    }
}

static inline int32_t trie_child(const glc_tokenizer* tk, int32_t node, unsigned char c) {
    if (node == 0) return tk->root_child[c];
    const tnode* nd = &tk->nodes[node];
    const uint8_t* eb = tk->edge_byte + nd->first_edge;
    size_t lo = 0, hi = nd->n_edges;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (eb[mid] < c) lo = mid + 1; else hi = mid;
    }
    return (lo < nd->n_edges && eb[lo] == c) ? (int32_t)tk->edge_node[nd->first_edge + lo] : -1;
}

static int32_t vocab_lookup(const glc_tokenizer* tk, const char* s, size_t n) {
    if (n == 0 || !tk->nnodes) return -1;
    int32_t node = 0;
    for (size_t i = 0; i < n; ++i) {
        node = trie_child(tk, node, (unsigned char)s[i]);
        if (node < 0) return -1;
    }
    return tk->nodes[node].id;
}

static int build_trie(glc_tokenizer* tk) {
    sort_tok* st = (sort_tok*)malloc((tk->nvocab ? tk->nvocab : 1) * sizeof(sort_tok));
    if (!st) return 0;
    size_t n = 0;
    for (size_t i = 0; i < tk->nvocab; ++i)
        if (tk->toklen[i]) { st[n].s = tk->tok[i]; st[n].len = tk->toklen[i]; st[n].id = (int32_t)i; ++n; }
    qsort(st, n, sizeof(sort_tok), cmp_sort_tok);
    trie_builder b = {tk, st, 0, 0, 0};
    uint32_t root = tb_new_node(&b);
    if (!b.oom) tb_build(&b, root, 0, n, 0);
    free(st);
    if (b.oom) return 0;
    for (int c = 0; c < 256; ++c) tk->root_child[c] = -1;
    const tnode* r = &tk->nodes[0];
    for (size_t e = 0; e < r->n_edges; ++e) tk->root_child[tk->edge_byte[r->first_edge + e]] = (int32_t)tk->edge_node[r->first_edge + e];
    return 1;
}

/* ---------------------------------------------------------------- normalisers */

/* spm_precompiled's DoubleArray::common_prefix_search, first result only; -1 = no prefix of `key` is in the map. */
static long precompiled_first_match(const norm_step* st, const unsigned char* key, size_t n) {
    const uint32_t* a = st->trie;
    if (!st->ntrie) return -1;
    size_t node = 0;
    uint32_t unit = a[0];
    node ^= (size_t)((unit >> 10) << ((unit & (1u << 9)) >> 6));
    for (size_t i = 0; i < n; ++i) {
        unsigned char c = key[i];
        if (c == 0) break;
        node ^= c;
        if (node >= st->ntrie) return -1;
        unit = a[node];
        if ((unit & ((1u << 31) | 0xFFu)) != c) return -1;
        node ^= (size_t)((unit >> 10) << ((unit & (1u << 9)) >> 6));
        if ((unit >> 8) & 1u) {
            if (node >= st->ntrie) return -1;
            return (long)(a[node] & 0x7FFFFFFFu);
        }
    }
    return -1;
}

static int precompiled_emit(const norm_step* st, const unsigned char* chunk, size_t n, sbuf* out) {
    long idx = precompiled_first_match(st, chunk, n);
    if (idx < 0) return 0;
    size_t b = (size_t)idx, e = b;
    while (e < st->nblob && st->blob[e] != 0) ++e;
    if (b <= st->nblob) sb_put(out, st->blob + b, e - b);
    return 1;
}

static void norm_precompiled(const norm_step* st, const char* in, size_t n, sbuf* out, int* lead_removed) {
    const unsigned char* s = (const unsigned char*)in;
    size_t i = 0;
    const size_t out0 = out->len;
    while (i < n) {
        if (i > 0 && out->len == out0) *lead_removed = 1;          /* everything consumed so far was rewritten to nothing */
        size_t e = grapheme_end(s, i, n);
        if (e - i < 6 && precompiled_emit(st, s + i, e - i, out)) { i = e; continue; }
        while (i < e) {
            size_t l = u8_len(s[i]);
            if (i + l > e) l = e - i;
            if (!precompiled_emit(st, s + i, l, out)) sb_put(out, in + i, l);
            i += l;
        }
    }
}

static void norm_strip(const norm_step* st, const char* in, size_t n, sbuf* out, int* lead_removed) {
    const unsigned char* s = (const unsigned char*)in;
    size_t b = 0, e = n, adv;
    if (st->left) while (b < n) { uint32_t c = u8_decode(s + b, n - b, &adv); if (!is_white_space(c)) break; b += adv; }
    if (st->right) {
        while (e > b) {
            size_t k = e - 1;
            while (k > b && (s[k] & 0xC0) == 0x80) --k;
            uint32_t c = u8_decode(s + k, e - k, &adv);
            if (!is_white_space(c)) break;
            e = k;
        }
    }
    if (b > 0) *lead_removed = 1;
    sb_put(out, in + b, e - b);
}

static void norm_replace_lit(const norm_step* st, const char* in, size_t n, sbuf* out) {
    if (!st->patlen) { sb_put(out, in, n); return; }
    size_t i = 0, run = 0;
    while (i + st->patlen <= n) {
        if (in[i] == st->pat[0] && !memcmp(in + i, st->pat, st->patlen)) {
            sb_put(out, in + run, i - run); sb_put(out, st->rep, st->replen);
            i += st->patlen; run = i;
        } else ++i;
    }
    sb_put(out, in + run, n - run);
}

static void norm_replace_run(const norm_step* st, const char* in, size_t n, sbuf* out) {
    const unsigned char* s = (const unsigned char*)in;
    size_t i = 0, run = 0, adv;
    while (i < n) {
        size_t j = i; int cnt = 0;
        while (j < n) {
            uint32_t c = u8_decode(s + j, n - j, &adv);
            if (!(st->atom_ws ? is_white_space(c) : c == st->atom_cp)) break;
            j += adv; ++cnt;
        }
        if (cnt >= st->min_rep && cnt > 0) {
            sb_put(out, in + run, i - run); sb_put(out, st->rep, st->replen);
            i = j; run = i;
        } else if (cnt > 0) i = j;
        else i += u8_len(s[i]);
    }
    sb_put(out, in + run, n - run);
}

static void norm_lowercase(const char* in, size_t n, sbuf* out) {
    const unsigned char* s = (const unsigned char*)in;
    const size_t nmap = sizeof(glc_lower_map) / sizeof(glc_lower_map[0]);
    size_t i = 0, adv; char tmp[8];
    while (i < n) {
        uint32_t c = u8_decode(s + i, n - i, &adv);
        if (c < 0x80) { char ch = (c >= 'A' && c <= 'Z') ? (char)(c + 32) : (char)c; sb_put(out, &ch, 1); i += adv; continue; }
        size_t lo = 0, hi = nmap; int found = 0;
        while (lo < hi) {
            size_t mid = (lo + hi) / 2;
            if (glc_lower_map[mid].cp < c) lo = mid + 1; else if (glc_lower_map[mid].cp > c) hi = mid;
            else { sb_put(out, tmp, u8_encode(tmp, glc_lower_map[mid].lo0)); if (glc_lower_map[mid].lo1) sb_put(out, tmp, u8_encode(tmp, glc_lower_map[mid].lo1)); found = 1; break; }
        }
        if (!found) sb_put(out, in + i, adv);
        i += adv;
    }
}

static void norm_nfc(const char* in, size_t n, sbuf* out);
/* Runs the whole chain; result in *a (b is scratch).  Returns 1 when the normalisers removed the START of the text (stripped
 * whitespace, a leading character rewritten to nothing): the piece then no longer begins at original offset 0, which is what
 * Metaspace's prepend_scheme "first" looks at. */
static int normalize_chain(const glc_tokenizer* tk, const char* in, size_t n, sbuf* a, sbuf* b) {
    int lead_removed = 0;
    sb_clear(a); sb_put(a, in, n);
    if (!a->s) return 0;
    for (size_t k = 0; k < tk->nsteps; ++k) {
        const norm_step* st = &tk->steps[k];
        sb_clear(b); sb_reserve(b, a->len);
        switch (st->kind) {
            case N_STRIP: norm_strip(st, a->s, a->len, b, &lead_removed); break;
            case N_PRECOMPILED: norm_precompiled(st, a->s, a->len, b, &lead_removed); break;
            case N_REPLACE_LIT: norm_replace_lit(st, a->s, a->len, b); break;
            case N_REPLACE_RUN: norm_replace_run(st, a->s, a->len, b); break;
            case N_LOWERCASE: norm_lowercase(a->s, a->len, b); break;
            case N_PREPEND: if (a->len) sb_put(b, st->pat, st->patlen); sb_put(b, a->s, a->len); break;
            case N_NFC: norm_nfc(a->s, a->len, b); break;
            default: sb_put(b, a->s, a->len);
        }
        sbuf t = *a; *a = *b; *b = t;
        if (!a->s) { sb_put(a, "", 0); }
    }
    return lead_removed;
}

/* ---------------------------------------------------------------- Unigram (Viterbi) */

typedef struct { double score; int32_t start; int32_t id; } vnode;
typedef struct { vnode* nodes; size_t cap; sbuf norm_a, norm_b, pre, fused; int32_t* seg; size_t seg_cap; } scratch;

static void emit_token(const glc_tokenizer* tk, const char* s, size_t n, int32_t id_hint, ivec* out) {
    int32_t id = id_hint >= 0 ? id_hint : vocab_lookup(tk, s, n);
    if (id >= 0) { iv_push(out, id); return; }
    if (tk->byte_fallback) {
        int all = 1;
        for (size_t i = 0; i < n; ++i) if (tk->byte_ids[(unsigned char)s[i]] < 0) { all = 0; break; }
        if (all) { for (size_t i = 0; i < n; ++i) iv_push(out, tk->byte_ids[(unsigned char)s[i]]); return; }
    }
    if (tk->unk_id >= 0) iv_push(out, tk->unk_id);
    else fprintf(stderr, "Error: tokenizer: unknown piece and the model has no unk_id\n");
}

static void unigram_encode(const glc_tokenizer* tk, const char* s, size_t n, scratch* sc, ivec* out) {
    if (!n) return;
    if (sc->cap < n + 1) {
        size_t nc = sc->cap ? sc->cap : 256;
        while (nc < n + 1) nc *= 2;
        vnode* nn = (vnode*)realloc(sc->nodes, nc * sizeof(vnode));
        if (!nn) return;
        sc->nodes = nn; sc->cap = nc;
    }
    vnode* best = sc->nodes;
    for (size_t i = 0; i <= n; ++i) { best[i].score = 0.0; best[i].start = -1; best[i].id = 0; }
    const unsigned char* u = (const unsigned char*)s;
    size_t at = 0;
    while (at < n) {
        const double here = best[at].score;
        size_t mblen = u8_len(u[at]);
        if (at + mblen > n) mblen = n - at;
        int has_single = 0;
        int32_t node = 0;
        for (size_t k = at; k < n; ++k) {
            node = trie_child(tk, node, u[k]);
            if (node < 0) break;
            int32_t id = tk->nodes[node].id;
            if (id < 0) continue;
            size_t key_pos = k + 1;
            vnode* t = &best[key_pos];
            double cand = tk->score[id] + here;
            if (t->start < 0 || cand > t->score) { t->score = cand; t->start = (int32_t)at; t->id = id; }
            if (!has_single && key_pos - at == mblen) has_single = 1;
        }
        if (!has_single) {
            vnode* t = &best[at + mblen];
            double cand = tk->unk_score + here;
            if (t->start < 0 || cand > t->score) { t->score = cand; t->start = (int32_t)at; t->id = tk->unk_id; }
        }
        at += mblen;
    }
    /* backtrack into (start) segments, then emit forward, fusing runs of unk */
    size_t nseg = 0, ends = n;
    while (ends > 0) {
        if (nseg == sc->seg_cap) {
            size_t nc = sc->seg_cap ? sc->seg_cap * 2 : 256;
            int32_t* ns = (int32_t*)realloc(sc->seg, nc * sizeof(int32_t));
            if (!ns) return;
            sc->seg = ns; sc->seg_cap = nc;
        }
        sc->seg[nseg++] = (int32_t)ends;
        ends = (size_t)best[ends].start;
    }
    size_t pos = 0, k = nseg;
    while (k > 0) {
        size_t e = (size_t)sc->seg[k - 1];
        int32_t id = best[e].id;
        if (tk->unk_id >= 0 && id == tk->unk_id) {                       /* fuse_unk: extend over the following unk segments */
            size_t fe = e, kk = k - 1;
            while (kk > 0 && best[sc->seg[kk - 1]].id == tk->unk_id) { fe = (size_t)sc->seg[kk - 1]; --kk; }
            emit_token(tk, s + pos, fe - pos, -1, out);
            pos = fe; k = kk;
        } else {
            emit_token(tk, s + pos, e - pos, id, out);
            pos = e; --k;
        }
    }
}

/* ---------------------------------------------------------------- pre-tokeniser + pieces */

static void bpe_encode_piece(const glc_tokenizer* tk, const char* s, size_t n, scratch* sc, ivec* out);
static void encode_text_piece(const glc_tokenizer* tk, const char* s, size_t n, int is_first, scratch* sc, ivec* out) {
    if (!n) return;
    if (tk->model_kind == MODEL_BPE) { bpe_encode_piece(tk, s, n, sc, out); return; }
    if (!tk->has_metaspace) { unigram_encode(tk, s, n, sc, out); return; }
    sbuf* b = &sc->pre;
    sb_clear(b); sb_reserve(b, n * tk->ms_replen + tk->ms_replen);
    for (size_t i = 0; i < n; ++i) {
        if (s[i] == ' ') sb_put(b, tk->ms_rep, tk->ms_replen); else sb_put(b, s + i, 1);
    }
    const int starts = b->len >= tk->ms_replen && !memcmp(b->s, tk->ms_rep, tk->ms_replen);
    const char* p = b->s; size_t len = b->len;
    if (!starts && (tk->ms_scheme == 0 || (tk->ms_scheme == 1 && is_first))) {
        sbuf* f = &sc->fused;
        sb_clear(f); sb_put(f, tk->ms_rep, tk->ms_replen); sb_put(f, b->s, b->len);
        p = f->s; len = f->len;
    }
    if (!tk->ms_split) { unigram_encode(tk, p, len, sc, out); return; }
    size_t start = 0, i = 0;
    while (i < len) {                                                    /* MergedWithNext: cut before every replacement char */
        if (i + tk->ms_replen <= len && p[i] == tk->ms_rep[0] && !memcmp(p + i, tk->ms_rep, tk->ms_replen)) {
            if (i > start) { unigram_encode(tk, p + start, i - start, sc, out); start = i; }
            i += tk->ms_replen;
        } else ++i;
    }
    if (start < len) unigram_encode(tk, p + start, len - start, sc, out);
}

static int is_word_char(uint32_t c) {                /* \w of the Rust regex crate (table probed from the library, glc_unicode_tables.h) */
    if (c < 0x80) return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_';
    return range_lookup(glc_word_ranges, sizeof(glc_word_ranges) / sizeof(glc_word_ranges[0]), c);
}

/* AddedVocabulary::find_matches over patterns with the given `normalized` flag; calls cb for each split. */
typedef void (*split_cb)(void* ctx, const char* s, size_t n, int32_t id);

static void split_added(const glc_tokenizer* tk, int normalized, const char* s, size_t n, split_cb cb, void* ctx) {
    if (n == 0) return;
    size_t start_offset = 0, i = 0;
    int any = 0;
    for (size_t a = 0; a < tk->nadded; ++a) if (tk->added[a].normalized == normalized && tk->added[a].plen) { any = 1; break; }
    if (!any) { cb(ctx, s, n, -1); return; }
    const unsigned char* u = (const unsigned char*)s;
    while (i < n) {
        const added_tok* best = NULL;
        for (size_t a = 0; a < tk->nadded; ++a) {
            const added_tok* t = &tk->added[a];
            if (t->normalized != normalized || !t->plen || t->pattern[0] != s[i] || i + t->plen > n) continue;
            if (memcmp(s + i, t->pattern, t->plen)) continue;
            if (!best || t->plen > best->plen) best = t;
        }
        if (!best) { ++i; continue; }
        size_t start = i, stop = i + best->plen, adv;
        i = stop;                                                        /* the automaton moves on even when the match is rejected */
        if (best->single_word) {
            int start_space = 1, stop_space = 1;
            if (start > 0) { size_t k = start - 1; while (k > 0 && (u[k] & 0xC0) == 0x80) --k; start_space = !is_word_char(u8_decode(u + k, start - k, &adv)); }
            if (stop < n) stop_space = !is_word_char(u8_decode(u + stop, n - stop, &adv));
            if (!start_space || !stop_space) continue;
        }
        if (best->lstrip) {
            size_t ns = start;
            while (ns > 0) { size_t k = ns - 1; while (k > 0 && (u[k] & 0xC0) == 0x80) --k; if (!is_white_space(u8_decode(u + k, ns - k, &adv))) break; ns = k; }
            start = ns > start_offset ? ns : start_offset;
        }
        if (best->rstrip) {
            while (stop < n) { uint32_t c = u8_decode(u + stop, n - stop, &adv); if (!is_white_space(c)) break; stop += adv; }
            if (stop > i) i = stop;
        }
        if (start_offset < start) cb(ctx, s + start_offset, start - start_offset, -1);
        cb(ctx, s + start, stop - start, best->id);
        start_offset = stop;
    }
    if (start_offset != n) cb(ctx, s + start_offset, n - start_offset, -1);
}

typedef struct { const glc_tokenizer* tk; scratch* sc; ivec* out; int first; } enc_ctx;

static void on_norm_split(void* vctx, const char* s, size_t n, int32_t id) {
    enc_ctx* c = (enc_ctx*)vctx;
    if (id >= 0) iv_push(c->out, id);
    else if (n) encode_text_piece(c->tk, s, n, c->first, c->sc, c->out);
    c->first = 0;
}

static void on_raw_split(void* vctx, const char* s, size_t n, int32_t id) {
    enc_ctx* c = (enc_ctx*)vctx;
    if (id >= 0) { iv_push(c->out, id); c->first = 0; return; }
    if (!n) return;
    scratch* sc = c->sc;
    if (normalize_chain(c->tk, s, n, &sc->norm_a, &sc->norm_b)) c->first = 0;     /* no longer at original offset 0 */
    if (sc->norm_a.len) split_added(c->tk, 1, sc->norm_a.s, sc->norm_a.len, on_norm_split, c);   /* the pre-tokeniser uses other scratch buffers */
    c->first = 0;
}

static void scratch_free(scratch* sc) {
    free(sc->nodes); free(sc->seg); sb_free(&sc->norm_a); sb_free(&sc->norm_b); sb_free(&sc->pre); sb_free(&sc->fused);
}

static void encode_one(const glc_tokenizer* tk, const char* text, size_t len, int add_special, TokenizerEncodeResult* res) {
    scratch sc; memset(&sc, 0, sizeof(sc));
    ivec body = {0, 0, 0};
    sbuf clean = {0, 0, 0};
    u8_sanitize(text, len, &clean);
    enc_ctx ctx = {tk, &sc, &body, 1};
    if (clean.len) split_added(tk, 0, clean.s, clean.len, on_raw_split, &ctx);
    sb_free(&clean);
    size_t nb = body.n, skip = 0;
    if (tk->trunc_max) {
        size_t extra = add_special ? tk->tmpl_specials : 0;
        size_t room = tk->trunc_max > extra ? tk->trunc_max - extra : 0;
        if (nb > room) { if (tk->trunc_left) skip = nb - room; nb = room; }
    }
    ivec out = {0, 0, 0};
    if (add_special && tk->ntmpl) {
        for (size_t t = 0; t < tk->ntmpl; ++t) {
            if (tk->tmpl[t].is_special) for (size_t k = 0; k < tk->tmpl[t].n; ++k) iv_push(&out, tk->tmpl[t].ids[k]);
            else for (size_t k = 0; k < nb; ++k) iv_push(&out, body.v[skip + k]);
        }
    } else {
        for (size_t k = 0; k < nb; ++k) iv_push(&out, body.v[skip + k]);
    }
    free(body.v);
    scratch_free(&sc);
    if (!out.v) out.v = (int32_t*)malloc(sizeof(int32_t));                /* len 0 still gets a valid pointer */
    res->token_ids = (int*)out.v;
    res->len = out.n;
}

/* ---------------------------------------------------------------- NFC (normalizer of the Qwen2 / GPT-style tokenizer.json files) */

/* Data: glc_bpe_tables.h (scripts/gen_bpe_tables.py: unicodedata, checked code point by code point against the Rust library).
 * Algorithm: UAX #15 — canonical decomposition (recursive, Hangul arithmetic), canonical ordering, canonical composition. */
static int nfc_ccc(uint32_t c) {
    if (c < 0x300) return 0;
    size_t lo = 0, hi = sizeof(glc_nfc_ccc) / sizeof(glc_nfc_ccc[0]);
    while (lo < hi) { size_t m = (lo + hi) / 2; if (glc_nfc_ccc[m].cp < c) lo = m + 1; else hi = m; }
    return lo < sizeof(glc_nfc_ccc) / sizeof(glc_nfc_ccc[0]) && glc_nfc_ccc[lo].cp == c ? glc_nfc_ccc[lo].ccc : 0;
}
static const glc_decomp* nfc_decomp(uint32_t c) {
    if (c < 0xC0) return NULL;
    size_t lo = 0, hi = sizeof(glc_nfc_decomp) / sizeof(glc_nfc_decomp[0]);
    while (lo < hi) { size_t m = (lo + hi) / 2; if (glc_nfc_decomp[m].cp < c) lo = m + 1; else hi = m; }
    return lo < sizeof(glc_nfc_decomp) / sizeof(glc_nfc_decomp[0]) && glc_nfc_decomp[lo].cp == c ? &glc_nfc_decomp[lo] : NULL;
}
enum { H_SB = 0xAC00, H_LB = 0x1100, H_VB = 0x1161, H_TB = 0x11A7, H_LC = 19, H_VC = 21, H_TC = 28, H_NC = 21 * 28, H_SC = 19 * 21 * 28 };
static void nfc_push_decomposed(uint32_t c, uint32_t** buf, size_t* n, size_t* cap) {
    if (*n + 4 > *cap) {
        const size_t nc = *cap ? *cap * 2 : 64;
        uint32_t* nb = (uint32_t*)realloc(*buf, nc * sizeof(uint32_t));
        if (!nb) return;                                                  /* out of memory: the character is dropped, the buffer stays valid */
        *buf = nb; *cap = nc;
    }
    if (c >= H_SB && c < H_SB + H_SC) {
        const uint32_t si = c - H_SB;
        (*buf)[(*n)++] = H_LB + si / H_NC; (*buf)[(*n)++] = H_VB + (si % H_NC) / H_TC;
        if (si % H_TC) (*buf)[(*n)++] = H_TB + si % H_TC;
        return;
    }
    const glc_decomp* d = nfc_decomp(c);
    if (!d) { (*buf)[(*n)++] = c; return; }
    nfc_push_decomposed(d->a, buf, n, cap);
    if (d->b) nfc_push_decomposed(d->b, buf, n, cap);
}
static uint32_t nfc_compose_pair(uint32_t a, uint32_t b) {
    if (a >= H_LB && a < H_LB + H_LC && b >= H_VB && b < H_VB + H_VC) return H_SB + ((a - H_LB) * H_VC + (b - H_VB)) * H_TC;
    if (a >= H_SB && a < H_SB + H_SC && (a - H_SB) % H_TC == 0 && b > H_TB && b < H_TB + H_TC) return a + (b - H_TB);
    size_t lo = 0, hi = sizeof(glc_nfc_comp) / sizeof(glc_nfc_comp[0]);
    while (lo < hi) {
        size_t m = (lo + hi) / 2;
        if (glc_nfc_comp[m].a < a || (glc_nfc_comp[m].a == a && glc_nfc_comp[m].b < b)) lo = m + 1; else hi = m;
    }
    return lo < sizeof(glc_nfc_comp) / sizeof(glc_nfc_comp[0]) && glc_nfc_comp[lo].a == a && glc_nfc_comp[lo].b == b ? glc_nfc_comp[lo].cp : 0;
}
static void norm_nfc(const char* in, size_t n, sbuf* out) {
    size_t i = 0;
    while (i < n && (unsigned char)in[i] < 0x80) ++i;
    if (i == n) { sb_put(out, in, n); return; }                          /* ASCII is NFC */
    uint32_t* d = NULL; size_t nd = 0, cap = 0;
    const unsigned char* u = (const unsigned char*)in;
    for (i = 0; i < n;) {
        size_t adv; uint32_t c = u8_decode(u + i, n - i, &adv);
        i += adv;
        const glc_nfc_over* ov = NULL;
        for (const glc_nfc_over* o = glc_nfc_override; o->cp; ++o) if (o->cp == c) { ov = o; break; }
        if (ov) { for (int k = 0; k < 3 && ov->r[k]; ++k) nfc_push_decomposed(ov->r[k], &d, &nd, &cap); }
        else nfc_push_decomposed(c, &d, &nd, &cap);
    }
    for (size_t k = 1; k < nd; ++k) {                                    /* canonical ordering: stable sort of each run of non-starters */
        const int cc = nfc_ccc(d[k]);
        if (!cc) continue;
        size_t j = k; const uint32_t v = d[k];
        while (j > 0) { const int pc = nfc_ccc(d[j - 1]); if (pc <= cc) break; d[j] = d[j - 1]; --j; }
        d[j] = v;
    }
    if (nd) {                                                            /* canonical composition */
        size_t starter = 0, comp = 1;
        uint32_t sch = d[0];
        int last = nfc_ccc(sch) ? 256 : 0;
        for (size_t k = 1; k < nd; ++k) {
            const uint32_t ch = d[k]; const int cc = nfc_ccc(ch);
            const uint32_t c2 = nfc_compose_pair(sch, ch);
            if (c2 && (last < cc || last == 0)) { d[starter] = c2; sch = c2; }
            else { if (cc == 0) { starter = comp; sch = ch; } last = cc; d[comp++] = ch; }
        }
        nd = comp;
    }
    sb_reserve(out, nd * 4 + 4);
    for (size_t k = 0; k < nd; ++k) { char t[4]; const size_t l = u8_encode(t, d[k]); sb_put(out, t, l); }
    free(d);
}

/* ---------------------------------------------------------------- byte-level BPE */

static int crange_lookup(const glc_crange* t, size_t n, uint32_t c) {
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t m = (lo + hi) / 2; if (t[m].hi < c) lo = m + 1; else hi = m; }
    return lo < n && t[lo].lo <= c;
}
static int is_letter(uint32_t c) {                  /* \p{L} of the pre-tokenizer's regex engine (probed from the Rust library) */
    if (c < 0x80) return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
    return crange_lookup(glc_letter_ranges, sizeof(glc_letter_ranges) / sizeof(glc_letter_ranges[0]), c);
}
static int is_number(uint32_t c) {                  /* \p{N} */
    if (c < 0x80) return c >= '0' && c <= '9';
    return crange_lookup(glc_number_ranges, sizeof(glc_number_ranges) / sizeof(glc_number_ranges[0]), c);
}
static inline uint32_t cp_at(const unsigned char* s, size_t n, size_t i, size_t* adv) {
    if (i >= n) { *adv = 0; return 0xFFFFFFFFu; }
    return u8_decode(s + i, n - i, adv);
}
static inline int is_punct_class(uint32_t c) { return c != 0xFFFFFFFFu && !is_white_space(c) && !is_letter(c) && !is_number(c); }   /* [^\s\p{L}\p{N}] */

/* End (byte offset) of the pre-token that starts at byte i, for the two split patterns byte-level BPE tokenizers ship with:
 *   GPT-2 : 's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
 *   Qwen2 : (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
 * Leftmost match, alternatives in order, greedy with backtracking (Oniguruma semantics); every position matches some alternative. */
static size_t bl_next(int kind, const unsigned char* s, size_t n, size_t i) {
    size_t a0, a1, a2;
    const uint32_t c0 = cp_at(s, n, i, &a0);
    const uint32_t c1 = cp_at(s, n, i + a0, &a1);
    if (c0 == '\'') {                                                     /* contractions */
        const uint32_t c2 = cp_at(s, n, i + a0 + a1, &a2);
        const int ci = kind == BLRE_QWEN;
        uint32_t l1 = c1, l2 = c2;
        if (ci) { if (l1 >= 'A' && l1 <= 'Z') l1 += 32; if (l2 >= 'A' && l2 <= 'Z') l2 += 32; if (l1 == 0x17F) l1 = 's'; }
        if (l1 == 's' || l1 == 't') return i + a0 + a1;
        if ((l1 == 'r' && l2 == 'e') || (l1 == 'v' && l2 == 'e')) return i + a0 + a1 + a2;
        if (l1 == 'm') return i + a0 + a1;
        if (l1 == 'l' && l2 == 'l') return i + a0 + a1 + a2;
        if (l1 == 'd') return i + a0 + a1;
    }
    if (kind == BLRE_QWEN) {
        size_t j = i;                                                     /* [^\r\n\p{L}\p{N}]?\p{L}+ */
        if (c0 != '\r' && c0 != '\n' && !is_letter(c0) && !is_number(c0) && is_letter(c1)) j = i + a0;
        if (is_letter(cp_at(s, n, j, &a2))) {
            while (j < n) { const uint32_t c = cp_at(s, n, j, &a2); if (!is_letter(c)) break; j += a2; }
            return j;
        }
        if (is_number(c0)) return i + a0;                                 /* \p{N} */
    } else {
        size_t j = (c0 == ' ' && (is_letter(c1) || is_number(c1))) ? i + a0 : i;
        const uint32_t f = cp_at(s, n, j, &a2);
        if (is_letter(f)) { while (j < n) { const uint32_t c = cp_at(s, n, j, &a2); if (!is_letter(c)) break; j += a2; } return j; }
        if (is_number(f)) { while (j < n) { const uint32_t c = cp_at(s, n, j, &a2); if (!is_number(c)) break; j += a2; } return j; }
    }
    {                                                                     /*  ?[^\s\p{L}\p{N}]+ ([\r\n]* for Qwen) */
        size_t j = (c0 == ' ' && is_punct_class(c1)) ? i + a0 : i;
        if (is_punct_class(cp_at(s, n, j, &a2))) {
            while (j < n) { const uint32_t c = cp_at(s, n, j, &a2); if (!is_punct_class(c)) break; j += a2; }
            if (kind == BLRE_QWEN) while (j < n && (s[j] == '\r' || s[j] == '\n')) ++j;
            return j;
        }
    }
    /* whitespace run [i, e); last_nl = end of its last \r / \n; prev = start of its last character */
    size_t e = i, last_nl = 0, prev = i, count = 0;
    while (e < n) {
        const uint32_t c = cp_at(s, n, e, &a2);
        if (!is_white_space(c)) break;
        prev = e; e += a2; ++count;
        if (c == '\r' || c == '\n') last_nl = e;
    }
    if (kind == BLRE_QWEN && last_nl) return last_nl;                     /* \s*[\r\n]+ */
    if (e == n) return e;                                                 /* \s+(?!\S) at the end of the text */
    if (count >= 2) return prev;                                          /* \s+(?!\S): all but the last whitespace character */
    return e > i ? e : i + a0;                                            /* \s+ (or, defensively, one character) */
}

static void bl_init_tables(glc_tokenizer* tk) {                          /* GPT-2 bytes_to_unicode */
    int next = 0;
    for (int b = 0; b < 512; ++b) tk->bl_byte_of[b] = -1;
    for (int b = 0; b < 256; ++b) {
        const int keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255);
        tk->bl_cp[b] = (uint16_t)(keep ? b : 256 + next++);
        tk->bl_byte_of[tk->bl_cp[b]] = (int16_t)b;
    }
}

static const bpe_merge* merge_find(const glc_tokenizer* tk, int32_t l, int32_t r) {
    if (!tk->merge_cap) return NULL;
    const uint64_t key = ((uint64_t)(uint32_t)l << 32) | (uint32_t)r | (1ull << 63);
    size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (tk->merge_cap - 1);
    while (tk->merges[h].key) { if (tk->merges[h].key == key) return &tk->merges[h]; h = (h + 1) & (tk->merge_cap - 1); }
    return NULL;
}

/* Long pre-tokens (a run of thousands of letters without whitespace): the rescan below is O(n^2) hash lookups; this is the Rust
 * implementation's own method — symbols as a linked list, candidate pairs in a min-heap ordered by (rank, position), stale entries
 * (a side already merged away, or merged into something else) skipped when popped — O(n log n), same merge order. */
#define BPE_HEAP_MIN 48
typedef struct { int32_t rank; uint32_t pos; int32_t l, r, merged; } bpe_cand;
static int cand_less(const bpe_cand* a, const bpe_cand* b) { return a->rank != b->rank ? a->rank < b->rank : a->pos < b->pos; }
static void heap_push(bpe_cand* h, size_t* n, bpe_cand c) {
    size_t i = (*n)++;
    while (i) { const size_t p = (i - 1) / 2; if (!cand_less(&c, &h[p])) break; h[i] = h[p]; i = p; }
    h[i] = c;
}
static bpe_cand heap_pop(bpe_cand* h, size_t* n) {
    const bpe_cand top = h[0], last = h[--(*n)];
    size_t i = 0;
    for (;;) {
        size_t k = 2 * i + 1;
        if (k >= *n) break;
        if (k + 1 < *n && cand_less(&h[k + 1], &h[k])) ++k;
        if (!cand_less(&h[k], &last)) break;
        h[i] = h[k]; i = k;
    }
    if (*n) h[i] = last;
    return top;
}
static int bpe_merge_heap(const glc_tokenizer* tk, int32_t* sym, size_t* pns) {
    const size_t ns = *pns;
    int32_t* prev = (int32_t*)malloc(ns * sizeof(int32_t));
    int32_t* next = (int32_t*)malloc(ns * sizeof(int32_t));
    bpe_cand* heap = (bpe_cand*)malloc(3 * ns * sizeof(bpe_cand));       /* n - 1 initial pairs + at most two per merge */
    if (!prev || !next || !heap) { free(prev); free(next); free(heap); return 0; }
    size_t hn = 0;
    for (size_t k = 0; k < ns; ++k) { prev[k] = (int32_t)k - 1; next[k] = k + 1 < ns ? (int32_t)k + 1 : -1; }
    for (size_t k = 0; k + 1 < ns; ++k) {
        const bpe_merge* m = merge_find(tk, sym[k], sym[k + 1]);
        if (m) heap_push(heap, &hn, (bpe_cand){m->rank, (uint32_t)k, sym[k], sym[k + 1], m->merged});
    }
    while (hn) {
        const bpe_cand c = heap_pop(heap, &hn);
        const int32_t i = (int32_t)c.pos, j = sym[i] >= 0 ? next[i] : -1;
        if (j < 0 || sym[i] != c.l || sym[j] != c.r) continue;            /* stale: a side was merged since this pair was queued */
        sym[i] = c.merged;
        sym[j] = -1;                                                      /* dead */
        next[i] = next[j];
        if (next[j] >= 0) prev[next[j]] = i;
        if (prev[i] >= 0) { const bpe_merge* m = merge_find(tk, sym[prev[i]], sym[i]); if (m) heap_push(heap, &hn, (bpe_cand){m->rank, (uint32_t)prev[i], sym[prev[i]], sym[i], m->merged}); }
        if (next[i] >= 0) { const bpe_merge* m = merge_find(tk, sym[i], sym[next[i]]); if (m) heap_push(heap, &hn, (bpe_cand){m->rank, (uint32_t)i, sym[i], sym[next[i]], m->merged}); }
    }
    size_t o = 0;
    for (int32_t k = 0; k >= 0; k = next[k]) sym[o++] = sym[k];          /* symbol 0 never dies (a merge keeps its left slot) */
    *pns = o;
    free(prev); free(next); free(heap);
    return 1;
}

/* One pre-token (already mapped to its printable stand-ins, UTF-8): symbols = characters, then merge the lowest-rank adjacent
 * pair (leftmost among equals) until none is left — the order the Rust implementation's heap pops them in. */
static void bpe_word(const glc_tokenizer* tk, const char* w, size_t n, scratch* sc, ivec* out) {
    if (tk->bpe_ignore_merges) { const int32_t id = vocab_lookup(tk, w, n); if (id >= 0) { iv_push(out, id); return; } }
    if (sc->seg_cap < n + 1) { free(sc->seg); sc->seg_cap = n + 64; sc->seg = (int32_t*)malloc(sc->seg_cap * sizeof(int32_t)); if (!sc->seg) { sc->seg_cap = 0; return; } }
    int32_t* sym = sc->seg; size_t ns = 0;
    for (size_t i = 0; i < n;) {
        const size_t l = u8_len((unsigned char)w[i]) <= n - i ? u8_len((unsigned char)w[i]) : n - i;
        const int32_t id = vocab_lookup(tk, w + i, l);
        if (id >= 0) sym[ns++] = id; else if (tk->unk_id >= 0) sym[ns++] = tk->unk_id;        /* no unk token: the character is dropped */
        i += l;
    }
    if (ns > BPE_HEAP_MIN && bpe_merge_heap(tk, sym, &ns)) { for (size_t k = 0; k < ns; ++k) iv_push(out, sym[k]); return; }
    while (ns > 1) {                                                      /* short words (and the heap's allocation-failure fallback): rescan */
        int32_t best = INT32_MAX; size_t at = 0; int32_t merged = -1;
        for (size_t k = 0; k + 1 < ns; ++k) {
            const bpe_merge* m = merge_find(tk, sym[k], sym[k + 1]);
            if (m && m->rank < best) { best = m->rank; at = k; merged = m->merged; }
        }
        if (merged < 0) break;
        sym[at] = merged;
        memmove(sym + at + 1, sym + at + 2, (ns - at - 2) * sizeof(int32_t));
        --ns;
    }
    for (size_t k = 0; k < ns; ++k) iv_push(out, sym[k]);
}

static void bpe_encode_piece(const glc_tokenizer* tk, const char* s, size_t n, scratch* sc, ivec* out) {
    const unsigned char* u = (const unsigned char*)s;
    sbuf* pre = &sc->pre; sbuf* w = &sc->fused;
    if (tk->pre_bytelevel && tk->bl_prefix_space && n && s[0] != ' ') {   /* ByteLevel add_prefix_space */
        sb_clear(pre); sb_put(pre, " ", 1); sb_put(pre, s, n);
        u = (const unsigned char*)pre->s; n = pre->len;
    }
    for (size_t i = 0; i < n;) {
        const size_t e = tk->bl_regex ? bl_next(tk->bl_regex, u, n, i) : n;
        sb_clear(w);
        if (tk->pre_bytelevel) {
            sb_reserve(w, 2 * (e - i) + 2);
            for (size_t k = i; k < e; ++k) { char t[4]; const size_t l = u8_encode(t, tk->bl_cp[u[k]]); sb_put(w, t, l); }
        } else sb_put(w, (const char*)u + i, e - i);
        if (w->len) bpe_word(tk, w->s, w->len, sc, out);
        i = e > i ? e : i + 1;
    }
}

/* ---------------------------------------------------------------- tokenizer.json loading */

static char* dup_n(const char* s, size_t n) {
    char* r = (char*)malloc(n + 1);
    if (r) { memcpy(r, s, n); r[n] = 0; }
    return r;
}

static const char* jstr(const gj_value* o, const char* key, size_t* len) {
    const gj_value* v = gj_get(o, key);
    if (!gj_is(v, GJ_STR)) return NULL;
    if (len) *len = v->u.str.len;
    return v->u.str.s;
}
static int jbool(const gj_value* o, const char* key, int dflt) {
    const gj_value* v = gj_get(o, key);
    return gj_is(v, GJ_BOOL) ? v->u.boolean : dflt;
}

static int b64_val(int c) {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+' || c == '-') return 62;
    if (c == '/' || c == '_') return 63;
    return -1;
}
static unsigned char* b64_decode(const char* s, size_t n, size_t* out_n) {
    unsigned char* o = (unsigned char*)malloc(n / 4 * 3 + 4);
    if (!o) return NULL;
    size_t k = 0; uint32_t acc = 0; int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        int v = b64_val((unsigned char)s[i]);
        if (v < 0) continue;                                             /* '=' padding / whitespace */
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; o[k++] = (unsigned char)((acc >> bits) & 0xFF); }
    }
    *out_n = k;
    return o;
}

static norm_step* add_step(glc_tokenizer* tk) {
    norm_step* ns = (norm_step*)realloc(tk->steps, (tk->nsteps + 1) * sizeof(norm_step));
    if (!ns) return NULL;
    tk->steps = ns;
    memset(&ns[tk->nsteps], 0, sizeof(norm_step));
    return &ns[tk->nsteps++];
}

/* "X{n,}" or "X+" where X is one literal character, an escaped character, or \s */
static int parse_run_regex(const char* re, size_t n, norm_step* st) {
    const unsigned char* u = (const unsigned char*)re;
    size_t i = 0, adv;
    if (!n) return 0;
    if (u[0] == '\\') {
        if (n < 2) return 0;
        if (u[1] == 's') { st->atom_ws = 1; i = 2; }
        else if (strchr("\\.+*?()[]{}|^$ /-", u[1])) { st->atom_cp = u[1]; i = 2; }
        else return 0;
    } else {
        if (strchr(".+*?()[]{}|^$", u[0])) return 0;
        st->atom_cp = u8_decode(u, n, &adv); i = adv;
    }
    if (i < n && u[i] == '+' && i + 1 == n) { st->min_rep = 1; return 1; }
    if (i < n && u[i] == '{') {
        char* endp = NULL;
        long m = strtol(re + i + 1, &endp, 10);
        if (endp && endp[0] == ',' && endp[1] == '}' && endp + 2 == re + n && m >= 1) { st->min_rep = (int)m; return 1; }
    }
    return 0;
}

static int load_normalizer(glc_tokenizer* tk, const gj_value* nz, char* err, size_t errlen) {
    if (!nz || nz->type == GJ_NULL) return 1;
    const char* type = jstr(nz, "type", NULL);
    if (!type) { snprintf(err, errlen, "normalizer without a type"); return 0; }
    if (!strcmp(type, "Sequence")) {
        const gj_value* lst = gj_get(nz, "normalizers");
        if (!gj_is(lst, GJ_ARR)) { snprintf(err, errlen, "normalizer Sequence without a list"); return 0; }
        for (size_t i = 0; i < lst->u.arr.n; ++i) if (!load_normalizer(tk, lst->u.arr.items[i], err, errlen)) return 0;
        return 1;
    }
    norm_step* st = add_step(tk);
    if (!st) { snprintf(err, errlen, "out of memory"); return 0; }
    if (!strcmp(type, "Strip")) { st->kind = N_STRIP; st->left = jbool(nz, "strip_left", 1); st->right = jbool(nz, "strip_right", 1); return 1; }
    if (!strcmp(type, "Lowercase")) { st->kind = N_LOWERCASE; return 1; }
    if (!strcmp(type, "NFC")) { st->kind = N_NFC; return 1; }
    if (!strcmp(type, "Prepend")) {
        size_t l; const char* p = jstr(nz, "prepend", &l);
        if (!p) { snprintf(err, errlen, "normalizer Prepend without text"); return 0; }
        st->kind = N_PREPEND; st->pat = dup_n(p, l); st->patlen = l; return st->pat != NULL;
    }
    if (!strcmp(type, "Precompiled")) {
        size_t l; const char* b64 = jstr(nz, "precompiled_charsmap", &l);
        st->kind = N_PRECOMPILED;
        if (!b64 || !l) return 1;                                        /* empty map = identity */
        size_t nb; unsigned char* raw = b64_decode(b64, l, &nb);
        if (!raw) { snprintf(err, errlen, "out of memory"); return 0; }
        uint32_t tsize = nb >= 4 ? ((uint32_t)raw[0] | ((uint32_t)raw[1] << 8) | ((uint32_t)raw[2] << 16) | ((uint32_t)raw[3] << 24)) : 0;
        if (nb < 4 || (size_t)tsize + 4 > nb || tsize % 4) { free(raw); snprintf(err, errlen, "malformed precompiled_charsmap"); return 0; }
        st->ntrie = tsize / 4;
        st->trie = (uint32_t*)malloc((st->ntrie ? st->ntrie : 1) * sizeof(uint32_t));
        st->nblob = nb - 4 - tsize;
        st->blob = (char*)malloc(st->nblob + 1);
        if (!st->trie || !st->blob) { free(raw); snprintf(err, errlen, "out of memory"); return 0; }
        for (size_t i = 0; i < st->ntrie; ++i) {
            const unsigned char* q = raw + 4 + 4 * i;
            st->trie[i] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        }
        memcpy(st->blob, raw + 4 + tsize, st->nblob); st->blob[st->nblob] = 0;
        free(raw);
        return 1;
    }
    if (!strcmp(type, "Replace")) {
        const gj_value* pat = gj_get(nz, "pattern");
        size_t rl; const char* rep = jstr(nz, "content", &rl);
        if (!pat || !rep) { snprintf(err, errlen, "normalizer Replace without pattern/content"); return 0; }
        st->rep = dup_n(rep, rl); st->replen = rl;
        size_t pl; const char* p;
        if ((p = jstr(pat, "String", &pl))) { st->kind = N_REPLACE_LIT; st->pat = dup_n(p, pl); st->patlen = pl; return st->rep && st->pat; }
        if ((p = jstr(pat, "Regex", &pl))) {
            st->kind = N_REPLACE_RUN;
            if (parse_run_regex(p, pl, st)) return st->rep != NULL;
            snprintf(err, errlen, "normalizer Replace: unsupported regex '%s' (supported: one character or \\s followed by + or {n,})", p);
            return 0;
        }
        snprintf(err, errlen, "normalizer Replace: unknown pattern kind");
        return 0;
    }
    snprintf(err, errlen, "unsupported normalizer '%s'", type);
    return 0;
}

static int load_pretokenizer(glc_tokenizer* tk, const gj_value* pt, char* err, size_t errlen) {
    if (!pt || pt->type == GJ_NULL) return 1;
    const char* type = jstr(pt, "type", NULL);
    if (!type) { snprintf(err, errlen, "pre_tokenizer without a type"); return 0; }
    if (!strcmp(type, "Sequence")) {
        const gj_value* lst = gj_get(pt, "pretokenizers");
        if (!gj_is(lst, GJ_ARR)) { snprintf(err, errlen, "pre_tokenizer Sequence without a list"); return 0; }
        for (size_t i = 0; i < lst->u.arr.n; ++i) if (!load_pretokenizer(tk, lst->u.arr.items[i], err, errlen)) return 0;
        return 1;
    }
    if (!strcmp(type, "Metaspace")) {
        if (tk->has_metaspace) { snprintf(err, errlen, "more than one Metaspace pre_tokenizer"); return 0; }
        size_t l; const char* rep = jstr(pt, "replacement", &l);
        if (!rep || !l || l >= sizeof(tk->ms_rep)) { snprintf(err, errlen, "Metaspace: bad replacement"); return 0; }
        memcpy(tk->ms_rep, rep, l); tk->ms_rep[l] = 0; tk->ms_replen = l;
        const char* scheme = jstr(pt, "prepend_scheme", NULL);
        if (scheme) tk->ms_scheme = !strcmp(scheme, "always") ? 0 : !strcmp(scheme, "first") ? 1 : 2;
        else tk->ms_scheme = jbool(pt, "add_prefix_space", 1) ? 0 : 2;     /* legacy files */
        tk->ms_split = jbool(pt, "split", 1);
        tk->has_metaspace = 1;
        return 1;
    }
    if (!strcmp(type, "Split")) {
        /* the split regex of the byte-level BPE families, recognised by its text (a general regex engine is not part of this library) */
        static const char QWEN[] = "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
        static const char GPT2[] = "'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";
        const gj_value* pat = gj_get(pt, "pattern");
        const char* re = pat ? jstr(pat, "Regex", NULL) : NULL;
        const char* beh = jstr(pt, "behavior", NULL);
        if (!re || !beh || strcmp(beh, "Isolated") || jbool(pt, "invert", 0)) { snprintf(err, errlen, "pre_tokenizer Split: only {Regex, Isolated, invert=false} is implemented"); return 0; }
        if (!strcmp(re, QWEN)) tk->bl_regex = BLRE_QWEN;
        else if (!strcmp(re, GPT2)) tk->bl_regex = BLRE_GPT2;
        else { snprintf(err, errlen, "pre_tokenizer Split: unsupported regex (implemented: the GPT-2 and the Qwen2 split patterns)"); return 0; }
        return 1;
    }
    if (!strcmp(type, "ByteLevel")) {
        tk->pre_bytelevel = 1;
        tk->bl_prefix_space = jbool(pt, "add_prefix_space", 1);
        if (jbool(pt, "use_regex", 1)) {
            if (tk->bl_regex) { snprintf(err, errlen, "pre_tokenizer ByteLevel: use_regex=true after a Split"); return 0; }
            tk->bl_regex = BLRE_GPT2;
        }
        bl_init_tables(tk);
        return 1;
    }
    snprintf(err, errlen, "unsupported pre_tokenizer '%s'", type);
    return 0;
}

/* BPE model: vocab {token: id}, merges ["a b", ...] or [["a", "b"], ...] (rank = position). */
static int load_model_bpe(glc_tokenizer* tk, const gj_value* m, char* err, size_t errlen) {
    const gj_value* vocab = gj_get(m, "vocab");
    const gj_value* merges = gj_get(m, "merges");
    if (!gj_is(vocab, GJ_OBJ) || !gj_is(merges, GJ_ARR)) { snprintf(err, errlen, "BPE model without vocab / merges"); return 0; }
    size_t l;
    const char* csp = jstr(m, "continuing_subword_prefix", &l);
    if (csp && l) { snprintf(err, errlen, "BPE: continuing_subword_prefix is not implemented"); return 0; }
    const char* eow = jstr(m, "end_of_word_suffix", &l);
    if (eow && l) { snprintf(err, errlen, "BPE: end_of_word_suffix is not implemented"); return 0; }
    if (jbool(m, "byte_fallback", 0)) { snprintf(err, errlen, "BPE: byte_fallback is not implemented"); return 0; }
    const gj_value* dr = gj_get(m, "dropout");
    if (gj_is(dr, GJ_NUM) && dr->u.num > 0) { snprintf(err, errlen, "BPE: dropout is not implemented"); return 0; }
    tk->model_kind = MODEL_BPE;
    tk->bpe_ignore_merges = jbool(m, "ignore_merges", 0);
    size_t maxid = 0;
    for (size_t i = 0; i < vocab->u.obj.n; ++i) {
        const gj_value* v = vocab->u.obj.vals[i];
        if (!gj_is(v, GJ_NUM) || v->u.num < 0) { snprintf(err, errlen, "BPE vocab entry %zu has no id", i); return 0; }
        if ((size_t)v->u.num > maxid) maxid = (size_t)v->u.num;
    }
    tk->nvocab = vocab->u.obj.n ? maxid + 1 : 0;
    tk->tok = (char**)calloc(tk->nvocab ? tk->nvocab : 1, sizeof(char*));
    tk->toklen = (uint32_t*)calloc(tk->nvocab ? tk->nvocab : 1, sizeof(uint32_t));
    tk->score = (double*)calloc(tk->nvocab ? tk->nvocab : 1, sizeof(double));
    if (!tk->tok || !tk->toklen || !tk->score) { snprintf(err, errlen, "out of memory"); return 0; }
    for (size_t i = 0; i < vocab->u.obj.n; ++i) {
        const size_t id = (size_t)vocab->u.obj.vals[i]->u.num;
        const char* k = vocab->u.obj.keys[i];
        if (tk->tok[id]) { snprintf(err, errlen, "BPE vocab: id %zu twice", id); return 0; }
        tk->tok[id] = dup_n(k, strlen(k));
        if (!tk->tok[id]) { snprintf(err, errlen, "out of memory"); return 0; }
        tk->toklen[id] = (uint32_t)strlen(k);
    }
    for (size_t i = 0; i < tk->nvocab; ++i) if (!tk->tok[i]) { tk->tok[i] = dup_n("", 0); if (!tk->tok[i]) { snprintf(err, errlen, "out of memory"); return 0; } }   /* holes in the id space */
    if (!build_trie(tk)) { snprintf(err, errlen, "out of memory"); return 0; }
    size_t ul; const char* unk = jstr(m, "unk_token", &ul);
    tk->unk_id = unk ? vocab_lookup(tk, unk, ul) : -1;
    tk->merge_cap = 16;
    while (tk->merge_cap < 2 * merges->u.arr.n + 2) tk->merge_cap *= 2;
    tk->merges = (bpe_merge*)calloc(tk->merge_cap, sizeof(bpe_merge));
    sbuf cat = {0, 0, 0};
    if (!tk->merges) { snprintf(err, errlen, "out of memory"); return 0; }
    for (size_t i = 0; i < merges->u.arr.n; ++i) {
        const gj_value* e = merges->u.arr.items[i];
        const char *a = NULL, *b = NULL; size_t al = 0, bl = 0;
        if (gj_is(e, GJ_STR)) {
            const char* sp = (const char*)memchr(e->u.str.s, ' ', e->u.str.len);
            if (sp) { a = e->u.str.s; al = (size_t)(sp - a); b = sp + 1; bl = e->u.str.len - al - 1; }
        } else if (gj_is(e, GJ_ARR) && e->u.arr.n == 2 && gj_is(e->u.arr.items[0], GJ_STR) && gj_is(e->u.arr.items[1], GJ_STR)) {
            a = e->u.arr.items[0]->u.str.s; al = e->u.arr.items[0]->u.str.len; b = e->u.arr.items[1]->u.str.s; bl = e->u.arr.items[1]->u.str.len;
        }
        if (!a || !al || !bl) { snprintf(err, errlen, "BPE merge %zu is malformed", i); sb_free(&cat); return 0; }
        const int32_t ia = vocab_lookup(tk, a, al), ib = vocab_lookup(tk, b, bl);
        sb_clear(&cat); sb_put(&cat, a, al); sb_put(&cat, b, bl);
        const int32_t im = vocab_lookup(tk, cat.s, cat.len);
        if (ia < 0 || ib < 0 || im < 0) { snprintf(err, errlen, "BPE merge %zu names a token outside the vocabulary", i); sb_free(&cat); return 0; }
        const uint64_t key = ((uint64_t)(uint32_t)ia << 32) | (uint32_t)ib | (1ull << 63);
        size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (tk->merge_cap - 1);
        while (tk->merges[h].key && tk->merges[h].key != key) h = (h + 1) & (tk->merge_cap - 1);
        if (!tk->merges[h].key) { tk->merges[h].key = key; tk->merges[h].rank = (int32_t)i; tk->merges[h].merged = im; }   /* first occurrence = lowest rank */
    }
    sb_free(&cat);
    for (int b = 0; b < 256; ++b) tk->byte_ids[b] = -1;
    tk->id_space = tk->nvocab;
    return 1;
}

static int load_model(glc_tokenizer* tk, const gj_value* m, char* err, size_t errlen) {
    const char* type = m ? jstr(m, "type", NULL) : NULL;
    if (m && type && !strcmp(type, "BPE")) return load_model_bpe(tk, m, err, errlen);
    const gj_value* vocab = gj_get(m, "vocab");
    if (!m || (type && strcmp(type, "Unigram")) || !gj_is(vocab, GJ_ARR)) {
        snprintf(err, errlen, "unsupported model '%s' (implemented: Unigram, byte-level BPE)", type ? type : "?");
        return 0;
    }
    const gj_value* unk = gj_get(m, "unk_id");
    tk->unk_id = gj_is(unk, GJ_NUM) ? (int32_t)unk->u.num : -1;
    tk->byte_fallback = jbool(m, "byte_fallback", 0);
    tk->nvocab = vocab->u.arr.n;
    tk->tok = (char**)calloc(tk->nvocab ? tk->nvocab : 1, sizeof(char*));
    tk->toklen = (uint32_t*)calloc(tk->nvocab ? tk->nvocab : 1, sizeof(uint32_t));
    tk->score = (double*)calloc(tk->nvocab ? tk->nvocab : 1, sizeof(double));
    if (!tk->tok || !tk->toklen || !tk->score) { snprintf(err, errlen, "out of memory"); return 0; }
    double min_score = 1e300;                                            /* f64::MAX in the Rust code; any real score is below */
    for (size_t i = 0; i < tk->nvocab; ++i) {
        const gj_value* e = vocab->u.arr.items[i];
        if (!gj_is(e, GJ_ARR) || e->u.arr.n != 2 || !gj_is(e->u.arr.items[0], GJ_STR) || !gj_is(e->u.arr.items[1], GJ_NUM)) {
            snprintf(err, errlen, "Unigram vocab entry %zu is not [piece, score]", i); return 0;
        }
        tk->tok[i] = dup_n(e->u.arr.items[0]->u.str.s, e->u.arr.items[0]->u.str.len);
        if (!tk->tok[i]) { snprintf(err, errlen, "out of memory"); return 0; }
        tk->toklen[i] = (uint32_t)e->u.arr.items[0]->u.str.len;
        tk->score[i] = e->u.arr.items[1]->u.num;
        if (tk->score[i] < min_score) min_score = tk->score[i];
    }
    if (tk->unk_id >= 0 && (size_t)tk->unk_id >= tk->nvocab) { snprintf(err, errlen, "unk_id outside the vocabulary"); return 0; }
    tk->unk_score = min_score - 10.0;                                    /* K_UNK_PENALTY */
    if (!build_trie(tk)) { snprintf(err, errlen, "out of memory"); return 0; }
    for (int b = 0; b < 256; ++b) { char name[8]; snprintf(name, sizeof name, "<0x%02X>", b); tk->byte_ids[b] = vocab_lookup(tk, name, 6); }
    tk->id_space = tk->nvocab;
    return 1;
}

static int32_t special_id(const glc_tokenizer* tk, const gj_value* pp_specials, const char* name) {
    const gj_value* e = gj_get(pp_specials, name);
    const gj_value* ids = gj_get(e, "ids");
    if (gj_is(ids, GJ_ARR) && ids->u.arr.n && gj_is(ids->u.arr.items[0], GJ_NUM)) return (int32_t)ids->u.arr.items[0]->u.num;
    return vocab_lookup(tk, name, strlen(name));
}

static int add_tmpl(glc_tokenizer* tk, int is_special, const int32_t* ids, size_t n) {
    tmpl_item* nt = (tmpl_item*)realloc(tk->tmpl, (tk->ntmpl + 1) * sizeof(tmpl_item));
    if (!nt) return 0;
    tk->tmpl = nt;
    tmpl_item* t = &nt[tk->ntmpl++];
    t->is_special = is_special; t->n = n; t->ids = NULL;
    if (n) { t->ids = (int32_t*)malloc(n * sizeof(int32_t)); if (!t->ids) return 0; memcpy(t->ids, ids, n * sizeof(int32_t)); }
    if (is_special) tk->tmpl_specials += n;
    return 1;
}

static int load_postprocessor(glc_tokenizer* tk, const gj_value* pp, char* err, size_t errlen) {
    if (!pp || pp->type == GJ_NULL) return 1;
    const char* type = jstr(pp, "type", NULL);
    if (!type) { snprintf(err, errlen, "post_processor without a type"); return 0; }
    if (!strcmp(type, "TemplateProcessing")) {
        const gj_value* single = gj_get(pp, "single");
        const gj_value* specials = gj_get(pp, "special_tokens");
        if (!gj_is(single, GJ_ARR)) { snprintf(err, errlen, "TemplateProcessing without 'single'"); return 0; }
        for (size_t i = 0; i < single->u.arr.n; ++i) {
            const gj_value* it = single->u.arr.items[i];
            const gj_value* sp = gj_get(it, "SpecialToken");
            const gj_value* sq = gj_get(it, "Sequence");
            if (sp) {
                const char* name = jstr(sp, "id", NULL);
                const gj_value* e = name ? gj_get(specials, name) : NULL;
                const gj_value* ids = gj_get(e, "ids");
                if (!gj_is(ids, GJ_ARR)) { snprintf(err, errlen, "TemplateProcessing: special token '%s' has no ids", name ? name : "?"); return 0; }
                int32_t tmp[16]; size_t n = ids->u.arr.n < 16 ? ids->u.arr.n : 16;
                for (size_t k = 0; k < n; ++k) tmp[k] = (int32_t)ids->u.arr.items[k]->u.num;
                if (!add_tmpl(tk, 1, tmp, n)) { snprintf(err, errlen, "out of memory"); return 0; }
            } else if (sq) {
                const char* which = jstr(sq, "id", NULL);
                if (which && !strcmp(which, "A")) { if (!add_tmpl(tk, 0, NULL, 0)) { snprintf(err, errlen, "out of memory"); return 0; } }
            } else { snprintf(err, errlen, "TemplateProcessing: unknown template item"); return 0; }
        }
        return 1;
    }
    if (!strcmp(type, "ByteLevel")) return 1;                            /* offsets only: adds no tokens */
    if (!strcmp(type, "Sequence")) {
        const gj_value* lst = gj_get(pp, "processors");
        if (!gj_is(lst, GJ_ARR)) { snprintf(err, errlen, "post_processor Sequence without a list"); return 0; }
        for (size_t i = 0; i < lst->u.arr.n; ++i) if (!load_postprocessor(tk, lst->u.arr.items[i], err, errlen)) return 0;
        return 1;
    }
    if (!strcmp(type, "BertProcessing") || !strcmp(type, "RobertaProcessing")) {
        const gj_value* cls = gj_get(pp, "cls"); const gj_value* sep = gj_get(pp, "sep");
        if (!gj_is(cls, GJ_ARR) || cls->u.arr.n != 2 || !gj_is(sep, GJ_ARR) || sep->u.arr.n != 2) { snprintf(err, errlen, "%s without cls/sep", type); return 0; }
        int32_t c = (int32_t)cls->u.arr.items[1]->u.num, s = (int32_t)sep->u.arr.items[1]->u.num;
        if (!add_tmpl(tk, 1, &c, 1) || !add_tmpl(tk, 0, NULL, 0) || !add_tmpl(tk, 1, &s, 1)) { snprintf(err, errlen, "out of memory"); return 0; }
        return 1;
    }
    (void)special_id;
    snprintf(err, errlen, "unsupported post_processor '%s'", type);
    return 0;
}

static int load_added(glc_tokenizer* tk, const gj_value* lst, char* err, size_t errlen) {
    if (!gj_is(lst, GJ_ARR) || !lst->u.arr.n) return 1;
    tk->added = (added_tok*)calloc(lst->u.arr.n, sizeof(added_tok));
    if (!tk->added) { snprintf(err, errlen, "out of memory"); return 0; }
    sbuf a = {0, 0, 0}, b = {0, 0, 0};
    for (size_t i = 0; i < lst->u.arr.n; ++i) {
        const gj_value* e = lst->u.arr.items[i];
        size_t l; const char* content = jstr(e, "content", &l);
        const gj_value* id = gj_get(e, "id");
        if (!content || !gj_is(id, GJ_NUM)) { snprintf(err, errlen, "added_tokens entry %zu is malformed", i); sb_free(&a); sb_free(&b); return 0; }
        added_tok* t = &tk->added[tk->nadded++];
        t->id = (int32_t)id->u.num;
        t->content = dup_n(content, l); t->len = l;
        t->single_word = jbool(e, "single_word", 0); t->lstrip = jbool(e, "lstrip", 0); t->rstrip = jbool(e, "rstrip", 0);
        t->normalized = jbool(e, "normalized", 1); t->special = jbool(e, "special", 0);
        if (t->normalized) { normalize_chain(tk, content, l, &a, &b); t->pattern = dup_n(a.s ? a.s : "", a.len); t->plen = a.len; }
        else { t->pattern = dup_n(content, l); t->plen = l; }
        if (!t->content || !t->pattern) { snprintf(err, errlen, "out of memory"); sb_free(&a); sb_free(&b); return 0; }
        if ((size_t)t->id + 1 > tk->id_space) tk->id_space = (size_t)t->id + 1;
    }
    sb_free(&a); sb_free(&b);
    return 1;
}

static void tokenizer_destroy(glc_tokenizer* tk) {
    if (!tk) return;
    for (size_t i = 0; i < tk->nvocab; ++i) free(tk->tok ? tk->tok[i] : NULL);
    free(tk->tok); free(tk->toklen); free(tk->score); free(tk->nodes); free(tk->edge_byte); free(tk->edge_node);
    for (size_t i = 0; i < tk->nsteps; ++i) { free(tk->steps[i].trie); free(tk->steps[i].blob); free(tk->steps[i].pat); free(tk->steps[i].rep); }
    free(tk->steps);
    for (size_t i = 0; i < tk->nadded; ++i) { free(tk->added[i].content); free(tk->added[i].pattern); }
    free(tk->added);
    for (size_t i = 0; i < tk->ntmpl; ++i) free(tk->tmpl[i].ids);
    free(tk->tmpl);
    free(tk->merges);
    sb_free(&tk->decoded);
    free(tk);
}

TokenizerHandle tokenizers_new_from_str(const char* json, size_t len) {
    char err[256] = "";
    gj_doc* doc = gj_parse(json, len, 0, err, sizeof err);
    if (!doc) { fprintf(stderr, "Error: tokenizer.json: %s\n", err); return NULL; }
    const gj_value* root = gj_root(doc);
    glc_tokenizer* tk = (glc_tokenizer*)calloc(1, sizeof(*tk));
    int ok = tk != NULL && gj_is(root, GJ_OBJ);
    if (ok) { tk->unk_id = -1; ok = load_model(tk, gj_get(root, "model"), err, sizeof err); }
    if (ok) ok = load_normalizer(tk, gj_get(root, "normalizer"), err, sizeof err);
    if (ok) ok = load_pretokenizer(tk, gj_get(root, "pre_tokenizer"), err, sizeof err);
    if (ok && (tk->pre_bytelevel || tk->bl_regex) && tk->model_kind != MODEL_BPE) { snprintf(err, sizeof err, "Split / ByteLevel pre_tokenizers are implemented for BPE models only"); ok = 0; }
    if (ok && tk->model_kind == MODEL_BPE && tk->has_metaspace) { snprintf(err, sizeof err, "a BPE model behind a Metaspace pre_tokenizer is not implemented"); ok = 0; }
    if (ok) ok = load_postprocessor(tk, gj_get(root, "post_processor"), err, sizeof err);
    if (ok) ok = load_added(tk, gj_get(root, "added_tokens"), err, sizeof err);
    if (ok) {
        const gj_value* tr = gj_get(root, "truncation");
        if (gj_is(tr, GJ_OBJ)) {
            const gj_value* ml = gj_get(tr, "max_length");
            const char* dir = jstr(tr, "direction", NULL);
            if (gj_is(ml, GJ_NUM) && ml->u.num > 0) tk->trunc_max = (size_t)ml->u.num;
            tk->trunc_left = dir && !strcmp(dir, "Left");
        }
        if (gj_is(gj_get(root, "padding"), GJ_OBJ))
            fprintf(stderr, "Note: tokenizer.json 'padding' section ignored (batches are padded by tokenize_inputs)\n");
        const gj_value* dec = gj_get(root, "decoder");
        const char* dt = dec ? jstr(dec, "type", NULL) : NULL;
        tk->dec_metaspace = dt && !strcmp(dt, "Metaspace");
        tk->dec_bytelevel = dt && !strcmp(dt, "ByteLevel");
        if (tk->dec_bytelevel && !tk->pre_bytelevel) bl_init_tables(tk);
    }
    gj_free(doc);
    if (!ok) {
        fprintf(stderr, "Error: tokenizer.json: %s\n", err[0] ? err : "not a tokenizer object");
        tokenizer_destroy(tk);
        return NULL;
    }
    return (TokenizerHandle)tk;
}

void tokenizers_free(TokenizerHandle handle) { tokenizer_destroy((glc_tokenizer*)handle); }

void tokenizers_encode(TokenizerHandle handle, const char* data, size_t len, int add_special_token, TokenizerEncodeResult* result) {
    result->token_ids = NULL; result->len = 0;
    if (!handle) return;
    encode_one((const glc_tokenizer*)handle, data, len, add_special_token, result);
}

void tokenizers_encode_batch(TokenizerHandle handle, const char** data, size_t* len, size_t num_seqs, int add_special_token,
                             TokenizerEncodeResult* results) {
    const glc_tokenizer* tk = (const glc_tokenizer*)handle;
#pragma omp parallel for schedule(dynamic, 1) if (num_seqs > 1) num_threads(glc_host_cpus())
    for (size_t i = 0; i < num_seqs; ++i) {
        results[i].token_ids = NULL; results[i].len = 0;
        if (tk) encode_one(tk, data[i], len[i], add_special_token, &results[i]);
    }
}

void tokenizers_free_encode_results(TokenizerEncodeResult* results, size_t num_seqs) {
    if (!results) return;
    for (size_t i = 0; i < num_seqs; ++i) { free(results[i].token_ids); results[i].token_ids = NULL; results[i].len = 0; }
    /* the array itself belongs to the caller (tokenizers-cpp only drops the id vectors) */
}

static const added_tok* added_by_id(const glc_tokenizer* tk, int32_t id) {
    for (size_t a = 0; a < tk->nadded; ++a) if (tk->added[a].id == id) return &tk->added[a];
    return NULL;
}

void tokenizers_get_vocab_size(TokenizerHandle handle, size_t* size) {
    const glc_tokenizer* tk = (const glc_tokenizer*)handle;
    *size = tk ? tk->id_space : 0;
}

void tokenizers_id_to_token(TokenizerHandle handle, uint32_t id, const char** data, size_t* len) {
    const glc_tokenizer* tk = (const glc_tokenizer*)handle;
    *data = ""; *len = 0;
    if (!tk) return;
    const added_tok* a = added_by_id(tk, (int32_t)id);
    if (a) { *data = a->content; *len = a->len; return; }
    if (id < tk->nvocab) { *data = tk->tok[id]; *len = tk->toklen[id]; }
}

void tokenizers_token_to_id(TokenizerHandle handle, const char* token, size_t len, int32_t* id) {
    const glc_tokenizer* tk = (const glc_tokenizer*)handle;
    *id = -1;
    if (!tk) return;
    for (size_t a = 0; a < tk->nadded; ++a)
        if (tk->added[a].len == len && !memcmp(tk->added[a].content, token, len)) { *id = tk->added[a].id; return; }
    *id = vocab_lookup(tk, token, len);
}

void tokenizers_decode(TokenizerHandle handle, const uint32_t* data, size_t len, int skip_special_token) {
    glc_tokenizer* tk = (glc_tokenizer*)handle;
    if (!tk) return;
    sbuf raw = {0, 0, 0};
    for (size_t i = 0; i < len; ++i) {
        const added_tok* a = added_by_id(tk, (int32_t)data[i]);
        if (a) { if (!(skip_special_token && a->special)) sb_put(&raw, a->content, a->len); continue; }
        if (data[i] < tk->nvocab) sb_put(&raw, tk->tok[data[i]], tk->toklen[data[i]]);
    }
    sb_clear(&tk->decoded); sb_put(&tk->decoded, "", 0);
    if (tk->dec_metaspace && tk->has_metaspace && raw.s) {
        size_t i = 0; int first = 1;
        while (i < raw.len) {
            if (i + tk->ms_replen <= raw.len && !memcmp(raw.s + i, tk->ms_rep, tk->ms_replen)) {
                if (!(first && tk->ms_scheme != 2)) sb_put(&tk->decoded, " ", 1);
                i += tk->ms_replen;
            } else { sb_put(&tk->decoded, raw.s + i, 1); ++i; }
            first = 0;
        }
    } else if (tk->dec_bytelevel && raw.s) {                             /* printable stand-ins back to bytes; anything else passes through */
        const unsigned char* u = (const unsigned char*)raw.s;
        for (size_t i = 0; i < raw.len;) {
            size_t adv; const uint32_t c = u8_decode(u + i, raw.len - i, &adv);
            if (c < 512 && tk->bl_byte_of[c] >= 0) { const char b = (char)tk->bl_byte_of[c]; sb_put(&tk->decoded, &b, 1); }
            else sb_put(&tk->decoded, raw.s + i, adv);
            i += adv;
        }
    } else if (raw.s) sb_put(&tk->decoded, raw.s, raw.len);
    sb_free(&raw);
}

void tokenizers_get_decode_str(TokenizerHandle handle, const char** data, size_t* len) {
    glc_tokenizer* tk = (glc_tokenizer*)handle;
    *data = tk && tk->decoded.s ? tk->decoded.s : ""; *len = tk ? tk->decoded.len : 0;
}

char* glc_tokenizer_normalize(TokenizerHandle handle, const char* data, size_t len, size_t* out_len) {
    const glc_tokenizer* tk = (const glc_tokenizer*)handle;
    sbuf clean = {0, 0, 0}, a = {0, 0, 0}, b = {0, 0, 0};
    if (!tk) return NULL;
    u8_sanitize(data, len, &clean);
    normalize_chain(tk, clean.s ? clean.s : "", clean.len, &a, &b);
    sb_free(&clean); sb_free(&b);
    if (!a.s) a.s = (char*)calloc(1, 1);
    if (out_len) *out_len = a.len;
    return a.s;
}

/* ---------------------------------------------------------------- the reference's src/tokenizer.c surface */

/* /root/reference/src/tokenizer.c:19-91: encode the batch with special tokens, cut every row at max_length (a raw cut:
 * the final [SEP] of an over-long row is dropped, :46-47), pad with id 0 / mask 0 to the longest remaining row. */
TokenizedInputs tokenize_inputs(TokenizerHandle tokenizer, const char* inputs[], size_t num_texts, size_t max_length) {
    TokenizerEncodeResult* results = (TokenizerEncodeResult*)malloc((num_texts ? num_texts : 1) * sizeof(TokenizerEncodeResult));
    size_t* input_lengths = (size_t*)malloc((num_texts ? num_texts : 1) * sizeof(size_t));
    if (!results || !input_lengths) {
        fprintf(stderr, "Error while allocating memmory for tokenization results\n");
        exit(1);                                                         /* as the reference does (:22-25) */
    }
    for (size_t i = 0; i < num_texts; ++i) input_lengths[i] = strlen(inputs[i]);
    tokenizers_encode_batch(tokenizer, inputs, input_lengths, num_texts, 1, results);

    size_t seq_length = 0;
    for (size_t i = 0; i < num_texts; ++i) {
        size_t l = results[i].len > max_length ? max_length : results[i].len;
        if (l > seq_length) seq_length = l;
    }
    TokenizedInputs tokenized;
    tokenized.input_ids = (int**)malloc((num_texts ? num_texts : 1) * sizeof(int*));
    tokenized.token_type_ids = (int**)malloc((num_texts ? num_texts : 1) * sizeof(int*));
    tokenized.attention_mask = (int**)malloc((num_texts ? num_texts : 1) * sizeof(int*));
    tokenized.batch_size = num_texts;
    tokenized.seq_length = seq_length;
    if (!tokenized.input_ids || !tokenized.token_type_ids || !tokenized.attention_mask) {
        fprintf(stderr, "Error while allocating memory for sequence lengths\n");
        exit(1);
    }
    for (size_t i = 0; i < num_texts; ++i) {
        tokenized.input_ids[i] = (int*)malloc((seq_length ? seq_length : 1) * sizeof(int));
        tokenized.token_type_ids[i] = (int*)malloc((seq_length ? seq_length : 1) * sizeof(int));
        tokenized.attention_mask[i] = (int*)malloc((seq_length ? seq_length : 1) * sizeof(int));
        if (!tokenized.input_ids[i] || !tokenized.token_type_ids[i] || !tokenized.attention_mask[i]) {
            fprintf(stderr, "Error while allocating memory for sequence lengths\n");
            exit(1);
        }
        for (size_t j = 0; j < seq_length; ++j) {
            const int real = j < results[i].len;                         /* j < seq_length <= max_length, so the reference's inner `break` (:71-74) is dead */
            tokenized.input_ids[i][j] = real ? results[i].token_ids[j] : 0;
            tokenized.token_type_ids[i][j] = 0;
            tokenized.attention_mask[i][j] = real ? 1 : 0;
        }
    }
    tokenizers_free_encode_results(results, num_texts);
    free(results);                                                       /* the reference leaks this array (:20, :86) */
    free(input_lengths);
    return tokenized;
}

/* /root/reference/src/tokenizer.c:98-120, same output bytes. */
void print_tokenized_inputs(const TokenizedInputs* tokenized) {
    for (size_t i = 0; i < tokenized->batch_size; ++i) {
        printf("Input %zu:\n", i);
        printf("input_ids: [");
        for (size_t j = 0; j < tokenized->seq_length; ++j) printf("%d, ", tokenized->input_ids[i][j]);
        printf("]\n");
        printf("token_type_ids: [");
        for (size_t j = 0; j < tokenized->seq_length; ++j) printf("%d, ", tokenized->token_type_ids[i][j]);
        printf("]\n");
        printf("attention_mask: [");
        for (size_t j = 0; j < tokenized->seq_length; ++j) printf("%d, ", tokenized->attention_mask[i][j]);
        printf("]\n");
    }
}

/* /root/reference/src/tokenizer.c:127-136. */
void free_tokenized_inputs(TokenizedInputs* tokenized) {
    for (size_t i = 0; i < tokenized->batch_size; ++i) {
        free(tokenized->input_ids[i]);
        free(tokenized->token_type_ids[i]);
        free(tokenized->attention_mask[i]);
    }
    free(tokenized->input_ids);
    free(tokenized->token_type_ids);
    free(tokenized->attention_mask);
    tokenized->input_ids = tokenized->token_type_ids = tokenized->attention_mask = NULL;
    tokenized->batch_size = 0;
}

/* /root/reference/src/tokenizer.c:145-184: read the JSON file, build the tokenizer; NULL + message on failure. */
TokenizerHandle create_tokenizer(const char* filepath) {
    FILE* file = fopen(filepath, "rb");
    if (!file) {
        fprintf(stderr, "Cant open file %s\n", filepath);
        return NULL;
    }
    fseek(file, 0, SEEK_END);
    long flen = ftell(file);
    fseek(file, 0, SEEK_SET);
    size_t json_len = flen > 0 ? (size_t)flen : 0;
    char* json = (char*)malloc(json_len + 1);
    if (!json) {
        fprintf(stderr, "Cant allocate memory for JSON\n");
        fclose(file);
        return NULL;
    }
    size_t read_len = fread(json, 1, json_len, file);
    fclose(file);
    if (read_len != json_len) {
        fprintf(stderr, "Failed to read %s\n", filepath);
        free(json);
        return NULL;
    }
    json[json_len] = '\0';
    TokenizerHandle handle = tokenizers_new_from_str(json, json_len);
    free(json);
    if (!handle) {
        fprintf(stderr, "Cant create tokenizer from %s\n", filepath);
        return NULL;
    }
    return handle;
}
