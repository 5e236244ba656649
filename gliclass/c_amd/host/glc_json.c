/* See glc_json.h.  Recursive-descent parser, arena-allocated DOM. */
#include "glc_json.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GJ_CHUNK (1u << 20)
#define GJ_MAX_DEPTH 512

typedef struct gj_chunk { struct gj_chunk* next; size_t used, cap; } gj_chunk;

struct gj_doc { gj_chunk* chunks; gj_value* root; };

typedef struct {
    const char* p; const char* end; const char* begin;
    gj_doc* doc; char* err; size_t errlen; int failed;
    gj_value** stack; size_t sp, scap;        /* scratch for array items / object values */
    const char** kstack; size_t ksp, kcap;    /* scratch for object keys */
} gj_parser;

static void* arena_alloc(gj_doc* d, size_t n) {
    n = (n + 15) & ~(size_t)15;
    gj_chunk* c = d->chunks;
    if (!c || c->used + n > c->cap) {
        size_t cap = n > GJ_CHUNK ? n : GJ_CHUNK;
        c = (gj_chunk*)malloc(sizeof(gj_chunk) + 16 + cap);
        if (!c) return NULL;
        c->next = d->chunks; c->used = 0; c->cap = cap; d->chunks = c;
    }
    char* base = (char*)(((uintptr_t)(c + 1) + 15) & ~(uintptr_t)15);
    void* r = base + c->used;
    c->used += n;
    return r;
}

static void fail(gj_parser* ps, const char* why) {
    if (!ps->failed && ps->err && ps->errlen)
        snprintf(ps->err, ps->errlen, "JSON error at byte %zu: %s", (size_t)(ps->p - ps->begin), why);
    ps->failed = 1;
}

static void skip_ws(gj_parser* ps) {
    while (ps->p < ps->end && (*ps->p == ' ' || *ps->p == '\t' || *ps->p == '\n' || *ps->p == '\r')) ++ps->p;
}

static int hex4(const char* p, unsigned* out) {
    unsigned v = 0;
    for (int i = 0; i < 4; ++i) {
        char c = p[i]; v <<= 4;
        if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
        else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
        else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
        else return 0;
    }
    *out = v; return 1;
}

static size_t put_utf8(char* o, unsigned cp) {
    if (cp < 0x80) { o[0] = (char)cp; return 1; }
    if (cp < 0x800) { o[0] = (char)(0xC0 | (cp >> 6)); o[1] = (char)(0x80 | (cp & 0x3F)); return 2; }
    if (cp < 0x10000) { o[0] = (char)(0xE0 | (cp >> 12)); o[1] = (char)(0x80 | ((cp >> 6) & 0x3F)); o[2] = (char)(0x80 | (cp & 0x3F)); return 3; }
    o[0] = (char)(0xF0 | (cp >> 18)); o[1] = (char)(0x80 | ((cp >> 12) & 0x3F)); o[2] = (char)(0x80 | ((cp >> 6) & 0x3F)); o[3] = (char)(0x80 | (cp & 0x3F));
    return 4;
}

/* p points at the opening quote; returns an arena copy, decoded. */
static const char* parse_string_raw(gj_parser* ps, size_t* out_len) {
    const char* q = ps->p + 1;
    while (q < ps->end && *q != '"') { if (*q == '\\') ++q; ++q; }
    if (q >= ps->end) { fail(ps, "unterminated string"); return NULL; }
    char* buf = (char*)arena_alloc(ps->doc, (size_t)(q - ps->p) + 1);          /* decoded length <= raw length */
    if (!buf) { fail(ps, "out of memory"); return NULL; }
    size_t n = 0;
    const char* s = ps->p + 1;
    while (s < q) {
        unsigned char c = (unsigned char)*s;
        if (c != '\\') { if (c < 0x20) { ps->p = s; fail(ps, "control character in string"); return NULL; } buf[n++] = (char)c; ++s; continue; }
        ++s;
        switch (*s) {
            case '"': buf[n++] = '"'; ++s; break;
            case '\\': buf[n++] = '\\'; ++s; break;
            case '/': buf[n++] = '/'; ++s; break;
            case 'b': buf[n++] = '\b'; ++s; break;
            case 'f': buf[n++] = '\f'; ++s; break;
            case 'n': buf[n++] = '\n'; ++s; break;
            case 'r': buf[n++] = '\r'; ++s; break;
            case 't': buf[n++] = '\t'; ++s; break;
            case 'u': {
                unsigned cp, lo;
                if (s + 5 > q || !hex4(s + 1, &cp)) { ps->p = s; fail(ps, "bad \\u escape"); return NULL; }
                s += 5;
                if (cp >= 0xD800 && cp < 0xDC00) {
                    if (s + 6 <= q && s[0] == '\\' && s[1] == 'u' && hex4(s + 2, &lo) && lo >= 0xDC00 && lo < 0xE000) {
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); s += 6;
                    } else cp = 0xFFFD;
                } else if (cp >= 0xDC00 && cp < 0xE000) cp = 0xFFFD;
                n += put_utf8(buf + n, cp);
                break;
            }
            default: ps->p = s; fail(ps, "bad escape"); return NULL;
        }
    }
    buf[n] = 0;
    *out_len = n;
    ps->p = q + 1;
    return buf;
}

static gj_value* new_value(gj_parser* ps, gj_type t) {
    gj_value* v = (gj_value*)arena_alloc(ps->doc, sizeof(gj_value));
    if (!v) { fail(ps, "out of memory"); return NULL; }
    memset(v, 0, sizeof(*v)); v->type = t;
    return v;
}

static int push_val(gj_parser* ps, gj_value* v) {
    if (ps->sp == ps->scap) {
        size_t nc = ps->scap ? ps->scap * 2 : 1024;
        gj_value** ns = (gj_value**)realloc(ps->stack, nc * sizeof(*ns));
        if (!ns) { fail(ps, "out of memory"); return 0; }
        ps->stack = ns; ps->scap = nc;
    }
    ps->stack[ps->sp++] = v; return 1;
}
static int push_key(gj_parser* ps, const char* k) {
    if (ps->ksp == ps->kcap) {
        size_t nc = ps->kcap ? ps->kcap * 2 : 256;
        const char** ns = (const char**)realloc(ps->kstack, nc * sizeof(*ns));
        if (!ns) { fail(ps, "out of memory"); return 0; }
        ps->kstack = ns; ps->kcap = nc;
    }
    ps->kstack[ps->ksp++] = k; return 1;
}

static gj_value* parse_value(gj_parser* ps, int depth);

static gj_value* parse_array(gj_parser* ps, int depth) {
    gj_value* v = new_value(ps, GJ_ARR);
    if (!v) return NULL;
    size_t base = ps->sp;
    ++ps->p; skip_ws(ps);
    if (ps->p < ps->end && *ps->p == ']') { ++ps->p; return v; }
    for (;;) {
        gj_value* e = parse_value(ps, depth + 1);
        if (!e || !push_val(ps, e)) return NULL;
        skip_ws(ps);
        if (ps->p >= ps->end) { fail(ps, "unterminated array"); return NULL; }
        if (*ps->p == ',') { ++ps->p; continue; }
        if (*ps->p == ']') { ++ps->p; break; }
        fail(ps, "expected ',' or ']'"); return NULL;
    }
    size_t n = ps->sp - base;
    v->u.arr.items = (gj_value**)arena_alloc(ps->doc, n * sizeof(gj_value*));
    if (!v->u.arr.items) { fail(ps, "out of memory"); return NULL; }
    memcpy(v->u.arr.items, ps->stack + base, n * sizeof(gj_value*));
    v->u.arr.n = n; ps->sp = base;
    return v;
}

static gj_value* parse_object(gj_parser* ps, int depth) {
    gj_value* v = new_value(ps, GJ_OBJ);
    if (!v) return NULL;
    size_t base = ps->sp, kbase = ps->ksp;
    ++ps->p; skip_ws(ps);
    if (ps->p < ps->end && *ps->p == '}') { ++ps->p; return v; }
    for (;;) {
        skip_ws(ps);
        if (ps->p >= ps->end || *ps->p != '"') { fail(ps, "expected object key"); return NULL; }
        size_t klen; const char* k = parse_string_raw(ps, &klen);
        if (!k || !push_key(ps, k)) return NULL;
        skip_ws(ps);
        if (ps->p >= ps->end || *ps->p != ':') { fail(ps, "expected ':'"); return NULL; }
        ++ps->p;
        gj_value* e = parse_value(ps, depth + 1);
        if (!e || !push_val(ps, e)) return NULL;
        skip_ws(ps);
        if (ps->p >= ps->end) { fail(ps, "unterminated object"); return NULL; }
        if (*ps->p == ',') { ++ps->p; continue; }
        if (*ps->p == '}') { ++ps->p; break; }
        fail(ps, "expected ',' or '}'"); return NULL;
    }
    size_t n = ps->sp - base;
    v->u.obj.vals = (gj_value**)arena_alloc(ps->doc, n * sizeof(gj_value*));
    v->u.obj.keys = (const char**)arena_alloc(ps->doc, n * sizeof(char*));
    if (!v->u.obj.vals || !v->u.obj.keys) { fail(ps, "out of memory"); return NULL; }
    memcpy(v->u.obj.vals, ps->stack + base, n * sizeof(gj_value*));
    memcpy(v->u.obj.keys, ps->kstack + kbase, n * sizeof(char*));
    v->u.obj.n = n; ps->sp = base; ps->ksp = kbase;
    return v;
}

static gj_value* parse_value(gj_parser* ps, int depth) {
    if (ps->failed) return NULL;
    if (depth > GJ_MAX_DEPTH) { fail(ps, "nesting too deep"); return NULL; }
    skip_ws(ps);
    if (ps->p >= ps->end) { fail(ps, "unexpected end of input"); return NULL; }
    char c = *ps->p;
    if (c == '{') return parse_object(ps, depth);
    if (c == '[') return parse_array(ps, depth);
    if (c == '"') {
        gj_value* v = new_value(ps, GJ_STR);
        if (!v) return NULL;
        v->u.str.s = parse_string_raw(ps, &v->u.str.len);
        return v->u.str.s ? v : NULL;
    }
    size_t left = (size_t)(ps->end - ps->p);
    if (left >= 4 && !memcmp(ps->p, "true", 4)) { gj_value* v = new_value(ps, GJ_BOOL); if (v) v->u.boolean = 1; ps->p += 4; return v; }
    if (left >= 5 && !memcmp(ps->p, "false", 5)) { gj_value* v = new_value(ps, GJ_BOOL); ps->p += 5; return v; }
    if (left >= 4 && !memcmp(ps->p, "null", 4)) { gj_value* v = new_value(ps, GJ_NULL); ps->p += 4; return v; }
    if (c == '-' || (c >= '0' && c <= '9')) {
        /* delimit the number first: the buffer need not be NUL-terminated */
        const char* q = ps->p;
        if (*q == '-') ++q;
        while (q < ps->end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) ++q;
        char tmp[64]; size_t n = (size_t)(q - ps->p);
        if (n == 0 || n >= sizeof(tmp)) { fail(ps, "bad number"); return NULL; }
        memcpy(tmp, ps->p, n); tmp[n] = 0;
        char* endp = NULL;
        double d = strtod(tmp, &endp);
        if (endp != tmp + n) { fail(ps, "bad number"); return NULL; }
        gj_value* v = new_value(ps, GJ_NUM);
        if (v) v->u.num = d;
        ps->p = q;
        return v;
    }
    fail(ps, "unexpected character");
    return NULL;
}

gj_doc* gj_parse(const char* text, size_t len, int flags, char* err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    gj_doc* d = (gj_doc*)calloc(1, sizeof(*d));
    if (!d) return NULL;
    gj_parser ps; memset(&ps, 0, sizeof(ps));
    ps.p = ps.begin = text; ps.end = text + len; ps.doc = d; ps.err = err; ps.errlen = errlen;
    if (len >= 3 && !memcmp(text, "\xEF\xBB\xBF", 3)) ps.p += 3;
    d->root = parse_value(&ps, 0);
    if (d->root && !(flags & GJ_ALLOW_TRAILING)) {
        skip_ws(&ps);
        if (ps.p != ps.end) { fail(&ps, "trailing characters"); d->root = NULL; }
    }
    free(ps.stack); free(ps.kstack);
    if (!d->root || ps.failed) { gj_free(d); return NULL; }
    return d;
}

const gj_value* gj_root(const gj_doc* doc) { return doc ? doc->root : NULL; }

void gj_free(gj_doc* d) {
    if (!d) return;
    gj_chunk* c = d->chunks;
    while (c) { gj_chunk* n = c->next; free(c); c = n; }
    free(d);
}

const gj_value* gj_get(const gj_value* obj, const char* key) {
    if (!obj || obj->type != GJ_OBJ) return NULL;
    for (size_t i = 0; i < obj->u.obj.n; ++i)
        if (!strcmp(obj->u.obj.keys[i], key)) return obj->u.obj.vals[i];
    return NULL;
}
