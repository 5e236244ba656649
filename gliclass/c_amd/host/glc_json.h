/*
 * Minimal JSON reader of the host layer (stands where the reference links cJSON: /root/reference/src/read_data.c:5,48-144,
 * and where tokenizers-cpp parses tokenizer.json: /root/reference/src/tokenizer.c:175).  DOM in one arena; strings are
 * decoded to UTF-8 (\uXXXX incl. surrogate pairs) and carry an explicit length (they may contain NUL).
 */
#ifndef GLC_JSON_H
#define GLC_JSON_H
#include <stddef.h>

typedef enum { GJ_NULL = 0, GJ_BOOL, GJ_NUM, GJ_STR, GJ_ARR, GJ_OBJ } gj_type;

typedef struct gj_value gj_value;
struct gj_value {
    gj_type type;
    union {
        int boolean;
        double num;
        struct { const char* s; size_t len; } str;                                   /* NUL-terminated copy, len excludes it */
        struct { gj_value** items; size_t n; } arr;
        struct { const char** keys; gj_value** vals; size_t n; } obj;                /* insertion order, duplicates kept */
    } u;
};

typedef struct gj_doc gj_doc;

/* Parses `len` bytes of JSON text.  Returns NULL and fills `err` (position + reason) on malformed input.
 * GJ_ALLOW_TRAILING: stop after the first value like cJSON_Parse does (the reference's parse_json relies on it). */
#define GJ_ALLOW_TRAILING 1
gj_doc* gj_parse(const char* text, size_t len, int flags, char* err, size_t errlen);
const gj_value* gj_root(const gj_doc* doc);
void gj_free(gj_doc* doc);

/* Object member by exact (case-sensitive) key; first match; NULL if `obj` is not an object or the key is absent. */
const gj_value* gj_get(const gj_value* obj, const char* key);
static inline int gj_is(const gj_value* v, gj_type t) { return v && v->type == t; }

#endif
