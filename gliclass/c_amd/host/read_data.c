/*
 * Input-file reader, same functions and field semantics as /root/reference/src/read_data.c:
 *   read_file (:14-28), parse_json (:45-145), string_to_bool (:159-168).
 * Input document: {"texts": [...], "labels": [[...], ...], "same_labels": bool, "classification_type": "multi-label" | "single-label"}.
 * Kept from the reference: `same_labels: true` reads only labels[0] and gives every text that count (:85-107); per-text
 * label arrays must match the number of texts (:113-117); outputs the document does not mention stay untouched; messages
 * (incl. their spelling) are the reference's.  Deliberate differences: a non-string entry becomes "" instead of an
 * uninitialised pointer, and a failed fread is reported.
 */
#include "read_data.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "glc_json.h"

char* read_file(const char* filename) {
    FILE* file = fopen(filename, "rb");
    if (!file) {
        fprintf(stderr, "Error: Faild to open file %s\n", filename);
        return NULL;
    }
    fseek(file, 0, SEEK_END);
    long length = ftell(file);
    fseek(file, 0, SEEK_SET);
    if (length < 0) length = 0;
    char* content = (char*)malloc((size_t)length + 1);
    if (!content) { fclose(file); fprintf(stderr, "Error: out of memory reading %s\n", filename); return NULL; }
    size_t got = fread(content, 1, (size_t)length, file);
    content[got] = '\0';
    fclose(file);
    if (got != (size_t)length) { fprintf(stderr, "Error: short read on %s\n", filename); free(content); return NULL; }
    return content;
}

static char* dup_str(const gj_value* v) {
    if (gj_is(v, GJ_STR)) {
        char* r = (char*)malloc(v->u.str.len + 1);
        if (r) memcpy(r, v->u.str.s, v->u.str.len + 1);
        return r;
    }
    return strdup("");
}

void parse_json(const char* json_string, char*** texts, size_t* num_texts, char**** labels, size_t** num_labels,
                size_t* num_labels_size, bool* same_labels, char** classification_type) {
    char err[160];
    gj_doc* doc = gj_parse(json_string, strlen(json_string), GJ_ALLOW_TRAILING, err, sizeof err);
    if (!doc) {
        fprintf(stderr, "Failed to parse JSON: %s\n", err);
        return;
    }
    const gj_value* json = gj_root(doc);

    const gj_value* texts_json = gj_get(json, "texts");
    if (gj_is(texts_json, GJ_ARR)) {
        *num_texts = texts_json->u.arr.n;
        *texts = (char**)malloc((*num_texts ? *num_texts : 1) * sizeof(char*));
        for (size_t i = 0; i < *num_texts; ++i) (*texts)[i] = dup_str(texts_json->u.arr.items[i]);
    }
    const gj_value* ct = gj_get(json, "classification_type");
    if (gj_is(ct, GJ_STR)) *classification_type = dup_str(ct);

    const gj_value* sl = gj_get(json, "same_labels");
    if (gj_is(sl, GJ_BOOL)) *same_labels = sl->u.boolean != 0;

    const gj_value* labels_json = gj_get(json, "labels");
    if (*same_labels) {
        if (gj_is(labels_json, GJ_ARR)) {
            *num_labels_size = labels_json->u.arr.n;
            if (*num_labels_size > 0) {
                const gj_value* first = labels_json->u.arr.items[0];
                if (gj_is(first, GJ_ARR)) {
                    *num_labels_size = first->u.arr.n;
                    *labels = (char***)malloc(sizeof(char**));                       /* one shared set */
                    (*labels)[0] = (char**)malloc((*num_labels_size ? *num_labels_size : 1) * sizeof(char*));
                    for (size_t i = 0; i < *num_labels_size; ++i) (*labels)[0][i] = dup_str(first->u.arr.items[i]);
                    *num_labels = (size_t*)calloc(*num_texts ? *num_texts : 1, sizeof(size_t));
                    (*num_labels)[0] = *num_labels_size;                             /* also when there are no texts: free_parsed_data reads it */
                    for (size_t i = 0; i < *num_texts; ++i) (*num_labels)[i] = *num_labels_size;
                }
            }
        }
    } else if (gj_is(labels_json, GJ_ARR)) {
        if (labels_json->u.arr.n != *num_texts) {
            fprintf(stderr, "Error:the number of arrays of labels does not match the number of texts.\n");
            gj_free(doc);
            return;
        }
        *num_labels = (size_t*)calloc(*num_texts ? *num_texts : 1, sizeof(size_t));
        *labels = (char***)calloc(*num_texts ? *num_texts : 1, sizeof(char**));
        for (size_t i = 0; i < *num_texts; ++i) {
            const gj_value* tl = labels_json->u.arr.items[i];
            if (gj_is(tl, GJ_ARR)) {
                size_t n = tl->u.arr.n;
                (*num_labels)[i] = n;
                (*labels)[i] = (char**)malloc((n ? n : 1) * sizeof(char*));
                if (!(*labels)[i]) {
                    fprintf(stderr, "Error: failed to allocate memory for text labels %zu.\n", i);
                    gj_free(doc);
                    return;
                }
                for (size_t j = 0; j < n; ++j) (*labels)[i][j] = dup_str(tl->u.arr.items[j]);
            } else {
                fprintf(stderr, "Error: labels forr text %zu are not array.\n", i);
            }
        }
    }
    gj_free(doc);
}

bool string_to_bool(const char* str) {
    if (strcmp(str, "true") == 0 || strcmp(str, "1") == 0) return true;
    if (strcmp(str, "false") == 0 || strcmp(str, "0") == 0) return false;
    printf("Invalid value for bool argument. Use 'true' or 'false'.\n");
    exit(1);                                                                         /* as the reference (:165-166) */
}

int glc_config_prompt_first(const char* model_path) {
    char path[4096];
    struct stat sb;
    if (!model_path || stat(model_path, &sb) != 0) { fprintf(stderr, "Error: cannot open model '%s'\n", model_path ? model_path : "(null)"); return -1; }
    if (S_ISDIR(sb.st_mode)) snprintf(path, sizeof path, "%s/config.json", model_path);
    else {
        snprintf(path, sizeof path, "%s", model_path);
        char* slash = strrchr(path, '/');
        snprintf(slash ? slash + 1 : path, sizeof path - (size_t)(slash ? slash + 1 - path : 0), "config.json");
    }
    char* text = read_file(path);
    if (!text) return -1;
    char err[160];
    gj_doc* doc = gj_parse(text, strlen(text), 0, err, sizeof err);
    free(text);
    if (!doc) { fprintf(stderr, "Error: %s: %s\n", path, err); return -1; }
    const gj_value* v = gj_get(gj_root(doc), "prompt_first");
    int r = gj_is(v, GJ_BOOL) ? (v->u.boolean ? 1 : 0) : -1;
    gj_free(doc);
    if (r < 0) fprintf(stderr, "Something wrong with model configuration file.\nExpected values: 'true' or 'false' for prompt_first in %s\n", path);
    return r;
}

void free_parsed_data(char** texts, size_t num_texts, char*** labels, size_t* num_labels, bool same_labels,
                      char* classification_type) {
    if (labels) {
        const size_t groups = same_labels ? 1 : num_texts;
        for (size_t g = 0; g < groups; ++g) {
            if (!labels[g]) continue;
            const size_t n = num_labels ? num_labels[same_labels ? 0 : g] : 0;
            for (size_t j = 0; j < n; ++j) free(labels[g][j]);
            free(labels[g]);
        }
        free(labels);
    }
    if (texts) { for (size_t i = 0; i < num_texts; ++i) free(texts[i]); free(texts); }
    free(num_labels);
    free(classification_type);
}
