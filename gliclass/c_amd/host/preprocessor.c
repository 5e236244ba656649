/*
 * Prompt builder — same interface and observable strings as /root/reference/src/preprocessor.c:23-124:
 * every label becomes "<<LABEL>>" + bytewise-lowercased label (:86-92, :98-105), the label block is
 * closed by "<<SEP>>", and it goes before the text when prompt_first, after it otherwise (:84-108).
 * One exact-size allocation and memcpy per piece (the reference grows the string with strcat/strncat).
 * Symbols are weak so an integrator who keeps the reference's own src/preprocessor.c wins the link.
 */
#include "preprocessor.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GLC_LABEL_TAG "<<LABEL>>"
#define GLC_SEP_TAG "<<SEP>>"

static char* append_labels(char* dst, const char* labels[], size_t num_labels) {
    for (size_t i = 0; i < num_labels; ++i) {
        memcpy(dst, GLC_LABEL_TAG, sizeof(GLC_LABEL_TAG) - 1);
        dst += sizeof(GLC_LABEL_TAG) - 1;
        for (const unsigned char* p = (const unsigned char*)labels[i]; *p; ++p) *dst++ = (char)tolower(*p);
    }
    memcpy(dst, GLC_SEP_TAG, sizeof(GLC_SEP_TAG) - 1);
    return dst + sizeof(GLC_SEP_TAG) - 1;
}

__attribute__((weak)) char* prepare_input(const char* text, const char* labels[], size_t num_labels, bool prompt_first) {
    if (!text || (num_labels && !labels)) return NULL;
    size_t n = strlen(text) + sizeof(GLC_SEP_TAG) - 1;
    for (size_t i = 0; i < num_labels; ++i) {
        if (!labels[i]) return NULL;
        n += sizeof(GLC_LABEL_TAG) - 1 + strlen(labels[i]);
    }
    char* out = (char*)malloc(n + 1);
    if (!out) { fprintf(stderr, "Cant allocate memmory for result prepared string\n"); return NULL; }
    char* p = out;
    const size_t tl = strlen(text);
    if (prompt_first) {
        p = append_labels(p, labels, num_labels);
        memcpy(p, text, tl);
        p += tl;
    } else {
        memcpy(p, text, tl);
        p = append_labels(p + tl, labels, num_labels);
    }
    *p = '\0';
    return out;
}

__attribute__((weak)) const char** prepare_inputs(const char* texts[], const char** const* labels, size_t num_texts,
                                                  size_t num_labels[], bool same_labels, bool prompt_first) {
    char** inputs = (char**)malloc((num_texts ? num_texts : 1) * sizeof(char*));
    if (!inputs) { fprintf(stderr, "Error: cant allocate memory for array inputs\n"); return NULL; }
    for (size_t i = 0; i < num_texts; ++i) {
        const size_t li = same_labels ? 0 : i;     /* shared label set lives in slot 0 (:34-38) */
        inputs[i] = prepare_input(texts[i], (const char**)labels[li], num_labels[li], prompt_first);
        if (!inputs[i]) {
            fprintf(stderr, "Error while preparing text for text: %zu\n", i);
            for (size_t j = 0; j < i; ++j) free(inputs[j]);
            free(inputs);
            return NULL;
        }
    }
    return (const char**)inputs;
}

__attribute__((weak)) void free_prepared_inputs(char** prepared_inputs, size_t num_texts) {
    if (!prepared_inputs) return;
    for (size_t i = 0; i < num_texts; ++i) free(prepared_inputs[i]);
    free(prepared_inputs);
}
