/*
 * Logits -> labels, same observable behaviour as /root/reference/src/postprocessor.c:14-156
 * (strict `>` threshold :95, single-label = argmax of sigmoid with max_prob starting at 0 :119-128,
 * label lookup rules :97-105/:131-139, output format :90,:108,:142,:149).  Each call formats into one
 * buffer and writes it with a single fwrite so batches printed from concurrent threads do not
 * interleave (the reference's printf calls do).
 */
#include "postprocessor.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

float sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

typedef struct { char* p; size_t len, cap; } sbuf;
static void sb_printf(sbuf* b, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(NULL, 0, fmt, ap);
    va_end(ap);
    if (n < 0) return;
    if (b->len + (size_t)n + 1 > b->cap) {
        size_t nc = (b->cap ? b->cap * 2 : 1024);
        while (nc < b->len + (size_t)n + 1) nc *= 2;
        char* np = (char*)realloc(b->p, nc);
        if (!np) return;
        b->p = np;
        b->cap = nc;
    }
    va_start(ap, fmt);
    vsnprintf(b->p + b->len, b->cap - b->len, fmt, ap);
    va_end(ap);
    b->len += (size_t)n;
}

void process_output_tensor(OrtValue* output_tensor, const OrtApi* api, bool same_labels, const char** const* labels,
                           const size_t* num_labels, size_t num_labels_size, float threshold, size_t num_texts,
                           const char** texts, const char* classification_type) {
    if (!output_tensor || !api) { fprintf(stderr, "Error: Unable to obtain information about the tensor type and shape.\n"); return; }
    OrtTensorTypeAndShapeInfo* info = NULL;
    OrtStatus* st = api->GetTensorTypeAndShape(output_tensor, &info);
    if (st) { fprintf(stderr, "Error: Unable to obtain information about the tensor type and shape.\n"); api->ReleaseStatus(st); return; }
    size_t nd = 0;
    st = api->GetDimensionsCount(info, &nd);
    if (st || nd != 2) {
        fprintf(stderr, "Error: Failed to get the number of dimensions of the tensor.\n");
        api->ReleaseTensorTypeAndShapeInfo(info);
        if (st) api->ReleaseStatus(st);
        return;
    }
    int64_t dims[2] = {0, 0};
    st = api->GetDimensions(info, dims, 2);
    if (st) { fprintf(stderr, "Error: Failed to get tensor dimension sizes.\n"); api->ReleaseTensorTypeAndShapeInfo(info); api->ReleaseStatus(st); return; }
    float* data = NULL;
    st = api->GetTensorMutableData(output_tensor, (void**)&data);
    if (st) { fprintf(stderr, "Error: Failed to get tensor data.\n"); api->ReleaseTensorTypeAndShapeInfo(info); api->ReleaseStatus(st); return; }

    const int batch = (int)dims[0], classes = (int)dims[1];
    sbuf out = {0};
    if (strcmp(classification_type, "multi-label") == 0) {
        for (int i = 0; i < batch; ++i) {
            sb_printf(&out, "Text_%d: %s:\n", i, texts[i]);
            for (int j = 0; j < classes; ++j) {
                const float prob = sigmoid(data[(size_t)i * classes + j]);
                if (!(prob > threshold)) continue;
                const char* label = NULL;
                if (same_labels) { if ((size_t)j < num_labels_size) label = labels[0][j]; }
                else if ((size_t)i < num_texts && (size_t)j < num_labels[i]) label = labels[i][j];
                if (label) sb_printf(&out, "  Text_%d Label: %s, Score: %.6f\n", i, label, prob);
                else sb_printf(&out, "  Text_%d Label: [Unknown], Score: %.6f\n", i, prob);
            }
            sb_printf(&out, "\n");
        }
    } else if (strcmp(classification_type, "single-label") == 0) {
        for (int i = 0; i < batch; ++i) {
            sb_printf(&out, "Text_%d: %s:\n", i, texts[i]);
            float best = 0.0f;
            int arg = -1;
            for (int j = 0; j < classes; ++j) {
                const float prob = sigmoid(data[(size_t)i * classes + j]);
                if (prob > best) { best = prob; arg = j; }
            }
            const char* label = NULL;
            if (arg >= 0) {
                if (same_labels) { if ((size_t)arg < num_labels_size) label = labels[0][arg]; }
                else if ((size_t)i < num_texts && (size_t)arg < num_labels[i]) label = labels[i][arg];
            }
            if (label) sb_printf(&out, "  Text_%d Label: %s, Score: %.6f\n", i, label, best);
            else sb_printf(&out, "  Text_%d Label: [Unknown], Score: %.6f\n", i, best);
            sb_printf(&out, "\n");
        }
    } else {
        sb_printf(&out, "This type of classification is not supported\n");
    }
    if (out.p) { fwrite(out.p, 1, out.len, stdout); free(out.p); }
    api->ReleaseTensorTypeAndShapeInfo(info);
}
