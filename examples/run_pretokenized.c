/*
 * Pure-C demonstration of the drop-in call sequence, mirroring /root/reference/main.c:83-187 with the
 * tokenizer stage replaced by pre-tokenized input (the Rust tokenizer is outside this repo's scope):
 *
 *   initialize_ort_api -> initialize_ort_environment -> create_ort_session
 *   -> prepare_input_tensors per batch of BATCH_SIZE rows      (src/parallel_processor.c:44)
 *   -> parallel_inference (the loop of main.c:141-150, spread over the session's GPUs)
 *   -> parallel_postprocess                                    (src/parallel_processor.c:70-90)
 *   -> g_ort->Release*                                         (main.c:173-187)
 *
 * usage: run_pretokenized <model.glcw | synthetic:cfg[:seed]> <tokens.txt> <classification_type> label1 [label2 ...]
 *   tokens.txt: one sequence per line, whitespace-separated token ids; rows of a batch are padded to the
 *   longest row with id 0 / mask 0 exactly like /root/reference/src/tokenizer.c:44-84.
 * build: gcc -std=c11 -Iinclude examples/run_pretokenized.c -Lgliclass/c_amd -lgliclass_model -lgliclass_hip \
 *            -Wl,-rpath,$PWD/gliclass/c_amd -fopenmp -o run_pretokenized
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "configs.h"
#include "model.h"
#include "parallel_processor.h"

const OrtApi* g_ort = NULL; /* defined by the caller, as in /root/reference/main.c:33 */

typedef struct { int* ids; size_t n; } row_t;

static int read_rows(const char* path, row_t** rows_out, size_t* n_out) {
    FILE* f = fopen(path, "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return -1; }
    row_t* rows = NULL;
    size_t n = 0, cap = 0;
    char* line = NULL;
    size_t lcap = 0;
    while (getline(&line, &lcap, f) > 0) {
        row_t r = {NULL, 0};
        size_t rcap = 0;
        for (char* tok = strtok(line, " \t\r\n"); tok; tok = strtok(NULL, " \t\r\n")) {
            if (r.n == rcap) { rcap = rcap ? rcap * 2 : 64; r.ids = (int*)realloc(r.ids, rcap * sizeof(int)); }
            r.ids[r.n++] = atoi(tok);
        }
        if (!r.n) { free(r.ids); continue; }
        if (n == cap) { cap = cap ? cap * 2 : 16; rows = (row_t*)realloc(rows, cap * sizeof(row_t)); }
        rows[n++] = r;
    }
    free(line);
    fclose(f);
    *rows_out = rows;
    *n_out = n;
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <model.glcw|synthetic:cfg[:seed]> <tokens.txt> <multi-label|single-label> label1 [label2 ...]\n", argv[0]);
        return 1;
    }
    row_t* rows = NULL;
    size_t num_texts = 0;
    if (read_rows(argv[2], &rows, &num_texts) != 0 || num_texts == 0) return 1;
    char* classification_type = argv[3];
    size_t num_labels_size = (size_t)argc - 4;
    char** label_set = &argv[4];
    char*** labels = &label_set;               /* same_labels = true: one shared label array in slot 0 */
    size_t num_labels[1] = {num_labels_size};
    char** texts = (char**)malloc(num_texts * sizeof(char*));
    for (size_t i = 0; i < num_texts; ++i) {
        texts[i] = (char*)malloc(32);
        snprintf(texts[i], 32, "row %zu (%zu tokens)", i, rows[i].n);
    }

    initialize_ort_api();
    OrtEnv* env = initialize_ort_environment();
    if (!env) return 1;
    OrtSession* session = create_ort_session(env, argv[1], NUM_THREADS);
    if (!session) { g_ort->ReleaseEnv(env); return 1; }

    const size_t num_batches = (num_texts + BATCH_SIZE - 1) / BATCH_SIZE;
    OrtValue** ids_t = (OrtValue**)calloc(num_batches, sizeof(OrtValue*));
    OrtValue** mask_t = (OrtValue**)calloc(num_batches, sizeof(OrtValue*));
    OrtValue** out_t = (OrtValue**)calloc(num_batches, sizeof(OrtValue*));
    for (size_t b = 0; b < num_batches; ++b) {             /* what tokenize_inputs would hand over, per batch */
        const size_t lo = b * BATCH_SIZE, n = (lo + BATCH_SIZE > num_texts) ? num_texts - lo : BATCH_SIZE;
        size_t S = 0;
        for (size_t i = 0; i < n; ++i) { size_t l = rows[lo + i].n > MAX_LENGTH ? MAX_LENGTH : rows[lo + i].n; if (l > S) S = l; }
        TokenizedInputs tok;
        tok.batch_size = n;
        tok.seq_length = S;
        tok.input_ids = (int**)malloc(n * sizeof(int*));
        tok.attention_mask = (int**)malloc(n * sizeof(int*));
        tok.token_type_ids = NULL;
        for (size_t i = 0; i < n; ++i) {
            tok.input_ids[i] = (int*)calloc(S, sizeof(int));
            tok.attention_mask[i] = (int*)calloc(S, sizeof(int));
            for (size_t j = 0; j < S && j < rows[lo + i].n; ++j) { tok.input_ids[i][j] = rows[lo + i].ids[j]; tok.attention_mask[i][j] = 1; }
        }
        if (prepare_input_tensors(&tok, &ids_t[b], &mask_t[b]) != 0) return 1;
        for (size_t i = 0; i < n; ++i) { free(tok.input_ids[i]); free(tok.attention_mask[i]); }
        free(tok.input_ids);
        free(tok.attention_mask);
    }

    parallel_inference(session, ids_t, mask_t, num_batches, out_t);
    parallel_postprocess(out_t, num_batches, num_texts, texts, labels, num_labels, true, num_labels_size, classification_type);

    for (size_t b = 0; b < num_batches; ++b) { g_ort->ReleaseValue(ids_t[b]); g_ort->ReleaseValue(mask_t[b]); }
    free(ids_t); free(mask_t); free(out_t);
    for (size_t i = 0; i < num_texts; ++i) { free(texts[i]); free(rows[i].ids); }
    free(texts); free(rows);
    g_ort->ReleaseSession(session);
    g_ort->ReleaseEnv(env);
    return 0;
}
