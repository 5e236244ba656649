/*
 * Batch sharding, same interface and chunking rules as /root/reference/src/parallel_processor.c
 * (chunks of BATCH_SIZE texts :28-30, chunk i <-> tensor slot i/BATCH_SIZE :44-45, short last chunk,
 * same_labels selects shared vs per-text label arrays :34-35/:79-81, outputs released after
 * post-processing :88).  Pre-processing = prompt builder (host/preprocessor.c) + tokenizer (host/tokenizer.c, the native
 * Unigram implementation of include/tokenizers_c.h) + prepare_input_tensors.  Both are ordinary exported symbols of this
 * library, so an integrator who links the reference's own src/preprocessor.c / src/tokenizer.c (+ tokenizers-cpp) into the
 * executable overrides them.
 */
#include "parallel_processor.h"

#include <omp.h>
#include <stdlib.h>

#include "model.h"
#include "postprocessor.h"
#include "preprocessor.h"


static size_t env_size(const char* name, size_t dflt) {
    const char* s = getenv(name);
    if (!s || !*s) return dflt;
    long v = strtol(s, NULL, 10);
    return v > 0 ? (size_t)v : dflt;
}

void parallel_preprocess(char** texts, char*** labels, size_t* num_labels, size_t num_texts, bool same_labels,
                         bool prompt_first, TokenizerHandle tokenizer_handler, OrtValue** input_ids_tensors,
                         OrtValue** attention_mask_tensors) {
    const size_t bs = env_size("GLICLASS_BATCH_SIZE", BATCH_SIZE), max_len = env_size("GLICLASS_MAX_LENGTH", MAX_LENGTH);
    const size_t nb = (num_texts + bs - 1) / bs;
    for (size_t i = 0; i < nb; ++i) { input_ids_tensors[i] = NULL; attention_mask_tensors[i] = NULL; }
    if (!tokenizer_handler) {
        fprintf(stderr, "Error: parallel_preprocess: NULL tokenizer handle\n");
        return;
    }
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < num_texts; i += bs) {
        const size_t n = (i + bs > num_texts) ? (num_texts - i) : bs;
        const char** batch_texts = (const char**)&texts[i];
        const char*** batch_labels = (const char***)(same_labels ? (void*)labels : (void*)&labels[i]);
        size_t* batch_num_labels = same_labels ? num_labels : &num_labels[i];
        const char** prepared = prepare_inputs(batch_texts, batch_labels, n, batch_num_labels, same_labels, prompt_first);
        if (!prepared) continue;
        TokenizedInputs tok = tokenize_inputs(tokenizer_handler, prepared, n, max_len);
        if (prepare_input_tensors(&tok, &input_ids_tensors[i / bs], &attention_mask_tensors[i / bs]) != 0)
            fprintf(stderr, "Error: failed to prepare input tensors for batch %zu\n", i / bs);
        free_prepared_inputs((char**)prepared, n);
        free_tokenized_inputs(&tok);
    }
}

void parallel_postprocess(OrtValue** output_tensors, size_t num_batches, size_t num_texts, char** texts, char*** labels,
                          size_t* num_labels, bool same_labels, size_t num_labels_size, const char* classification_type) {
    const size_t bs = env_size("GLICLASS_BATCH_SIZE", BATCH_SIZE);
    const char* ts = getenv("GLICLASS_THRESHOLD");
    const float threshold = (ts && *ts) ? strtof(ts, NULL) : THRESHOLD;
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < num_batches; ++i) {
        const size_t n = (i == num_batches - 1) ? (num_texts - i * bs) : bs;
        const char** batch_texts = (const char**)&texts[i * bs];
        const char*** batch_labels = (const char***)(same_labels ? (void*)labels : (void*)&labels[i * bs]);
        size_t* batch_num_labels = same_labels ? num_labels : &num_labels[i * bs];
        if (output_tensors[i]) {
            process_output_tensor(output_tensors[i], g_ort, same_labels, (const char** const*)batch_labels, batch_num_labels,
                                  num_labels_size, threshold, n, batch_texts, classification_type);
            g_ort->ReleaseValue(output_tensors[i]);
            output_tensors[i] = NULL;
        }
    }
}
