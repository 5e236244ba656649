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

#include "glc_cpus.h"
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
#pragma omp parallel for schedule(dynamic) num_threads(glc_host_cpus())
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
#pragma omp parallel for schedule(dynamic) num_threads(glc_host_cpus())
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

/*
 * Extension (stage pipelining): the reference runs pre-processing, inference and post-processing as
 * three phases with a barrier after each (/root/reference/main.c:116, :141, :153).  Here every batch flows through the three
 * stages on its own: a team of host threads takes batches in order, each thread builds the prompts and tokenizes its batch,
 * hands it to run_inference (which deals concurrent calls round-robin to the session's GPUs and serialises per engine) and
 * then retires finished batches IN ORDER (a batch is printed as soon as every earlier batch has been printed), so tokenizing
 * batch i+k overlaps the GPU forward of batch i and the printing of batch i-1, and the output order is deterministic — the
 * reference's is not (unsynchronised printf from an OpenMP team, src/parallel_processor.c:73).  Per-batch results are the same
 * bytes as the three-phase path.  Returns the number of batches that failed.
 */
size_t parallel_classify(OrtSession* session, TokenizerHandle tokenizer_handler, char** texts, char*** labels, size_t* num_labels,
                         size_t num_texts, bool same_labels, size_t num_labels_size, bool prompt_first,
                         const char* classification_type) {
    const size_t bs = env_size("GLICLASS_BATCH_SIZE", BATCH_SIZE), max_len = env_size("GLICLASS_MAX_LENGTH", MAX_LENGTH);
    const char* ts = getenv("GLICLASS_THRESHOLD");
    const float threshold = (ts && *ts) ? strtof(ts, NULL) : THRESHOLD;
    const size_t nb = (num_texts + bs - 1) / bs;
    if (!nb) return 0;
    if (!session || !tokenizer_handler) { fprintf(stderr, "Error: parallel_classify: NULL session or tokenizer\n"); return nb; }
    OrtValue** outs = (OrtValue**)calloc(nb, sizeof(OrtValue*));
    unsigned char* state = (unsigned char*)calloc(nb, 1);            /* 0 pending, 1 done, 2 failed */
    if (!outs || !state) { free(outs); free(state); fprintf(stderr, "Error: parallel_classify: out of memory\n"); return nb; }
    size_t next_print = 0, failed = 0;
    int team = (int)env_size("GLICLASS_PIPELINE_THREADS", (size_t)glc_host_cpus());
    if ((size_t)team > nb) team = (int)nb;
    if (team < 1) team = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(team)
    for (size_t b = 0; b < nb; ++b) {
        const size_t i = b * bs, n = (i + bs > num_texts) ? (num_texts - i) : bs;
        const char** batch_texts = (const char**)&texts[i];
        const char*** batch_labels = (const char***)(same_labels ? (void*)labels : (void*)&labels[i]);
        size_t* batch_num_labels = same_labels ? num_labels : &num_labels[i];
        OrtValue *ids = NULL, *mask = NULL, *out = NULL;
        const char** prepared = prepare_inputs(batch_texts, batch_labels, n, batch_num_labels, same_labels, prompt_first);
        if (prepared) {
            TokenizedInputs tok = tokenize_inputs(tokenizer_handler, prepared, n, max_len);
            if (prepare_input_tensors(&tok, &ids, &mask) != 0) {
                fprintf(stderr, "Error: failed to prepare input tensors for batch %zu\n", b);
                ids = mask = NULL;
            }
            free_prepared_inputs((char**)prepared, n);
            free_tokenized_inputs(&tok);
        }
        if (ids && mask) out = run_inference(session, ids, mask);
        if (ids) g_ort->ReleaseValue(ids);
        if (mask) g_ort->ReleaseValue(mask);
#pragma omp critical(glc_pipeline_retire)
        {
            outs[b] = out;
            state[b] = out ? 1 : 2;
            while (next_print < nb && state[next_print]) {               /* retire the finished prefix, in order */
                const size_t k = next_print, lo = k * bs, cnt = (lo + bs > num_texts) ? (num_texts - lo) : bs;
                if (state[k] == 1) {
                    process_output_tensor(outs[k], g_ort, same_labels, (const char** const*)(same_labels ? (void*)labels : (void*)&labels[lo]),
                                          same_labels ? num_labels : &num_labels[lo], num_labels_size, threshold, cnt,
                                          (const char**)&texts[lo], classification_type);
                    g_ort->ReleaseValue(outs[k]);
                    outs[k] = NULL;
                } else ++failed;
                ++next_print;
            }
        }
    }
    free(outs);
    free(state);
    return failed;
}
