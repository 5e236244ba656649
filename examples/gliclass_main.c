/*
 * The launcher: same stages, same stdout lines and same exit codes as /root/reference/main.c:53-190, on the native
 * host layer (no ONNXRuntime, no Rust tokenizer, no cJSON):
 *
 *   read_file + parse_json            (main.c:63-75)     include/read_data.h
 *   create_tokenizer                  (main.c:77-81)     include/tokenizer.h -> native Unigram tokenizer
 *   initialize_ort_api / environment / create_ort_session (main.c:83-99)  -> one gfx950 engine per GPU of the session
 *   parallel_preprocess               (main.c:116-118)   prompt builder + tokenizer + tensors, OpenMP over batches
 *   inference loop                    (main.c:141-150)   parallel_inference: batches dealt to the session's GPUs
 *   parallel_postprocess              (main.c:153-155)   sigmoid / threshold / argmax, batch-atomic printing
 *     (default: the three stages pipelined per batch by parallel_classify, output in batch order; GLICLASS_PIPELINE=0 runs
 *      them as the reference's three phases)
 *   release                           (main.c:173-187)
 *
 * usage: gliclass_main /path/to/data.json <prompt_first: true|false|auto> [tokenizer.json] [model dir | model.glcw | synthetic:cfg[:seed]]
 *   (auto = `prompt_first` of the model directory's config.json, the job of run_GLiClass.sh:84-89)
 *   (the two optional arguments default to GLICLASS_TOKENIZER / GLICLASS_MODEL, then include/paths.h)
 * build: make -C gliclass/c_amd gliclass_main
 */
#include <omp.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "configs.h"
#include "model.h"
#include "parallel_processor.h"
#include "paths.h"
#include "postprocessor.h"
#include "preprocessor.h"
#include "read_data.h"
#include "tokenizer.h"

const OrtApi* g_ort = NULL; /* defined by the caller, as in /root/reference/main.c:33 */

static const char* pick(int argc, char** argv, int idx, const char* env, const char* dflt) {
    if (argc > idx && argv[idx][0]) return argv[idx];
    const char* e = getenv(env);
    return (e && *e) ? e : dflt;
}

int main(int argc, char* argv[]) {
    if (argc < 3) {
        printf("Usage: %s /path/to/your_data.json [prompt_first: true/false] [tokenizer.json] [model.glcw]\n", argv[0]);
        printf("NOTE: use this option only if you sure that all required model parts are initialized correctly\n\n");
        return 1;
    }
    char** texts = NULL; size_t num_texts = 0;
    char*** labels = NULL; size_t* num_labels = NULL; size_t num_labels_size = 0;
    bool same_labels = false; char* classification_type = NULL;

    char* json_string = read_file(argv[1]);
    if (!json_string) return 1;
    bool prompt_first;
    if (strcmp(argv[2], "auto") == 0) {          /* what run_GLiClass.sh:84-89 does with jq: take it from the model's config.json */
        int pf = glc_config_prompt_first(pick(argc, argv, 4, "GLICLASS_MODEL", MODEL_PATH));
        if (pf < 0) return 1;
        prompt_first = pf != 0;
    } else prompt_first = string_to_bool(argv[2]);
    parse_json(json_string, &texts, &num_texts, &labels, &num_labels, &num_labels_size, &same_labels, &classification_type);
    printf("DONE: parse_json;\n");
    if (classification_type == NULL) {
        printf("classification type is not provided\n");
        return 1;
    }
    free(json_string);
    if (!texts || !labels || !num_labels) {                  /* the reference would crash further down (main.c:116) */
        fprintf(stderr, "Error: the input file needs \"texts\" and matching \"labels\".\n");
        return 1;
    }

    TokenizerHandle tokenizer_handler = create_tokenizer(pick(argc, argv, 3, "GLICLASS_TOKENIZER", TOKENIZER_PATH));
    if (!tokenizer_handler) return 1;
    printf("DONE: create_tokenizer;\n");

    initialize_ort_api();
    printf("DONE: initialize_ort_api;\n");
    OrtEnv* env = initialize_ort_environment();
    if (env == NULL) {
        fprintf(stderr, "Error: Failed to initialize ONNX Runtime.\n");
        return -1;
    }
    printf("DONE: initialize_ort_environment;\n");
    OrtSession* session = create_ort_session(env, pick(argc, argv, 4, "GLICLASS_MODEL", MODEL_PATH), NUM_THREADS);
    if (session == NULL) {
        fprintf(stderr, "Error: Failed to create session ONNX Runtime.\n");
        g_ort->ReleaseEnv(env);
        return -1;
    }
    printf("DONE: create_ort_session;\n\n");
    fflush(stdout);

    const char* bs_env = getenv("GLICLASS_BATCH_SIZE");
    size_t bs = (bs_env && atol(bs_env) > 0) ? (size_t)atol(bs_env) : BATCH_SIZE;
    size_t num_batches = (num_texts + bs - 1) / bs;
    OrtValue** input_ids_tensors = (OrtValue**)calloc(num_batches ? num_batches : 1, sizeof(OrtValue*));
    OrtValue** attention_mask_tensors = (OrtValue**)calloc(num_batches ? num_batches : 1, sizeof(OrtValue*));
    OrtValue** output_tensors = (OrtValue**)calloc(num_batches ? num_batches : 1, sizeof(OrtValue*));

    const char* pl = getenv("GLICLASS_PIPELINE");
    size_t failed_batches = 0;
    double start_time = omp_get_wtime();
    if (pl && pl[0] == '0') {                     /* the reference's three phases, a barrier after each (main.c:116-155) */
        parallel_preprocess(texts, labels, num_labels, num_texts, same_labels, prompt_first, tokenizer_handler,
                            input_ids_tensors, attention_mask_tensors);
        parallel_inference(session, input_ids_tensors, attention_mask_tensors, num_batches, output_tensors);
        for (size_t i = 0; i < num_batches; i++) failed_batches += output_tensors[i] == NULL;    /* counted before post releases them */
        parallel_postprocess(output_tensors, num_batches, num_texts, texts, labels, num_labels, same_labels, num_labels_size,
                             classification_type);
    } else {                                      /* default: the same stages pipelined per batch, results printed in batch order */
        failed_batches = parallel_classify(session, tokenizer_handler, texts, labels, num_labels, num_texts, same_labels, num_labels_size,
                                           prompt_first, classification_type);
    }
    double end_time = omp_get_wtime();
    printf("Execution time: %f seconds\n", end_time - start_time);

    for (size_t i = 0; i < num_batches; i++) {
        if (input_ids_tensors[i]) g_ort->ReleaseValue(input_ids_tensors[i]);
        if (attention_mask_tensors[i]) g_ort->ReleaseValue(attention_mask_tensors[i]);
    }
    free(input_ids_tensors);
    free(attention_mask_tensors);
    free(output_tensors);
    tokenizers_free(tokenizer_handler);
    g_ort->ReleaseSession(session);
    g_ort->ReleaseEnv(env);
    free_parsed_data(texts, num_texts, labels, num_labels, same_labels, classification_type);
    if (failed_batches) {
        fprintf(stderr, "Error: %zu of %zu batches failed\n", failed_batches, num_batches);
        return 2;
    }
    return 0;
}
