/* Same interface as /root/reference/include/parallel_processor.h:10-16. */
#ifndef PARALLEL_PROCESSOR_H
#define PARALLEL_PROCESSOR_H

#include <stdbool.h>
#include <stdio.h>
#include "configs.h"
#include "onnxruntime_c_api.h"
#include "tokenizer.h"

void parallel_preprocess(char** texts, char*** labels, size_t* num_labels, size_t num_texts, bool same_labels,
                         bool prompt_first, TokenizerHandle tokenizer_handler, OrtValue** input_ids_tensors,
                         OrtValue** attention_mask_tensors);

void parallel_postprocess(OrtValue** output_tensors, size_t num_batches, size_t num_texts, char** texts, char*** labels,
                          size_t* num_labels, bool same_labels, size_t num_labels_size, const char* classification_type);

/* Extension (not in the reference): the three stages of /root/reference/main.c:116-155 pipelined per batch — prompts + tokenizer,
 * run_inference, ordered post-processing — on a team of host threads; same per-batch output bytes, deterministic batch order.
 * Returns the number of failed batches. */
size_t parallel_classify(OrtSession* session, TokenizerHandle tokenizer_handler, char** texts, char*** labels, size_t* num_labels,
                         size_t num_texts, bool same_labels, size_t num_labels_size, bool prompt_first,
                         const char* classification_type);
#endif
