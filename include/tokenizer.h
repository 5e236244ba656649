/*
 * Boundary INPUT type of the hot path: the struct `tokenize_inputs` returns in the reference
 * (/root/reference/include/tokenizer.h:13-19, filled at /root/reference/src/tokenizer.c:58-84).
 * The tokenizer itself (Rust tokenizers-cpp) is out of scope; an integrator links
 * the reference's own src/tokenizer.c, which provides tokenize_inputs()/free_tokenized_inputs().
 */
#ifndef TOKENIZER_H
#define TOKENIZER_H

#include <stdbool.h>
#include <stddef.h>

#ifndef TOKENIZERS_C_H_
typedef void* TokenizerHandle; /* tokenizers-cpp's opaque handle (tokenizers_c.h) */
#endif

typedef struct {
    int** input_ids;      /* [batch_size][seq_length], one malloc per row */
    int** token_type_ids; /* all zero, unused by the model */
    int** attention_mask; /* 1 = token, 0 = padding */
    size_t batch_size;
    size_t seq_length;    /* longest row of the batch after truncation */
} TokenizedInputs;

TokenizedInputs tokenize_inputs(TokenizerHandle tokenizer, const char* inputs[], size_t num_texts, size_t max_length);
void free_tokenized_inputs(TokenizedInputs* tokenized);

#endif
