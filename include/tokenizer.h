/*
 * Same interface as /root/reference/include/tokenizer.h:13-25.  `TokenizedInputs` is the boundary INPUT type of the
 * hot path (filled at /root/reference/src/tokenizer.c:58-84).  The functions are implemented natively in
 * gliclass/c_amd/host/tokenizer.c on top of include/tokenizers_c.h (no Rust tokenizers-cpp needed); an integrator who
 * links the reference's own src/tokenizer.c + tokenizers-cpp into the executable overrides them (ELF interposition).
 */
#ifndef TOKENIZER_H
#define TOKENIZER_H

#include <stdbool.h>
#include <stddef.h>

#include "tokenizers_c.h"

typedef struct {
    int** input_ids;      /* [batch_size][seq_length], one malloc per row */
    int** token_type_ids; /* all zero, unused by the model */
    int** attention_mask; /* 1 = token, 0 = padding */
    size_t batch_size;
    size_t seq_length;    /* longest row of the batch after truncation */
} TokenizedInputs;

TokenizedInputs tokenize_inputs(TokenizerHandle tokenizer, const char* inputs[], size_t num_texts, size_t max_length);
void print_tokenized_inputs(const TokenizedInputs* tokenized);
void free_tokenized_inputs(TokenizedInputs* tokenized);
TokenizerHandle create_tokenizer(const char* filepath);

#endif
