/*
 * C tokenizer API the reference programs against: mlc-ai/tokenizers-cpp's `tokenizers_c.h` (a git submodule whose
 * directory is empty in /root/reference, .gitmodules:1-3; HEAD not recorded).  The call sites that fix the shape of
 * this interface are /root/reference/src/tokenizer.c:20 (`TokenizerEncodeResult* results`), :33
 * (`tokenizers_encode_batch(tokenizer, inputs, input_lengths, num_texts, add_special_tokens, results)`), :46-49,75
 * (`results[i].len`, `results[i].token_ids[j]`), :86 (`tokenizers_free_encode_results(results, num_texts)`), :175
 * (`tokenizers_new_from_str(json, json_len)`) and /root/reference/main.c:184 (`tokenizers_free`).
 *
 * Here the functions are implemented natively in C (gliclass/c_amd/host/tokenizer.c) -- no Rust, no tokenizers-cpp:
 * a reader of HF `tokenizer.json` for the DeBERTa-v3 family (Strip / Precompiled / Replace normalisers, Metaspace
 * pre-tokeniser, Unigram model, added tokens, TemplateProcessing).  Anything else in the file (BPE, WordPiece, other
 * normalisers ...) makes tokenizers_new_from_str fail with a message naming the unsupported piece.
 */
#ifndef TOKENIZERS_C_H_
#define TOKENIZERS_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* TokenizerHandle;

typedef struct {
    int* token_ids;
    size_t len;
} TokenizerEncodeResult;

/* NULL on failure (message on stderr). */
TokenizerHandle tokenizers_new_from_str(const char* json, size_t len);

void tokenizers_encode(TokenizerHandle handle, const char* data, size_t len, int add_special_token, TokenizerEncodeResult* result);
/* Encodes `num_seqs` texts (OpenMP across texts); results[i].token_ids is malloc'ed, release with tokenizers_free_encode_results. */
void tokenizers_encode_batch(TokenizerHandle handle, const char** data, size_t* len, size_t num_seqs, int add_special_token,
                             TokenizerEncodeResult* results);
void tokenizers_free_encode_results(TokenizerEncodeResult* results, size_t num_seqs);

/* Decoded text is kept inside the handle (not thread-safe, like tokenizers-cpp); fetch it with tokenizers_get_decode_str. */
void tokenizers_decode(TokenizerHandle handle, const uint32_t* data, size_t len, int skip_special_token);
void tokenizers_get_decode_str(TokenizerHandle handle, const char** data, size_t* len);

void tokenizers_get_vocab_size(TokenizerHandle handle, size_t* size);
void tokenizers_id_to_token(TokenizerHandle handle, uint32_t id, const char** data, size_t* len);
/* *id = -1 when the token is unknown. */
void tokenizers_token_to_id(TokenizerHandle handle, const char* token, size_t len, int32_t* id);

void tokenizers_free(TokenizerHandle handle);

/* Extension (test hook, not in tokenizers-cpp): runs only the normaliser chain; returns a malloc'ed UTF-8 string. */
char* glc_tokenizer_normalize(TokenizerHandle handle, const char* data, size_t len, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
