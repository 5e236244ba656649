/*
 * Drop-in replacement for /root/reference/include/model.h:8-19 — same seven functions, same
 * argument meaning and error convention (NULL / -1 + message on stderr), implemented on the gfx950
 * engine instead of ONNXRuntime (gliclass/c_amd/host/model.c).
 */
#ifndef MODEL_H
#define MODEL_H

#include <stddef.h>
#include "onnxruntime_c_api.h"
#include "tokenizer.h"

#ifdef __cplusplus
extern "C" {
#endif

extern const OrtApi* g_ort; /* defined by the caller (/root/reference/main.c:33); a weak definition
                               in libgliclass_model.so serves hosts that do not define it */

/* tensors (/root/reference/src/model.c:17-108) */
int64_t* flatten_int_array(int** data, size_t rows, size_t cols);
OrtValue* create_tensor(int64_t* data, size_t rows, size_t cols);
int prepare_input_tensors(TokenizedInputs* tokenized, OrtValue** input_ids_tensor, OrtValue** attention_mask_tensor);

/* session (/root/reference/src/model.c:122-305).  model_path is a .glcw weight blob
 * (gliclass/c_amd/weights.py) or "synthetic:<tiny|mini|small|base|large>[:seed]". */
void initialize_ort_api();
OrtEnv* initialize_ort_environment();
OrtSession* create_ort_session(OrtEnv* env, const char* model_path, int num_threads);
OrtValue* run_inference(OrtSession* session, OrtValue* input_ids_tensor, OrtValue* attention_mask_tensor);

/* ---- extensions (not in the reference) ---- */
/* The inference stage of /root/reference/main.c:141-150 as one call: batches are dealt to the
 * session's GPUs (one host thread + one HIP stream per device). outputs[i] = NULL on failure. */
void parallel_inference(OrtSession* session, OrtValue** input_ids_tensors, OrtValue** attention_mask_tensors,
                        size_t num_batches, OrtValue** output_tensors);
int glc_session_num_devices(const OrtSession* session);

#ifdef __cplusplus
}
#endif
#endif
