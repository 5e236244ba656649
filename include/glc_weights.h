/* Host-side weight source for create_ort_session(): .glcw blob reader + deterministic synthetic
 * weights (the C twin of gliclass/c_amd/prng.py / weights.py). Pure C, no GPU. */
#ifndef GLC_WEIGHTS_H
#define GLC_WEIGHTS_H
#include <stddef.h>
#include <stdint.h>
#include "gliclass_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct glc_weights {
    glc_model_config cfg;
    int n_tensors;
    const float** tensors;  /* canonical order of gliclass_hip.h */
    void* _map;             /* mmap of the blob (or NULL) */
    size_t _map_len;
    float* _owned;          /* synthetic: one allocation holding every tensor */
} glc_weights;

/* path = .glcw blob | "synthetic:<name>[:seed]" | HF checkpoint (a directory with config.json + model.safetensors, or
 * the .safetensors file itself).  Returns 0 / -1 (message on stderr). */
int glc_weights_load(const char* path, glc_weights* out);
/* The HF-checkpoint branch of glc_weights_load (host/glc_safetensors.c). `out` must be zeroed. */
int glc_load_hf_checkpoint(const char* path, glc_weights* out);
void glc_weights_free(glc_weights* w);

uint64_t glc_fnv1a64(const char* s);
/* n floats uniform in [mean-amp, mean+amp): element i is a pure function of (seed, name, i) */
void glc_prng_fill(uint64_t seed, const char* name, size_t n, double amp, double mean, float* out);
int glc_named_config(const char* name, glc_model_config* out);
/* canonical tensor list: fills name (<=95 chars), shape, amp, mean for index i; returns ndim or -1 */
int glc_tensor_spec(const glc_model_config* cfg, int i, char* name, uint64_t shape[4], double* amp, double* mean);

#ifdef __cplusplus
}
#endif
#endif
