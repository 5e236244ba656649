/* Private definitions behind the opaque Ort* handles of include/onnxruntime_c_api.h. */
#ifndef GLC_HOST_INTERNAL_H
#define GLC_HOST_INTERNAL_H
#include <stdint.h>
#include "gliclass_hip.h"
#include "onnxruntime_c_api.h"

#define GLC_MAX_DEVICES 16

struct OrtStatus { char msg[256]; };
struct OrtEnv { char logid[64]; int level; };
struct OrtMemoryInfo { int dummy; };
struct OrtValue {
    ONNXTensorElementDataType type;
    size_t ndim;
    int64_t dims[4];
    void* data;
    int owns_data;          /* outputs of run_inference and inputs made by create_tensor own their buffer */
};
struct OrtTensorTypeAndShapeInfo { ONNXTensorElementDataType type; size_t ndim; int64_t dims[4]; };
struct OrtSession {
    glc_model_config cfg;
    int n_engines;
    glc_engine* engines[GLC_MAX_DEVICES];
    int devices[GLC_MAX_DEVICES];
    volatile unsigned next;  /* round-robin cursor */
};

OrtStatus* glc_make_status(const char* fmt, ...);
OrtValue* glc_value_new(ONNXTensorElementDataType type, const int64_t* dims, size_t ndim, void* data, int owns);
#endif
