/* Private definitions behind the opaque Ort* handles of include/onnxruntime_c_api.h. */
#ifndef GLC_HOST_INTERNAL_H
#define GLC_HOST_INTERNAL_H
#include <pthread.h>
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
/* One pending run_inference call (host/model.c, request coalescing). */
typedef struct glc_req {
    const int64_t* ids; const int64_t* mask;
    int B, S, C;
    float* logits;            /* [B, max(C,1)], owned by the caller */
    int status;               /* 0 pending, 1 done, -1 failed */
    struct glc_req* next;
} glc_req;
typedef struct glc_queue {
    pthread_mutex_t mu; pthread_cond_t cv;
    glc_req *head, *tail;
    int busy;                 /* a leader thread is serving this engine */
} glc_queue;

struct OrtSession {
    glc_model_config cfg;
    int n_engines;
    glc_engine* engines[GLC_MAX_DEVICES];
    int devices[GLC_MAX_DEVICES];
    volatile unsigned next;  /* round-robin cursor */
    glc_queue q[GLC_MAX_DEVICES];   /* per engine: concurrent run_inference calls waiting for it */
    int coalesce_rows;       /* merge waiting calls into one forward of up to this many rows (GLICLASS_COALESCE_ROWS, default 64; <= 1 = off) */
};

OrtStatus* glc_make_status(const char* fmt, ...);
OrtValue* glc_value_new(ONNXTensorElementDataType type, const int64_t* dims, size_t ndim, void* data, int owns);
#endif
