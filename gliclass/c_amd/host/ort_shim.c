/*
 * The OrtApi vtable slice of include/onnxruntime_c_api.h, implemented over plain host structs.
 * Which reference call site uses which member is listed in that header.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "glc_host_internal.h"

OrtStatus* glc_make_status(const char* fmt, ...) {
    OrtStatus* s = (OrtStatus*)calloc(1, sizeof(OrtStatus));
    if (!s) return NULL; /* out of memory: indistinguishable from success, like a failed malloc in the reference */
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(s->msg, sizeof s->msg, fmt, ap);
    va_end(ap);
    return s;
}

OrtValue* glc_value_new(ONNXTensorElementDataType type, const int64_t* dims, size_t ndim, void* data, int owns) {
    if (ndim > 4) return NULL;
    OrtValue* v = (OrtValue*)calloc(1, sizeof(OrtValue));
    if (!v) return NULL;
    v->type = type;
    v->ndim = ndim;
    for (size_t i = 0; i < ndim; ++i) v->dims[i] = dims[i];
    v->data = data;
    v->owns_data = owns;
    return v;
}

static OrtStatus* api_CreateEnv(OrtLoggingLevel level, const char* logid, OrtEnv** out) {
    if (!out) return glc_make_status("CreateEnv: null output");
    OrtEnv* e = (OrtEnv*)calloc(1, sizeof(OrtEnv));
    if (!e) return glc_make_status("CreateEnv: out of memory");
    e->level = (int)level;
    snprintf(e->logid, sizeof e->logid, "%s", logid ? logid : "");
    *out = e;
    return NULL;
}
static void api_ReleaseEnv(OrtEnv* env) { free(env); }

static void api_ReleaseSession(OrtSession* s) {
    if (!s) return;
    for (int i = 0; i < s->n_engines; ++i) { glc_engine_destroy(s->engines[i]); pthread_mutex_destroy(&s->q[i].mu); pthread_cond_destroy(&s->q[i].cv); }
    free(s);
}
static void api_ReleaseValue(OrtValue* v) {
    if (!v) return;
    if (v->owns_data) free(v->data);
    free(v);
}
static void api_ReleaseStatus(OrtStatus* s) { free(s); }
static const char* api_GetErrorMessage(const OrtStatus* s) { return s ? s->msg : ""; }

static OrtStatus* api_CreateCpuMemoryInfo(OrtAllocatorType t, OrtMemType m, OrtMemoryInfo** out) {
    (void)t; (void)m;
    if (!out) return glc_make_status("CreateCpuMemoryInfo: null output");
    *out = (OrtMemoryInfo*)calloc(1, sizeof(OrtMemoryInfo));
    return *out ? NULL : glc_make_status("CreateCpuMemoryInfo: out of memory");
}
static void api_ReleaseMemoryInfo(OrtMemoryInfo* i) { free(i); }

static OrtStatus* api_CreateTensorWithDataAsOrtValue(const OrtMemoryInfo* info, void* p, size_t len, const int64_t* shape,
                                                     size_t shape_len, ONNXTensorElementDataType type, OrtValue** out) {
    (void)info;
    if (!p || !shape || !out || shape_len == 0 || shape_len > 4) return glc_make_status("CreateTensorWithDataAsOrtValue: bad argument");
    size_t esz = type == ONNX_TENSOR_ELEMENT_DATA_TYPE_INT64 ? 8 : type == ONNX_TENSOR_ELEMENT_DATA_TYPE_FLOAT ? 4 : 0;
    if (!esz) return glc_make_status("CreateTensorWithDataAsOrtValue: unsupported element type %d", (int)type);
    size_t n = 1;
    for (size_t i = 0; i < shape_len; ++i) { if (shape[i] < 0) return glc_make_status("negative dimension"); n *= (size_t)shape[i]; }
    if (n * esz != len) return glc_make_status("CreateTensorWithDataAsOrtValue: buffer is %zu bytes, shape needs %zu", len, n * esz);
    *out = glc_value_new(type, shape, shape_len, p, 0);
    return *out ? NULL : glc_make_status("CreateTensorWithDataAsOrtValue: out of memory");
}

static OrtStatus* api_GetTensorTypeAndShape(const OrtValue* v, OrtTensorTypeAndShapeInfo** out) {
    if (!v || !out) return glc_make_status("GetTensorTypeAndShape: null argument");
    OrtTensorTypeAndShapeInfo* i = (OrtTensorTypeAndShapeInfo*)calloc(1, sizeof(*i));
    if (!i) return glc_make_status("GetTensorTypeAndShape: out of memory");
    i->type = v->type;
    i->ndim = v->ndim;
    memcpy(i->dims, v->dims, sizeof i->dims);
    *out = i;
    return NULL;
}
static OrtStatus* api_GetDimensionsCount(const OrtTensorTypeAndShapeInfo* i, size_t* out) {
    if (!i || !out) return glc_make_status("GetDimensionsCount: null argument");
    *out = i->ndim;
    return NULL;
}
static OrtStatus* api_GetDimensions(const OrtTensorTypeAndShapeInfo* i, int64_t* d, size_t n) {
    if (!i || !d) return glc_make_status("GetDimensions: null argument");
    for (size_t k = 0; k < n && k < i->ndim; ++k) d[k] = i->dims[k];
    return NULL;
}
static OrtStatus* api_GetTensorElementType(const OrtTensorTypeAndShapeInfo* i, ONNXTensorElementDataType* out) {
    if (!i || !out) return glc_make_status("GetTensorElementType: null argument");
    *out = i->type;
    return NULL;
}
static OrtStatus* api_GetTensorMutableData(OrtValue* v, void** out) {
    if (!v || !out) return glc_make_status("GetTensorMutableData: null argument");
    *out = v->data;
    return NULL;
}
static void api_ReleaseTensorTypeAndShapeInfo(OrtTensorTypeAndShapeInfo* i) { free(i); }

static const OrtApi k_api = {
    api_CreateEnv, api_ReleaseEnv, api_ReleaseSession, api_ReleaseValue, api_ReleaseStatus, api_GetErrorMessage,
    api_CreateCpuMemoryInfo, api_ReleaseMemoryInfo, api_CreateTensorWithDataAsOrtValue, api_GetTensorTypeAndShape,
    api_GetDimensionsCount, api_GetDimensions, api_GetTensorElementType, api_GetTensorMutableData,
    api_ReleaseTensorTypeAndShapeInfo,
};
static const OrtApi* base_GetApi(uint32_t version) { return version <= ORT_API_VERSION ? &k_api : NULL; }
static const char* base_GetVersionString(void) { return "gliclass-mi355x shim (no ONNXRuntime)"; }
static const OrtApiBase k_base = {base_GetApi, base_GetVersionString};
const OrtApiBase* OrtGetApiBase(void) { return &k_base; }
