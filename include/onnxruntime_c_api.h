/*
 * Minimal stand-in for the slice of ONNXRuntime's C API that the reference's translation units
 * reach through `g_ort`.  This is NOT ONNXRuntime and links no ONNXRuntime code:
 * the "session" is a gfx950 GLiClass engine (include/gliclass_hip.h) and `OrtValue` is a plain host
 * tensor.  Only the members below exist; names/argument meaning follow the call sites:
 *
 *   ReleaseValue                  /root/reference/main.c:174-175, src/parallel_processor.c:88
 *   ReleaseSession / ReleaseEnv   /root/reference/main.c:186-187,96
 *   GetTensorTypeAndShape         /root/reference/src/postprocessor.c:39
 *   GetDimensionsCount            /root/reference/src/postprocessor.c:48
 *   GetDimensions                 /root/reference/src/postprocessor.c:58
 *   GetTensorMutableData          /root/reference/src/postprocessor.c:75
 *   ReleaseTensorTypeAndShapeInfo /root/reference/src/postprocessor.c:51,62,79,154
 *   ReleaseStatus / GetErrorMessage  /root/reference/src/postprocessor.c:42,52,63,80,155; src/model.c:44
 *   CreateTensorWithDataAsOrtValue, CreateCpuMemoryInfo, ReleaseMemoryInfo  /root/reference/src/model.c:41-61
 *
 * Member ORDER differs from the real OrtApi (callers are recompiled against this header).
 */
#ifndef GLC_ONNXRUNTIME_C_API_SHIM_H
#define GLC_ONNXRUNTIME_C_API_SHIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORT_API_VERSION 19

typedef struct OrtStatus OrtStatus;
typedef struct OrtEnv OrtEnv;
typedef struct OrtSession OrtSession;
typedef struct OrtValue OrtValue;
typedef struct OrtMemoryInfo OrtMemoryInfo;
typedef struct OrtTensorTypeAndShapeInfo OrtTensorTypeAndShapeInfo;

typedef enum ONNXTensorElementDataType {
    ONNX_TENSOR_ELEMENT_DATA_TYPE_UNDEFINED = 0,
    ONNX_TENSOR_ELEMENT_DATA_TYPE_FLOAT = 1,
    ONNX_TENSOR_ELEMENT_DATA_TYPE_INT64 = 7
} ONNXTensorElementDataType;

typedef enum OrtLoggingLevel { ORT_LOGGING_LEVEL_VERBOSE, ORT_LOGGING_LEVEL_INFO, ORT_LOGGING_LEVEL_WARNING,
                               ORT_LOGGING_LEVEL_ERROR, ORT_LOGGING_LEVEL_FATAL } OrtLoggingLevel;
typedef enum OrtAllocatorType { OrtInvalidAllocator = -1, OrtDeviceAllocator = 0, OrtArenaAllocator = 1 } OrtAllocatorType;
typedef enum OrtMemType { OrtMemTypeCPUInput = -2, OrtMemTypeCPUOutput = -1, OrtMemTypeCPU = OrtMemTypeCPUOutput,
                          OrtMemTypeDefault = 0 } OrtMemType;

typedef struct OrtApi {
    OrtStatus* (*CreateEnv)(OrtLoggingLevel level, const char* logid, OrtEnv** out);
    void (*ReleaseEnv)(OrtEnv* env);
    void (*ReleaseSession)(OrtSession* session);
    void (*ReleaseValue)(OrtValue* value);
    void (*ReleaseStatus)(OrtStatus* status);
    const char* (*GetErrorMessage)(const OrtStatus* status);
    OrtStatus* (*CreateCpuMemoryInfo)(OrtAllocatorType type, OrtMemType mem_type, OrtMemoryInfo** out);
    void (*ReleaseMemoryInfo)(OrtMemoryInfo* info);
    /* wraps `p_data` without copying and without taking ownership (as ONNXRuntime does) */
    OrtStatus* (*CreateTensorWithDataAsOrtValue)(const OrtMemoryInfo* info, void* p_data, size_t p_data_len,
                                                 const int64_t* shape, size_t shape_len,
                                                 ONNXTensorElementDataType type, OrtValue** out);
    OrtStatus* (*GetTensorTypeAndShape)(const OrtValue* value, OrtTensorTypeAndShapeInfo** out);
    OrtStatus* (*GetDimensionsCount)(const OrtTensorTypeAndShapeInfo* info, size_t* out);
    OrtStatus* (*GetDimensions)(const OrtTensorTypeAndShapeInfo* info, int64_t* dim_values, size_t dim_values_length);
    OrtStatus* (*GetTensorElementType)(const OrtTensorTypeAndShapeInfo* info, ONNXTensorElementDataType* out);
    OrtStatus* (*GetTensorMutableData)(OrtValue* value, void** out);
    void (*ReleaseTensorTypeAndShapeInfo)(OrtTensorTypeAndShapeInfo* info);
} OrtApi;

typedef struct OrtApiBase {
    const OrtApi* (*GetApi)(uint32_t version);
    const char* (*GetVersionString)(void);
} OrtApiBase;

const OrtApiBase* OrtGetApiBase(void);

#ifdef __cplusplus
}
#endif
#endif
