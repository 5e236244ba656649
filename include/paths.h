/* Default file locations of the launcher, as /root/reference/include/paths.h:4-5; the model is this build's weight
 * blob (gliclass/c_amd/weights.py::write_blob) instead of onnx/model.onnx.  Both can be overridden at run time:
 * argv[3] / GLICLASS_TOKENIZER and argv[4] / GLICLASS_MODEL (examples/gliclass_main.c). */
#ifndef PATHS_H
#define PATHS_H
#define TOKENIZER_PATH "tokenizer/tokenizer.json"
#define MODEL_PATH "model/model.glcw"
#endif
