#!/bin/bash
# CPU only: rebuild the pure-C host layer with AddressSanitizer + UBSan and run its tests against that build
# (GPU sanitizers are not available on this pool; the HIP side is covered by the parity tests).
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/glc_asan}
mkdir -p "$OUT"
cd "$REPO/gliclass/c_amd"
for f in model ort_shim glc_weights parallel_processor postprocessor preprocessor tokenizer glc_json read_data glc_safetensors; do
    gcc -O1 -g -D_GNU_SOURCE -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -fPIC -I../../include -c host/$f.c -o "$OUT/$f.o"
done
gcc -shared -fsanitize=address,undefined -fopenmp -o "$OUT/libgliclass_model.so" "$OUT"/*.o -L. -lgliclass_hip -Wl,-rpath,"$PWD" -lm
cd "$REPO"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" GLC_MODEL_SO="$OUT/libgliclass_model.so" \
    python -m pytest tests/test_host.py tests/test_tokenizer.py tests/test_tokenizer_bpe.py -x -q
