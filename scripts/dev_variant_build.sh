#!/bin/bash
# Developer tool: build a DEV=1 library of the current sources in a scratch copy (/tmp/devb) and place it under gliclass/c_amd/variants/
# (git-ignored; travels to the GPU box) — load it with GLC_HIP_SO=$PWD/gliclass/c_amd/variants/libgliclass_hip_dev.so.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/devb/gliclass
(cd "$REPO" && tar cf - --exclude='*.o' --exclude='*.so' --exclude=variants gliclass/c_amd include examples) | tar xf - -C /tmp/devb      # (sources only: the copy keeps its own objects, make rebuilds what changed)
make -C /tmp/devb/gliclass/c_amd -j8 DEV=1 libgliclass_hip.so 2>&1 | grep -E "error|Error" || true
mkdir -p "$REPO/gliclass/c_amd/variants"
cp /tmp/devb/gliclass/c_amd/libgliclass_hip.so "$REPO/gliclass/c_amd/variants/libgliclass_hip_dev.so"
ls -la "$REPO/gliclass/c_amd/variants/libgliclass_hip_dev.so"
