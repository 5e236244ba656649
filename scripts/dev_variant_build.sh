#!/bin/bash
# Developer tool: the DEV=1 library of the current sources (developer objects carry the suffix .dev.o, so both flavours build in-tree) as
# gliclass/c_amd/variants/libgliclass_hip_dev.so (git-ignored; travels to the GPU box) — load it with GLC_HIP_SO=$PWD/gliclass/c_amd/variants/libgliclass_hip_dev.so.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
make -C "$REPO/gliclass/c_amd" -j8 DEV=1 devlib 2>&1 | grep -E "error|Error" || true
ls -la "$REPO/gliclass/c_amd/variants/libgliclass_hip_dev.so"
