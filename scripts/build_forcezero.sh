#!/bin/bash
# Builds gliclass/c_amd/_ab/libgliclass_hip_fzall.so: every HIP translation unit with -mllvm -amdgpu-waitcnt-forcezero (scripts/waitcnt_screen.py)
set -e
cd "$(dirname "$0")/../gliclass/c_amd"
mkdir -p _ab /tmp/glc_fz
for f in engine gemm gemm256 gemm256s attention attention_wg decoder rows; do
  X=""; case $f in attention|attention_wg) X="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $X -mllvm -amdgpu-waitcnt-forcezero -I../../include -c csrc/$f.hip -o /tmp/glc_fz/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o _ab/libgliclass_hip_fzall.so /tmp/glc_fz/*.o
echo built _ab/libgliclass_hip_fzall.so
