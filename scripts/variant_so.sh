#!/bin/bash
# Developer tool: a PRODUCT-flavour library in which ONE kernel file is compiled with extra flags — for whole-forward A/Bs of an experiment on one box
# (scripts/ab_so.sh - gliclass/c_amd/variants/libgliclass_hip_<name>.so).  usage: scripts/variant_so.sh <name> <file.hip (under csrc/)> <flags...>
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO/gliclass/c_amd"
NAME=$1; FILE=$2; shift 2
make -j8 libgliclass_hip.so > /dev/null
mkdir -p variants
EXTRA=""; case "$FILE" in attention*) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EXTRA "$@" -I../../include -c csrc/$FILE -o variants/${FILE%.hip}_$NAME.o
OBJS=$(ls csrc/*.o | grep -v "\.dev\.o" | grep -v "csrc/${FILE%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/libgliclass_hip_$NAME.so $OBJS variants/${FILE%.hip}_$NAME.o
ls -la variants/libgliclass_hip_$NAME.so
