#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of one bench.py configuration.
# usage: scripts/kstats_cfg_gpu.sh <tag> <bench flags...>      e.g.  c5 --config dec-1.5b --batch 16 --seq 2048
TAG=${1:-k}; shift
export TMPDIR=/tmp
R=$(pwd)
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_$TAG -o t -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seqs 0 --no-profile --throughput-dtype none "$@" > $R/gpurun_out/kst_$TAG.log 2>&1
cd $R
find gpurun_out/kst_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/kst_$TAG
python3 - "$TAG" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_kernel_stats.csv")))
for r in rows[:14]: print(r["Name"][:110], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
