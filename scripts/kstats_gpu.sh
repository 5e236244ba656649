#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of a few c3 forwards (scripts/c3_forwards.py) -> gpurun_out/<tag>_kernel_stats.csv + top lines
# usage: scripts/kstats_gpu.sh <tag>      (environment switches such as GLC_MX pass through)
TAG=${1:-k}
export TMPDIR=/tmp
R=$(pwd)
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_$TAG -o t -- python3 $R/scripts/c3_forwards.py > $R/gpurun_out/kst_$TAG.log 2>&1
cd $R
find gpurun_out/kst_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/kst_$TAG
python3 - "$TAG" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_kernel_stats.csv")))
for r in rows[:12]: print(r["Name"][:100], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
