#!/bin/bash
# Runs on the GPU box (via gpurun): FETCH_SIZE per kernel over a few c3 forwards (scripts/c3_forwards.py); gfx950: x2 (MI355X_MICROARCH.md HBM section)
TAG=${1:-f}
export TMPDIR=/tmp
R=$(pwd)
cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/fetch_$TAG -o pmc -- python3 $R/scripts/c3_forwards.py > $R/gpurun_out/fetch_$TAG.log 2>&1
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"gpurun_out/fetch_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"][:90]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
    print(f"{k:92s} launches={n:4d} fetch_MB_per_launch={t / n * 1024 * 2 / 1e6:9.1f}")
PY
rm -rf gpurun_out/fetch_$TAG
