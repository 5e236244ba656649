#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/l2; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for ORD in 0 1 2; do
  export GLC_GEMM_ORDER=$ORD
  python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-seqs 0 > $OUT/bench_o$ORD.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$OUT/bench_o$ORD.json')); pk=d['roofline']['per_kernel']; print('order $ORD', d['ms_per_step'], {k: pk[k]['avg_ms'] for k in pk if 'gemm' in k})"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/o$ORD -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 1 --cpu-seqs 0 --no-profile > $OUT/o$ORD.log 2>&1
  python3 - $OUT/o$ORD <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0,0]))
for f in glob.glob(os.path.join(sys.argv[1], "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:44]
        a = acc[n][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k, cs in acc.items():
    if "gemm256" in k or "attn_band" in k:
        print("  ", k, {c: round(t/n/1e6,2) for c,(t,n) in cs.items()}, "(M per launch)")
PY
done
find $OUT -name "*.csv" -size +1M -delete
