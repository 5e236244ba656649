#!/bin/bash
# Runs on the GPU box: WRITE_SIZE / FETCH_SIZE per launch of the three MX attention kernels (scripts/attn_write_probe.py) -> gpurun_out/attn_write.txt
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/attn_write; mkdir -p $O
cd /tmp
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o pmc -- python3 $R/scripts/attn_write_probe.py > $O/$C.log 2>&1
done
cd $R
python3 - $O > $R/gpurun_out/attn_write.txt <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
out = sys.argv[1]
for C, corr in (("WRITE_SIZE", 1.0), ("FETCH_SIZE", 2.0)):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, C, "**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
            if "attn_" not in n: continue
            a = acc[n]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for n, (t, k) in sorted(acc.items()):
        print(f"{C:11s} {n:50s} launches {k:3d}  avg {t / k * 1024 * corr / 1e6:9.1f} MB per launch (KiB x 1024{', x2 gfx950' if corr == 2 else ''})")
PY
rm -rf $O
cat $R/gpurun_out/attn_write.txt
