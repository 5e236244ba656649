#!/bin/bash
# Developer A/B (GPU box): the MX GEMM's operand requests under different cache-policy bits (gemm256x.hip GLC_GX_A_AUX / GLC_GX_W_AUX),
# one library per variant under gliclass/c_amd/variants/ (built by hand, see profiles/r05/gemm_cache_policy.txt), the c3 forward under
# rocprofv3: per-kernel average duration (two interleaved repetitions) and FETCH_SIZE per launch (one pass).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/gxpol; mkdir -p "$OUT"; export TMPDIR=/tmp
VARS=${VARS:-"base ant wnt asc1"}
BENCH="$REPO/bench.py --steps 3 --warmup 1 --cpu-seqs 0 --no-profile --throughput-dtype none"
cd /tmp
for rep in 1 2; do for v in $VARS; do
  if [ $v = base ]; then unset GLC_HIP_SO; else export GLC_HIP_SO=$REPO/gliclass/c_amd/variants/libgliclass_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t_${v}_$rep" -o trace -- python3 $BENCH > "$OUT/t_${v}_$rep.log" 2>&1; echo "trace $v $rep rc=$?"
done; done
for v in $VARS; do
  if [ $v = base ]; then unset GLC_HIP_SO; else export GLC_HIP_SO=$REPO/gliclass/c_amd/variants/libgliclass_hip_$v.so; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/f_$v" -o pmc -- python3 $BENCH > "$OUT/f_$v.log" 2>&1; echo "fetch $v rc=$?"
done
cd "$REPO"
python3 - "$OUT" $VARS <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
out, vs = sys.argv[1], sys.argv[2:]
def short(n): return re.sub(r"\(GemmArgs.*|\(AttnArgs.*", "", n)
print("avg us per launch (rep1 / rep2) and FETCH MB per launch")
rows = defaultdict(dict)
for v in vs:
    for rep in (1, 2):
        for f in glob.glob(os.path.join(out, f"t_{v}_{rep}/**/*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm256x" in r["Name"] or "attn_mx" in r["Name"]: rows[short(r["Name"])][(v, rep)] = float(r["AverageNs"]) / 1e3
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, f"f_{v}/**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and ("gemm256x" in r["Kernel_Name"] or "attn_mx" in r["Kernel_Name"]):
                a = acc[short(r["Kernel_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (t, n) in acc.items(): rows[k][(v, "f")] = t / n * 1024 * 2.0 / 1e6       # KiB raw, gfx950 correction x2 (guide)
for k in sorted(rows):
    print(k)
    for v in vs:
        d = rows[k]
        print(f"   {v:6s} {d.get((v,1),0):8.1f} / {d.get((v,2),0):8.1f} us   fetch {d.get((v,'f'),0):8.1f} MB")
PY
rm -rf "$OUT"/t_* "$OUT"/f_*/ 2>/dev/null; find "$OUT" -type d -empty -delete 2>/dev/null; true
