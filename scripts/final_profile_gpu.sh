#!/bin/bash
# Runs on the GPU box (via gpurun): ONE measurement set of ONE build — the c3 bench line, the rocprofv3 kernel trace + stats of the same
# command, the FETCH_SIZE / WRITE_SIZE passes (separate: they do not share a pass on gfx950) and an SQ pass (matrix-pipe busy cycles, f16 and
# f8 MFMA ops) — summarised into gpurun_out/final_<tag>/ {bench_c3.json, kernel_stats.csv, summary.txt, pmc_sq.txt, traffic.json}, every
# file stamped with the commit the caller passes.  Copy the directory to profiles/<tag>/ and traffic.json over profiles/traffic.json,
# then run bench.py once more for the committed line (its roofline.traffic reads profiles/traffic.json).
# usage: scripts/final_profile_gpu.sh <tag> <commit>
set -u
TAG=${1:-r04}
COMMIT=${2:-unknown}
REPO=$(pwd)
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp GLICLASS_BENCH_COMMIT=$COMMIT
echo "$COMMIT" > "$OUT/commit.txt"
python3 bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"; echo "bench rc=$?"
cd /tmp
BENCH="$REPO/bench.py --steps 3 --warmup 1 --cpu-seqs 0 --no-profile --throughput-dtype none"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $BENCH > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 $BENCH > "$OUT/pmc_fetch.log" 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 $BENCH > "$OUT/pmc_write.log" 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o pmc -- python3 $BENCH > "$OUT/pmc_sq.log" 2>&1; echo "sq rc=$?"
cd "$REPO"
python3 scripts/summarize_prof.py "$OUT" f32 base:64:1024 "$COMMIT" > "$OUT/summary.txt" 2>&1
python3 - "$OUT" "$COMMIT" > "$OUT/pmc_sq.txt" <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
out, commit = sys.argv[1], sys.argv[2]
print("commit", commit)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "pmc_sq/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
        a = acc[n][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    if not any(t in k for t in ("attn_", "gemm256")): continue
    print("==", k)
    for c, (tot, n) in sorted(cs.items()):
        print(f"   {c:34s} avg/launch = {tot/n:16.0f}   ({n} launches)")
PY
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
head -12 "$OUT/summary.txt"; tail -c 400 "$OUT/bench_c3.json"
