#!/bin/bash
# Runs on the GPU box (via gpurun): ONE measurement set of ONE build — the c3 bench line, the rocprofv3 kernel trace + stats of the same
# command, the FETCH_SIZE / WRITE_SIZE passes (separate: they do not share a pass on gfx950) and an SQ pass (matrix-pipe busy cycles, f16 and
# f8 MFMA ops) — summarised into gpurun_out/final_<tag>/ {bench_c3.json, kernel_stats.csv, summary.txt, pmc_sq.txt, traffic.json}, every
# file stamped with the commit the caller passes.  Copy the directory to profiles/<tag>/ and traffic.json over profiles/traffic.json,
# then run bench.py once more for the committed line (its roofline.traffic reads profiles/traffic.json).
# Round 5: also the 16-bit throughput mode's FETCH / WRITE passes (traffic.json's f16 entry carries the same commit) and config c5 (bench line + kernel stats).
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
# (round 6, review item 4) second SQ pass + the clock: what the pipe waits for, per kernel — issue stalls, LDS, vector memory, MFMA / VALU co-execution
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_FLAT SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -o pmc -- python3 $BENCH > "$OUT/pmc_sq2.log" 2>&1; echo "sq2 rc=$?"
cd "$REPO"
python3 scripts/summarize_prof.py "$OUT" f32 base:64:1024 "$COMMIT" > "$OUT/summary.txt" 2>&1
python3 scripts/pipe_account.py "$OUT" "$COMMIT" > "$OUT/pipe_account.txt" 2>&1
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
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/pmc_sq2"
# (round 5) the opt-in 16-bit throughput mode's FETCH / WRITE passes of the SAME build: the f16 entry of traffic.json (it used to be a round-1 pass)
mkdir -p "$OUT/f16"
cd /tmp
BENCH16="$REPO/bench.py --dtype f16 --steps 3 --warmup 1 --cpu-seqs 0 --no-profile --throughput-dtype none"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/f16/pmc_fetch" -o pmc -- python3 $BENCH16 > "$OUT/f16/pmc_fetch.log" 2>&1; echo "f16 fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/f16/pmc_write" -o pmc -- python3 $BENCH16 > "$OUT/f16/pmc_write.log" 2>&1; echo "f16 write rc=$?"
cd "$REPO"
python3 scripts/summarize_prof.py "$OUT/f16" f16 base:64:1024 "$COMMIT" > "$OUT/f16/summary.txt" 2>&1
python3 - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
a = json.load(open(os.path.join(out, "traffic.json")))
b = json.load(open(os.path.join(out, "f16", "traffic.json")))
b["f16"]["source"] = os.path.basename(out.rstrip("/")) + "/f16"
a.update(b)
json.dump(a, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("traffic.json:", {k: (v.get("source"), v.get("commit"), sorted(v.get("kernels", {}))) for k, v in a.items()})
PY
rm -rf "$OUT/f16/pmc_fetch" "$OUT/f16/pmc_write"
# (round 5, review item 7) config c5 (decoder backbone, B = 16, S = 2048) of the same build: bench line + rocprofv3 kernel stats
python3 bench.py --config c5 --cpu-seqs 0 --throughput-dtype none > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"; echo "c5 bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c5" -o trace -- python3 $REPO/bench.py --config c5 --steps 2 --warmup 1 --cpu-seqs 0 --no-profile --throughput-dtype none > "$OUT/trace_c5.log" 2>&1; echo "c5 trace rc=$?"
cd "$REPO"
find "$OUT/trace_c5" -name "*kernel_stats.csv" -exec cp {} "$OUT/c5_kernel_stats.csv" \;
rm -rf "$OUT/trace_c5"
# (round 6, review item 6) BASELINE.json's c4 as ONE GPU sees it: gliclass-large, 32 rows of 1024 — bench line + kernel stats
python3 bench.py --config large --batch 32 --cpu-seqs 0 --throughput-dtype none > "$OUT/bench_c4_shard.json" 2> "$OUT/bench_c4_shard.err"; echo "c4 shard bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c4" -o trace -- python3 $REPO/bench.py --config large --batch 32 --steps 2 --warmup 1 --cpu-seqs 0 --no-profile --throughput-dtype none > "$OUT/trace_c4.log" 2>&1; echo "c4 trace rc=$?"
cd "$REPO"
find "$OUT/trace_c4" -name "*kernel_stats.csv" -exec cp {} "$OUT/c4_shard_kernel_stats.csv" \;
rm -rf "$OUT/trace_c4"
head -12 "$OUT/summary.txt"; cat "$OUT/pipe_account.txt"; tail -c 400 "$OUT/bench_c3.json"
