#!/bin/bash
# SQ counter passes for the hot kernels (runs on the GPU box). Usage: scripts/pmc_gpu.sh <tag>
set -u
TAG=${1:-pmc}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
DT=${2:-f32}
BENCH="$REPO/bench.py --steps 1 --warmup 1 --cpu-seqs 0 --no-profile --dtype $DT --throughput-dtype none"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES"
P3="SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- python3 $BENCH > "$OUT/p$i.log" 2>&1
  tail -2 "$OUT/p$i.log"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, re
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:60]
        a = acc[n][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    if not any(t in k for t in ("attn_band", "attn_wg", "attn_mx", "gemm256")): continue
    print("==", k)
    for c, (tot, n) in sorted(cs.items()):
        print(f"   {c:34s} avg/launch = {tot/n:16.0f}")
PY
find "$OUT" -name "*.csv" -size +2M -delete
