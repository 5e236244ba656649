#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the bench in one arithmetic mode, then two PMC passes
# (FETCH_SIZE, WRITE_SIZE — they do not fit one pass on gfx950; never combined with trace domains other than --kernel-trace) and a
# text summary + traffic.json (per dtype, merged into profiles/traffic.json by hand).
# Usage: scripts/profile_gpu.sh <tag> [dtype f32|f16|bf16] [extra bench flags...]
set -u
TAG=${1:-r02}
DT=${2:-f32}
shift; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="$REPO/bench.py --steps 3 --warmup 1 --cpu-seqs 0 --no-profile --dtype $DT --throughput-dtype none $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 $BENCH > "$OUT/pmc_write.log" 2>&1
cd "$REPO"
python3 scripts/summarize_prof.py "$OUT" "$DT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# keep the merged artefacts small
find "$OUT" -name "*.csv" -size +4M -delete
