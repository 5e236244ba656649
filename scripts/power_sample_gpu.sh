#!/bin/bash
# Runs on the GPU box (via gpurun): samples rocm-smi power / clocks twice a second while bench.py loops, to see whether the forward runs
# against the power cap.  usage: scripts/power_sample_gpu.sh [bench flags...]
R=$(pwd)
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use" | tr '\n' ' '; echo; sleep 0.5; done ) > $R/gpurun_out/power_samples.txt 2>&1 &
SP=$!
sleep 1
python3 $R/bench.py --steps 250 --warmup 5 --cpu-seqs 0 --no-profile --throughput-dtype none "$@" > $R/gpurun_out/power_bench.log 2>&1
wait $SP
tail -1 $R/gpurun_out/power_bench.log | cut -c1-200
sed -n '1,40p' $R/gpurun_out/power_samples.txt | cut -c1-260
