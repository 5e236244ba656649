#!/bin/bash
# A/B on ONE box: runs bench.py under each env setting given as arguments ("VAR=1" or "-" for baseline), 2 rounds.
REPO=$(pwd)
for round in 1 2; do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs python3 $REPO/bench.py --steps 5 --warmup 2 --cpu-seqs 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('round $round [$setting]', d['ms_per_step'], {k: pk[k]['avg_ms'] for k in ('attention','gemm_qkv','gemm_ffn1_gelu','gemm_ffn2','gemm_attn_out','last_layer_pruned')})"
  done
done
