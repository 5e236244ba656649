#!/bin/bash
# A/B of whole-library builds on ONE box: scripts/ab_so.sh <lib1.so|-> <lib2.so> ...  ("-" = the default in-tree build), 2 rounds
REPO=$(pwd)
for round in 1 2; do
  for so in "$@"; do
    if [ "$so" = "-" ]; then envs="GLC_X=0"; else envs="GLC_HIP_SO=$REPO/$so"; fi
    env $envs python3 $REPO/bench.py --steps 5 --warmup 2 --cpu-seqs 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('round $round [$so]', d['ms_per_step'], {k: pk[k]['avg_ms'] for k in ('attention','gemm_qkv','gemm_ffn1_gelu','gemm_ffn2','gemm_attn_out','last_layer_pruned')})"
  done
done
