#!/bin/bash
REPO=$(pwd)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder.py -m gpu -x -q -k "f32 or fixture or golden or edge or full_size or c4 or sweep" > gpurun_out/pytest_f32split.log 2>&1; tail -5 gpurun_out/pytest_f32split.log
for mode in native split; do
  for cfg in "base 64 1024" "small 8 512"; do
    set -- $cfg
    GLICLASS_F32_GEMM=$mode python bench.py --dtype f32 --config $1 --batch $2 --seq $3 --steps 5 --warmup 2 --cpu-seqs 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('$mode', '$1', d['value'], d['ms_per_step'], 'err', d['cpu_baseline'].get('gpu_vs_cpu_max_prob_err'), {k: pk[k]['avg_ms'] for k in ('attention','gemm_qkv','gemm_ffn1_gelu','gemm_ffn2','gemm_attn_out') if k in pk})"
  done
done
