#!/bin/bash
# fp32 (parity-grade) mode A/B on ONE box: f32 parity tests with the default (split-f16) kernels, then bench lines for
# {fp32-MFMA, split-f16} GEMMs x {fp32-MFMA, split-f16} band attention at configs c3 (base 64x1024) and c2 (small 8x512).
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_decoder.py -m gpu -x -q -k "f32 or fixture or golden or edge or full_size or c4 or sweep" > gpurun_out/pytest_f32split.log 2>&1; tail -4 gpurun_out/pytest_f32split.log
for mode in "native native" "split native" "split split"; do
  set -- $mode; gm=$1; am=$2
  for cfg in "base 64 1024" "small 8 512"; do
    set -- $cfg
    GLICLASS_F32_GEMM=$gm GLICLASS_F32_ATTN=$am python bench.py --dtype f32 --config $1 --batch $2 --seq $3 --steps 5 --warmup 2 --cpu-seqs 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('gemm=$gm attn=$am', '$1', d['value'], d['ms_per_step'], 'err', d['cpu_baseline'].get('gpu_vs_cpu_max_prob_err'), {k: pk[k]['avg_ms'] for k in ('attention','gemm_qkv','gemm_ffn1_gelu','gemm_ffn2','gemm_attn_out','last_layer_pruned') if k in pk})"
  done
done
