OUT=gpurun_out/r06; mkdir -p $OUT
V=gliclass/c_amd/variants
GLC_HIP_SO=$PWD/$V/libgliclass_hip_ring5.so timeout -k 10 600 python3 -m pytest tests/test_gpu_mx.py -q -x -k "epilogue or gemm" 2>&1 | tail -3
bash scripts/ab_so.sh - $V/libgliclass_hip_ring5.so > $OUT/ab_ring5.txt 2>&1; cat $OUT/ab_ring5.txt
GLC_HIP_SO=$PWD/$V/libgliclass_hip_ring5.so timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --cpu-seqs 4 --throughput-dtype none 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ring5 parity', d['cpu_baseline'].get('gpu_vs_cpu_max_prob_err'), d['parity_ok'], d['value'])"
GLICLASS_MX=0 GLC_HIP_SO=$PWD/$V/libgliclass_hip_dev.so timeout -k 10 300 python3 scripts/attn_bench.py 4,20,36 0 f32 > $OUT/attn_wg_stagger.txt 2>&1; cat $OUT/attn_wg_stagger.txt
