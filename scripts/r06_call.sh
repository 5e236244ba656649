OUT=gpurun_out/r06; mkdir -p $OUT
V=gliclass/c_amd/variants
GLC_HIP_SO=$PWD/$V/libgliclass_hip_ring5b.so timeout -k 10 600 python3 -m pytest tests/test_gpu_mx.py -q -x -k "epilogue or gemm" 2>&1 | tail -3
bash scripts/ab_so.sh - $V/libgliclass_hip_ring5b.so $V/libgliclass_hip_ring5.so > $OUT/ab_ring5b.txt 2>&1; cat $OUT/ab_ring5b.txt
GLC_HIP_SO=$PWD/$V/libgliclass_hip_ring5b.so timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --cpu-seqs 4 --throughput-dtype none 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ring5b parity', d['cpu_baseline'].get('gpu_vs_cpu_max_prob_err'), d['parity_ok'], d['value'])"
