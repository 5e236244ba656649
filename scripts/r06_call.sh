OUT=gpurun_out/r06; mkdir -p $OUT
V=gliclass/c_amd/variants
bash scripts/ab_so.sh - $V/libgliclass_hip_gx_max-ilp.so $V/libgliclass_hip_gx_iterative-ilp.so > $OUT/ab_gemm_sched.txt 2>&1; cat $OUT/ab_gemm_sched.txt
