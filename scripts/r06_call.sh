OUT=gpurun_out/r06; mkdir -p $OUT
V=gliclass/c_amd/variants
bash scripts/ab_so.sh - $V/libgliclass_hip_w128.so $V/libgliclass_hip_w128a1.so $V/libgliclass_hip_w128a2.so $V/libgliclass_hip_w128a4.so $V/libgliclass_hip_w128a7.so > $OUT/ab_w128_abl.txt 2>&1; cat $OUT/ab_w128_abl.txt
