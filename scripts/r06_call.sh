OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests/test_gpu_fullsize.py -q -x -s 2>&1 | grep -E "c[2345] |passed|failed|rows vs|shard" | tee $OUT/fullsize_values.txt | tail -40
