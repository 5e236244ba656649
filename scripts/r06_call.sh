OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -q -x -m gpu 2>&1 | tail -6
