timeout -k 10 900 python3 -m pytest tests/test_gpu_mx.py -q -x -k "resident" 2>&1 | tail -3
