OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 10 600 python3 scripts/parity_report.py > $OUT/parity_report.txt 2>&1; cat $OUT/parity_report.txt | tail -20
