#!/bin/bash
# Runs on the GPU box (via gpurun): the round's measurement set -> gpurun_out/final_<tag>/ (copied into profiles/ by hand)
TAG=${1:-r02}
R=$(pwd)
O=$R/gpurun_out/final_$TAG
mkdir -p $O
python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python3 bench.py --config small --batch 8 --seq 512 --steps 50 --warmup 5 --cpu-seqs 8 --cpu-seqs-8 0 --throughput-dtype none > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python3 bench.py --config large --batch 32 --seq 1024 --steps 5 --warmup 2 --cpu-seqs 0 --throughput-dtype none > $O/bench_c4shard.json 2> $O/bench_c4.err; echo "c4 rc=$?"
python3 bench.py --config qwen-1.5b --batch 16 --seq 2048 --steps 3 --warmup 1 --cpu-seqs 0 --throughput-dtype none > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
python3 scripts/attn_bench.py 4 1 f32 > $O/attn_stamps.txt 2>&1; echo "attn rc=$?"
python3 scripts/gemm_gs_bench.py > $O/gemm_gs_bench.txt 2>&1; echo "gemm rc=$?"
for f in c3 c2 c4shard c5; do tail -1 $O/bench_$f.json | cut -c1-230; done
