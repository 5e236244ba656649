#!/bin/bash
# Developer probe (GPU): do two half-batches on two streams (two processes here) beat one full batch?  The projections run at the chip's power
# envelope, the attention does not: interleaving them might use the headroom.  One process B=64, one process B=32, then two B=32 processes at once.
R=$(pwd)
B="python3 $R/bench.py --cpu-seqs 0 --throughput-dtype none --no-profile --steps 30 --warmup 5"
one() { $B --batch $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], 'seq/s', d['ms_per_step'], 'ms')"; }
one 64 "single B=64:"
one 32 "single B=32:"
one 32 "concurrent A B=32:" & one 32 "concurrent B B=32:" & wait
one 64 "concurrent A B=64:" & one 64 "concurrent B B=64:" & wait
