#!/bin/bash
# int8 batched scan against the fp16 one (10 M x 256, top-8 and top-64) and the sample density of the int8 mode
set -e
O=gpurun_out/r3i; mkdir -p $O
run() { echo "== $*"; env "$@" timeout -k 10 240 python3 tools/bench_batched.py 10000000 256 ${K:-8} 6 | tee -a $O/sweep.txt; }
: > $O/sweep.txt
run LBH_BQ_MODE=2
for d in 3 4 6 8 12; do run LBH_BQ_MODE=3 LBH_BQ_SAMPLE_DIV_I8=$d; done
K=64 run LBH_BQ_MODE=2
K=64 run LBH_BQ_MODE=3 LBH_BQ_SAMPLE_DIV_I8=4
K=64 run LBH_BQ_MODE=3 LBH_BQ_SAMPLE_DIV_I8=8
