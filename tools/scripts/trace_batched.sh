# usage: bash tools/scripts/trace_batched.sh <bq_mode> <tag>   (run on the GPU box from the repo root)
set -e
export LBH_BQ_MODE=$1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$2 -o $2 -- python3 $GRAFT_REPO_ROOT/tools/bench_batched.py 10000000 256 8 6 > $GRAFT_REPO_ROOT/gpurun_out/prof_$2.log 2>&1
