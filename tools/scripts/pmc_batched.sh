# usage: bash tools/scripts/pmc_batched.sh <bq_mode> <tag> "<counters>"   (GPU box, repo root)
set -e
export LBH_BQ_MODE=$1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $3 -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$2 -o $2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_batched.py 10000000 256 8 2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$2.log 2>&1
