#!/bin/bash
# one rocprofv3 --pmc pass over tools/bench_batched.py (10 M x 256, 2 reps); usage: pmc_pass.sh <tag> "<counters>" [bench args]
tag=$1; ctr=$2; shift 2
args=${@:-10000000 256 8 2}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctr -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $tag --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_batched.py $args > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
rc=$?
tail -n 2 $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log | cut -c1-300
exit $rc
