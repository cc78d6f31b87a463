# kernel trace of K1 on 3 x 1 M synthetic claims (GPU box, repo root): bash tools/scripts/k1_trace.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -o $1 -- python3 $GRAFT_REPO_ROOT/tools/run_build_only.py 3000000 > $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log 2>&1
