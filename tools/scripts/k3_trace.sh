# kernel trace of the batched search (GPU box, repo root): bash tools/scripts/k3_trace.sh [queries] [top_k]
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/k3_trace
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/k3_trace -o k3 --output-format csv -- python3 $R/tools/bench_batched.py 10000000 ${1:-256} ${2:-8} 3 > $O/k3_trace.log 2>&1
tail -1 $O/k3_trace.log | cut -c1-300
