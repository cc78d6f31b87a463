set -e
for a in 1 2 3; do
  echo "ABLATE=$a"
  LBM_HIP_LIB=$PWD/lbm_amd/csrc/abl/liblbhip_abl$a.so timeout -k 10 200 python3 tools/bench_batched.py 10000000 256 8 6 2>/dev/null | cut -c1-260
done
echo "FULL + kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -o bf16 -- python3 $GRAFT_REPO_ROOT/tools/bench_batched.py 10000000 256 8 6 > $GRAFT_REPO_ROOT/gpurun_out/prof_bf16.log 2>&1
