# A/B of tile hand-out variants of the single-query scan (instrumented builds): bash tools/scripts/k2_dynamic_ab.sh "<suffixes>" [claims]
set -e
R=$GRAFT_REPO_ROOT
n=${2:-1000000}
for v in $1; do
  echo "== $v"
  K2_DUMP=$R/gpurun_out/k2_blocks_$v.npy LBM_HIP_LIB=$R/lbm_amd/csrc/abl/liblbhip_k2t$v.so timeout -k 10 120 python3 tools/scripts/k2_phases.py $n
done
echo "== tune 1M"; timeout -k 10 200 python3 tools/tune_scan.py 1000000 8 0 | grep -E "blocks/CU=(2|3|4) "
echo "== tune 10M"; timeout -k 10 300 python3 tools/tune_scan.py 10000000 8 0 | grep -E "blocks/CU=(2|3|4) "
echo "== tune 10M fp16"; timeout -k 10 300 python3 tools/tune_scan.py 10000000 8 1 | grep -E "blocks/CU=(2|3|4) "
echo "== parity"; timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py tests/test_gpu_concurrency.py -x -q --timeout 300 --timeout-method thread -m gpu | tail -3
