# scan workgroup size x blocks per CU x index size (GPU box, repo root); variant library: lbm_amd/csrc/abl/liblbhip_t512.so
for n in 100000 300000 1000000 3000000 10000000; do
  for lib in lbm_amd/liblbhip.so lbm_amd/csrc/abl/liblbhip_t512.so; do
    echo "== n=$n $lib"
    LBM_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 tools/tune_scan.py $n 8 ${1:-0} | grep -E "CU=(1|2|3) nt=1" || exit 1
  done
done
