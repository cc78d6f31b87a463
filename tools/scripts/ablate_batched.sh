# timing-only builds of the batched kernel (lbm_amd/csrc/abl/liblbhip_abl<N>.so, see LBH_BQ_ABLATE in search.hip)
set -e
for a in "$@"; do
  echo "ABLATE=$a"
  LBM_HIP_LIB=$PWD/lbm_amd/csrc/abl/liblbhip_abl$a.so timeout -k 10 200 python3 tools/bench_batched.py 10000000 256 8 6 2>/dev/null | cut -c1-200
done
