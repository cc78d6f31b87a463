# wall / kernel time of K3 build variants (timing-only switches give wrong results: numbers only)
for v in "$@"; do echo "== $v"; LBM_HIP_LIB=$PWD/lbm_amd/csrc/abl/liblbhip_k3_$v.so timeout -k 10 200 python3 tools/bench_batched.py 10000000 256 8 10 | cut -c1-170 || exit 1; done
