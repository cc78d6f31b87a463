# K3 sample-pass size sweep: LBH_BQ_SAMPLE_DIV = the sample is 1/div of the tiles (GPU box, repo root)
set -e
for d in ${1:-6 10 14 20 32}; do
  echo "== div $d"
  LBH_BQ_SAMPLE_DIV=$d timeout -k 10 200 python3 tools/bench_batched.py 10000000 256 8 10 | cut -c1-330
done
