# K3 sample-pass sweep: bash tools/scripts/sweep_shifts.sh "9,5" "8,4" ...   (GPU box, repo root)
set -e
for sh in "$@"; do
  echo "SHIFTS=$sh"
  LBH_BQ_SAMPLE_SHIFTS="$sh" timeout -k 10 200 python3 tools/bench_batched.py 10000000 256 8 8 2>/dev/null | cut -c1-220
done
