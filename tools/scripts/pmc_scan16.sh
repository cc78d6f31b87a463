# HBM traffic of the fp16 flavour of the single-query scan: two separate PMC passes (GPU box, repo root)
set -e
export LBH_SCAN_MODE=1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_s16_fetch -o s16f --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_search.py 10000000 8 10 > $GRAFT_REPO_ROOT/gpurun_out/pmc_s16_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_s16_write -o s16w --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_search.py 10000000 8 10 > $GRAFT_REPO_ROOT/gpurun_out/pmc_s16_write.log 2>&1
