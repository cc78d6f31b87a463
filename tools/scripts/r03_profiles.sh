#!/bin/bash
# Round-3 profile set (GPU box, repo root): bench line, kernel stats of the same command, HBM traffic (separate PMC
# passes) of the single-query scan, K1 instruction counters, matrix-pipe counters + kernel trace of the batched scan,
# the per-config record.  Each step has its own limit; the first failure ends the script.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $*"; }
step bench;  timeout -k 10 400 python3 $R/bench.py > $O/r03_bench.json 2> $O/r03_bench.err; cut -c1-400 $O/r03_bench.json
step stats;  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/r03_stats -o bench --output-format csv -- python3 $R/bench.py > $O/r03_bench_rocprof.json 2> $O/r03_bench_rocprof.err
P="python3 $R/tools/prof_search.py 10000000 8 10"
step fetch;  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_scan_fetch -o f --output-format csv -- $P > $O/pmc_scan_fetch.log 2>&1
step write;  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_scan_write -o w --output-format csv -- $P > $O/pmc_scan_write.log 2>&1
step k1pmc;  timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_k1 -o k1 --output-format csv -- python3 $R/tools/prof_k1.py 1000000 5 > $O/pmc_k1.log 2>&1
B="python3 $R/tools/bench_batched.py 10000000 256 8 2"
step k3sq;   timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_k3_sq -o sq --output-format csv -- $B > $O/pmc_k3_sq.log 2>&1
step k3fetch; timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_k3_fetch -o f --output-format csv -- $B > $O/pmc_k3_fetch.log 2>&1
step k3write; timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_k3_write -o w --output-format csv -- $B > $O/pmc_k3_write.log 2>&1
step k3trace; timeout -k 10 200 rocprofv3 --kernel-trace -d $O/r03_k3_trace -o k3 --output-format csv -- $B > $O/r03_k3_trace.log 2>&1
find $O -name '*_agent_info.csv' -delete
echo profiles done
cd $R
echo "== configs"; timeout -k 10 900 python3 tools/run_configs.py > $O/r03_configs.json 2> $O/r03_configs.err
echo all done
