#!/bin/bash
# Round-2 profile set (GPU box, repo root): bench line, kernel stats of the same command, HBM traffic (separate PMC
# passes) of the single-query scans, matrix-pipe and traffic counters of the batched scan.  Each step has its own limit;
# the first failure ends the script.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "== $*"; }
step bench;  timeout -k 10 400 python3 $R/bench.py > $O/r02_bench.json 2> $O/r02_bench.err; tail -c 600 $O/r02_bench.json
step stats;  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/r02_stats -o bench --output-format csv -- python3 $R/bench.py > $O/r02_bench_rocprof.json 2> $O/r02_bench_rocprof.err
P="python3 $R/tools/prof_search.py 10000000 8 10"
step fetch;  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_scan_fetch -o f --output-format csv -- $P > $O/pmc_scan_fetch.log 2>&1
step write;  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_scan_write -o w --output-format csv -- $P > $O/pmc_scan_write.log 2>&1
export LBH_SCAN_MODE=1
step fetch16; timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_scan16_fetch -o f --output-format csv -- $P > $O/pmc_scan16_fetch.log 2>&1
step write16; timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_scan16_write -o w --output-format csv -- $P > $O/pmc_scan16_write.log 2>&1
unset LBH_SCAN_MODE
B="python3 $R/tools/bench_batched.py 10000000 256 8 2"
step k3sq;   timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_k3_sq -o sq --output-format csv -- $B > $O/pmc_k3_sq.log 2>&1
step k3fetch; timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_k3_fetch -o f --output-format csv -- $B > $O/pmc_k3_fetch.log 2>&1
step k3write; timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_k3_write -o w --output-format csv -- $B > $O/pmc_k3_write.log 2>&1
step k3trace; timeout -k 10 200 rocprofv3 --kernel-trace -d $O/r02_k3_trace -o k3 --output-format csv -- $B > $O/r02_k3_trace.log 2>&1
find $O -name '*_agent_info.csv' -delete
echo profiles done
cd $R
echo "== configs"; timeout -k 10 900 python3 tools/run_configs.py > $O/r02_configs.json 2> $O/r02_configs.err
echo "== sharded overhead"; timeout -k 10 300 python3 tools/sharded_overhead.py > $O/r02_sharded_overhead.json 2> $O/r02_sharded_overhead.err
echo "== concurrent"; timeout -k 10 200 python3 tools/concurrent_search.py 1000000 2000 > $O/r02_concurrent_1m.json 2> $O/r02_concurrent_1m.err
echo all done
