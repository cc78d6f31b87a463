#!/bin/bash
# K3-only part of the round-2 profile set (GPU box, repo root): counters + kernel trace of the batched search, configs.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/bench_batched.py 10000000 256 8 2"
rm -rf $O/pmc_k3_sq $O/pmc_k3_fetch $O/pmc_k3_write $O/r02_k3_trace
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_k3_sq -o sq --output-format csv -- $B > $O/pmc_k3_sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_k3_fetch -o f --output-format csv -- $B > $O/pmc_k3_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_k3_write -o w --output-format csv -- $B > $O/pmc_k3_write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/r02_k3_trace -o k3 --output-format csv -- $B > $O/r02_k3_trace.log 2>&1
find $O -name '*_agent_info.csv' -delete
cd $R
timeout -k 10 900 python3 tools/run_configs.py > $O/r02_configs.json 2> $O/r02_configs.err
echo k3 profiles done
