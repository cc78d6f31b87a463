#!/bin/bash
# Round 3, first GPU call: the GPU test suite on the new build, a short bench, the K1 counter pass.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=8 > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -12 $O/gputests.log
timeout -k 10 400 python3 bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-1800 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_k1 -o k1 --output-format csv -- python3 $R/tools/prof_k1.py 1000000 5 > $O/pmc_k1.log 2>&1 || { tail -5 $O/pmc_k1.log; exit 1; }
tail -2 $O/pmc_k1.log
find $O -name '*_agent_info.csv' -delete
echo step1 done
