#!/bin/bash
# After `gpurun -- bash tools/scripts/r03_profiles.sh` (here, repo root): turn gpurun_out/r03prof into the records kept under profiles/.
set -e
O=gpurun_out/r03prof; P=profiles
cp $O/r03_bench.json $P/r03_bench_10m.json
cp $O/r03_bench_rocprof.json $P/r03_bench_10m_under_rocprof.json
cp $O/r03_stats/bench_kernel_stats.csv $P/r03_bench_10m_kernel_stats.csv
cp $O/r03_configs.json $P/r03_configs.json
SRC="tools/scripts/r03_profiles.sh: two rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE) -- python3 tools/prof_search.py 10000000 8 10"
python3 tools/pmc_summary.py scan $O/pmc_scan_fetch/f_counter_collection.csv $O/pmc_scan_write/w_counter_collection.csv scan_f32_kernel 10000000 256 "$SRC" > $P/r03_pmc_scan.json
MS=$(python3 -c "import re,statistics,sys; m=re.findall(r'ms per launch: \[([^\]]*)\]', open('$O/pmc_k1.log').read()); print(statistics.median(float(x) for x in m[-1].split(',')) if m else 2.4)")
python3 tools/k1_pmc_summary.py $O/pmc_k1/k1_counter_collection.csv 1000000 ${MS:-2.4} "gpurun_out/r03prof: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/prof_k1.py 1000000 5 (token memo on: launch 1 fills the table, the median launch reads it)" > $P/r03_pmc_k1.json
python3 tools/k3_pmc_summary.py $O > $P/r03_pmc_batched.json
python3 tools/summarize_trace.py $O/r03_k3_trace/k3_kernel_trace.csv > $P/r03_k3_kernel_trace.txt
echo collected; ls -la $P | grep r03_ | wc -l
