#!/bin/bash
# Quick look at a scan change on one box: GPU tests, block + select timelines (instrumented build), scan timings.
set -e
timeout -k 10 700 python -u -m pytest tests -x -q --timeout 400 --timeout-method thread -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
T=lbm_amd/csrc/abl/liblbhip_k2t.so
LBM_HIP_LIB=$T LBM_AMD_LIB=$T timeout -k 10 200 python3 tools/scripts/k2_phases.py 1000000 > gpurun_out/k2_phases_1m.txt
LBM_HIP_LIB=$T LBM_AMD_LIB=$T timeout -k 10 200 python3 tools/scripts/k2_phases.py 10000000 > gpurun_out/k2_phases_10m.txt
timeout -k 10 100 python3 tools/tune_scan.py 1000000 8 0 | grep -E "CU=1 nt=1"
timeout -k 10 100 python3 tools/tune_scan.py 1000000 64 0 | grep -E "CU=1 nt=1"
timeout -k 10 100 python3 tools/tune_scan.py 10000000 8 0 | grep -E "CU=2 nt=1"
