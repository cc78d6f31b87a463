#!/bin/bash
# A/B of the single-query scan's tile split on ONE box: grid-stride against contiguous ranges with equal and weighted shares.
# Needs the instrumented build (make -C lbm_amd/csrc k2timing) for the timelines.  Run from the repo root on the GPU box.
set -e
T=lbm_amd/csrc/abl/liblbhip_k2t.so
for n in ${SIZES:-1000000 10000000}; do
  for w in ${SPLITS:-0 1000,1000,1000,1000 1127,1016,977,880}; do
    echo "=== claims $n split $w"
    LBH_SCAN_SPLIT=$w timeout -k 10 200 python3 tools/tune_scan.py $n 8 0 | grep -E "nt=1"
    LBH_SCAN_SPLIT=$w LBM_HIP_LIB=$T LBM_AMD_LIB=$T timeout -k 10 200 python3 tools/scripts/k2_phases.py $n | grep -E "loop done|written  |blockIdx|select"
  done
done
