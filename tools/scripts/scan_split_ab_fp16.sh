#!/bin/bash
# Same-box A/B of the tile split for the fp16 flavour of the scan (scan_mode 1) and, beside it, the fp32 one.
for mode in 1 0; do for n in ${SIZES:-10000000 1000000}; do for rep in 1 2; do for w in 0 1000,1000,1000,1000 default; do
  if [ $w = default ]; then unset LBH_SCAN_SPLIT; else export LBH_SCAN_SPLIT=$w; fi
  echo "mode $mode claims $n rep $rep split $w: $(timeout -k 10 100 python3 tools/tune_scan.py $n 8 $mode | grep -E 'CU=[12] nt=1' | tr '\n' ' ')"
done; done; done; done
