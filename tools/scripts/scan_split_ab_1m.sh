#!/bin/bash
# Same-box A/B of the scan's tile split at small sizes: grid-stride ("0") against the default contiguous ranges.
for n in ${SIZES:-100000 300000 1000000 2000000}; do for rep in 1 2; do for w in 0 default; do
  if [ $w = default ]; then unset LBH_SCAN_SPLIT; else export LBH_SCAN_SPLIT=$w; fi
  echo "claims $n rep $rep split $w: $(timeout -k 10 100 python3 tools/tune_scan.py $n 8 0 | grep -E 'CU=1 nt=1')"
done; done; done
