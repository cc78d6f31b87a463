#!/bin/bash
# Same-box A/B of the two-blocks-per-CU tile shares at mid sizes (4-8 M claims): grid-stride, equal ranges, the default weights.
for n in ${SIZES:-4000000 6000000 8000000}; do for w in 0 1000,1000,1000,1000 default 1060,1010,990,940; do
  if [ $w = default ]; then unset LBH_SCAN_SPLIT; else export LBH_SCAN_SPLIT=$w; fi
  echo "claims $n split $w: $(timeout -k 10 150 python3 tools/tune_scan.py $n 8 0 | grep -E 'CU=2 nt=1')"
done; done
