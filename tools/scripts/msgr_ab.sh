#!/bin/bash
# (Record of a dropped experiment: the LBH_MSGR_* switches lived in the messenger-wave patch of search.hip, which was not
# kept -- see DESIGN.md, K2, item (g).)
# Which part of the messenger-wave build slows the streaming loop?  _a: messenger leaves at once; _b: streaming waves skip
# their LDS publication (messenger then leaves when wave 0 ends); _c: both.  One box.
for v in "" _a _b _c; do
  L=lbm_amd/csrc/abl/liblbhip_k2t$v.so
  echo "variant '$v': $(LBM_HIP_LIB=$L timeout -k 10 100 python3 tools/tune_scan.py 1000000 8 0 | grep -E 'CU=1 nt=1')"
done
