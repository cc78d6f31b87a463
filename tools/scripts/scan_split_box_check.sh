#!/bin/bash
# Do the hard-coded tile shares of the two-blocks-per-CU scan (1127 : 1016 : 977 : 880) still win on THIS box?  Same
# process layout, alternating: built-in shares / equal shares / grid-stride loop, 10 M claims, top-8.
set -e
O=gpurun_out/r3s; mkdir -p $O
id=$(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1); echo "box: $(hostname) gpu ${id:-unknown}"
for rep in 1 2; do
  for sp in builtin 1000,1000,1000,1000 0; do
    if [ $sp = builtin ]; then unset LBH_SCAN_SPLIT; else export LBH_SCAN_SPLIT=$sp; fi
    timeout -k 10 300 python3 tools/lat_ab.py 10000000 scan_fuse 1 > $O/tmp.json
    python3 - "$sp" <<PY
import json, sys
r = json.load(open("$O/tmp.json"))["runs"][-1]["k8"]
print("shares %-22s scan %.1f us  p50 %.1f us" % (sys.argv[1], r["scan_us"], r["p50_us"]))
PY
  done
done
