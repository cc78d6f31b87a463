#!/usr/bin/env python3
"""Print the last N dispatches matching a pattern from a rocprofv3 kernel_trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
pat = sys.argv[2].split(",")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = [r for r in csv.DictReader(open(f)) if any(p in r["Kernel_Name"] for p in pat)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-n:]:
    print(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40],
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us  grid", r["Grid_Size_X"])
