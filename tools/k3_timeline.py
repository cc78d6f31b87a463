#!/usr/bin/env python3
"""Timeline of the last batched search in a rocprofv3 kernel trace of tools/bench_batched.py: start / end / gap per kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = [i for i, r in enumerate(rows) if "prep_bfrag" in r["Kernel_Name"]][-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev = None
for r in rows[i0:]:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-42:]
    st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{nm:42s} start {st:8.1f} end {en:8.1f} dur {en - st:7.1f} gap {(st - prev) if prev is not None else 0:6.1f}")
    prev = en
