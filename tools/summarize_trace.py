#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace CSV: per-kernel calls / mean / min duration and mean gap to the previous kernel."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(list); gaps = collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[name].append(en - st)
    if prev_end is not None: gaps[name].append(st - prev_end)
    prev_end = en
for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    g = gaps.get(name, [0])
    d2 = sorted(d)
    print(f"{name[:70]:70s} calls {len(d):6d} mean {sum(d)/len(d)/1e3:9.2f} us  median {d2[len(d2)//2]/1e3:9.2f} us  min {d2[0]/1e3:9.2f} us  gap-before median {sorted(g)[len(g)//2]/1e3:7.2f} us")
