#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection CSVs) into the JSON kept under profiles/.

  python3 tools/pmc_summary.py scan  <fetch.csv> <write.csv> <kernel> <rows> <algorithmic bytes per row> > profiles/rNN_pmc_scan.json
  python3 tools/pmc_summary.py launches <counters.csv> [<more.csv> ...] <kernel substring>     (per-launch counter table)

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half
of a wide coalesced read, hence x2 (calibrated in round 1 on a kernel that reads 512 MB and writes 256 MB)."""
import collections
import csv
import json
import statistics
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


def per_kernel(path):
    """{kernel: {counter: [value per launch]}} in launch order"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def scan(fetch_csv, write_csv, kernel, rows, bytes_per_row, source=""):
    f, w = per_kernel(fetch_csv), per_kernel(write_csv)
    res = {"source": source,
           "corrections": "counter unit = KiB; gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read -> x2",
           "FETCH_SIZE": {k: {"launches": len(v["FETCH_SIZE"]), "median_raw": statistics.median(v["FETCH_SIZE"])} for k, v in f.items()},
           "WRITE_SIZE": {k: {"launches": len(v["WRITE_SIZE"]), "median_raw": statistics.median(v["WRITE_SIZE"])} for k, v in w.items()}}
    rd = res["FETCH_SIZE"][kernel]["median_raw"] * 1024 * 2
    wr = res["WRITE_SIZE"][kernel]["median_raw"] * 1024
    alg = rows * bytes_per_row
    res[kernel] = {"rows": rows, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                   "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_row": (rd + wr) / rows,
                   "traffic_over_algorithmic": (rd + wr) / alg}
    return res


def launches(csvs, kernel_sub):
    table = collections.defaultdict(dict)
    for p in csvs:
        for k, ctrs in per_kernel(p).items():
            if kernel_sub not in k:
                continue
            for c, vals in ctrs.items():
                for i, v in enumerate(vals):
                    table[(k, i)][c] = v
    return [{"kernel": k, "launch": i, **v} for (k, i), v in sorted(table.items(), key=lambda kv: (kv[0][1], kv[0][0]))]


if __name__ == "__main__":
    if sys.argv[1] == "scan":
        src = sys.argv[7] if len(sys.argv) > 7 else ""
        print(json.dumps(scan(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), src), indent=1))
    else:
        print(json.dumps(launches(sys.argv[2:-1], sys.argv[-1]), indent=1))
