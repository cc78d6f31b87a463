#!/usr/bin/env python3
"""Print the last N kernel dispatches (name, µs, grid) of a rocprofv3 results .db (rocpd sqlite output)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(db.execute("select name, start, end, grid_x, vgpr_count, scratch_size from kernels order by start"))
for r in rows[-n:]:
    print(f"{r[0].replace('(anonymous namespace)::', '')[:64]:64s} {(r[2] - r[1]) / 1e3:10.1f} us  grid {r[3]}  vgpr {r[4]} scratch {r[5]}")
