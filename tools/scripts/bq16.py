#!/usr/bin/env python3
"""Batched fp16 scan: operands from the fp32 matrix (converted while staging) vs from the fp16 copy (scan_mode 1)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbm_amd import _hip, synth
from tools.run_configs import build_index
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ix, _ = build_index(n)
qv = _hip.embed_batch(synth.make_queries(256, seed=4321))
st = _hip.Stats(); st.want_timing = 1
out = {}
ref = None
for mode in (0, 1, 0, 1):
    _hip.set_option("scan_mode", mode)
    for _ in range(3):
        r = ix.search(qv, 8, _hip.SINCE_NONE, 0, st)
    ts, ws = [], []
    for _ in range(15):
        t0 = time.perf_counter(); r = ix.search(qv, 8, _hip.SINCE_NONE, 0, st); ws.append(time.perf_counter() - t0); ts.append(st.scan_ms)
    if ref is None:
        ref = r
    same = ref[0].tobytes() == r[0].tobytes() and ref[1].tobytes() == r[1].tobytes()
    out.setdefault(f"scan_mode={mode}", []).append({"scan_ms_median": float(np.median(ts)), "wall_ms_median": float(np.median(ws) * 1e3),
        "bytes_per_row": st.bytes_per_row, "fallbacks": st.fallbacks, "max_score_err": st.max_score_err, "same_as_first": same})
print(json.dumps(out, indent=1))
