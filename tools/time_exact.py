#!/usr/bin/env python3
"""Latency of the paths behind large top_k and mass ties on a big index: top_k = 64 (one fast page), 65..200 (fast
page + exact pages with a cursor) and the empty query (every score 0.0: exact path, pure hash order)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth  # noqa: E402
from tools.run_configs import build_index  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ix, _ = build_index(n)
qv = _hip.embed_batch(synth.make_queries(16, seed=4321))
st = _hip.Stats()
st.want_timing = 1
out = {"claims": n}
for name, q, k in [("k64", qv[0], 64), ("k100", qv[0], 100), ("k200", qv[1], 200), ("k1000", qv[2], 1000),
                   ("empty_query_k8", np.zeros(64), 8), ("empty_query_k100", np.zeros(64), 100)]:
    ix.search_shard(q, k, _hip.SINCE_NONE, 0, st)
    w = []
    for r in range(5):
        t0 = time.perf_counter()
        ix.search_shard(q, k, _hip.SINCE_NONE, 0, st)
        w.append(time.perf_counter() - t0)
    out[name] = {"top_k": k, "wall_ms": float(np.median(w) * 1e3), "scan_ms_total": st.scan_ms, "launches": st.scan_launches,
                 "path_of_last_page": st.path, "fallbacks": st.fallbacks,
                 "exact_scan_GBps": (n * 512 / (st.scan_ms / st.scan_launches * 1e-3) / 1e9) if st.path == 1 else None}
print(json.dumps(out, indent=1))
