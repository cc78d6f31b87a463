#!/usr/bin/env python3
"""Searches on ONE index from several threads (reader/writer lock + per-search slots in liblbhip).
Usage: python3 tools/concurrent_search.py [claims] [queries per thread]  -> JSON: queries/s with 1, 2 and 4 threads."""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(1)
ix = _hip.Index(); ix.reserve(n)
for a in range(0, n, 1_000_000):
    m = min(1_000_000, n - a)
    v = rng.standard_normal((m, 64)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    ix.add_vectors(v, np.arange(a, a + m, dtype=np.int64), None, rng.integers(0, 256, size=(m, 32), dtype=np.uint8))
qs = rng.standard_normal((64, 64)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
want = [ix.search_shard(qs[j], 8)[0].tolist() for j in range(64)]
out = {"claims": n, "top_k": 8, "queries_per_thread": reps, "device": _hip.device_name()}
for T in (1, 2, 4):
    bad = []
    def work(t):
        for r in range(reps):
            j = (r + 7 * t) % 64
            if ix.search_shard(qs[j], 8)[0].tolist() != want[j]:
                bad.append((t, r))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    assert not bad, bad[:3]
    out["threads_%d" % T] = {"queries_per_s": T * reps / dt, "us_per_query_per_thread": dt / reps * 1e6}
print(json.dumps(out))
