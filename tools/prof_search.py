#!/usr/bin/env python3
"""Small fixed workload for rocprofv3: N random unit vectors, `reps` single-query searches.
Usage: python3 tools/prof_search.py [claims] [top_k] [reps]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rng = np.random.default_rng(0)
ix = _hip.Index(); ix.reserve(n)
for s in range(0, n, 1_000_000):
    m = min(1_000_000, n - s)
    v = rng.standard_normal((m, 64)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    ix.add_vectors(v, np.arange(s, s + m, dtype=np.int64), None, rng.integers(0, 256, size=(m, 32), dtype=np.uint8))
qs = rng.standard_normal((16, 64)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
for i in range(reps):
    ix.search_shard(qs[i % 16], k, _hip.SINCE_NONE, 0, None)
print("done", n, k, reps)
