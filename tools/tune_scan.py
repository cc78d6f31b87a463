#!/usr/bin/env python3
"""A/B the scan kernel's launch knobs in ONE process (interleaved rounds, median + min).
Usage: python tools/tune_scan.py [claims] [top_k] [scan_mode]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bpr = 128 if mode else 256
rng = np.random.default_rng(0)
ix = _hip.Index(); ix.reserve(n)
chunk = 1_000_000
for s in range(0, n, chunk):
    m = min(chunk, n - s)
    v = rng.standard_normal((m, 64)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    ix.add_vectors(v, np.arange(s, s + m, dtype=np.int64), None, rng.integers(0, 256, size=(m, 32), dtype=np.uint8))
qs = rng.standard_normal((16, 64)); qs /= np.linalg.norm(qs, axis=1, keepdims=True)
st = _hip.Stats(); st.want_timing = 2
_hip.set_option("scan_mode", mode)
configs = [(b, nt) for b in (1, 2, 3, 4, 5, 6, 8) for nt in (0, 1)]
res = {c: [] for c in configs}; tot = {c: [] for c in configs}
for rnd in range(6):
    for c in configs:
        _hip.set_option("scan_blocks_per_cu", c[0]); _hip.set_option("scan_nt", c[1])
        for i in range(8):
            ix.search_shard(qs[i % 16], k, _hip.SINCE_NONE, 0, st)
            if rnd > 0:
                res[c].append(st.scan_ms * 1e3); tot[c].append(st.total_ms * 1e3)
for c in configs:
    a = np.array(res[c]); t = np.array(tot[c])
    print(f"blocks/CU={c[0]} nt={c[1]}: scan median {np.median(a):8.1f} us  min {a.min():8.1f} us  -> {n*bpr/np.median(a)/1e3:7.1f} GB/s | total median {np.median(t):8.1f} us")
