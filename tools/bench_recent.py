#!/usr/bin/env python3
"""the recency query (get_recent_claims) on a large index.

Reports wall-clock latency and the scan kernel's HBM rate (algorithmic bytes: 8 B/claim, the created_ms
column) for a time-ordered corpus, and the CPU restatement (numpy, one core) on the same arrays.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip  # noqa: E402


def numpy_recent(created, order, since, limit):
    """The same answer on one host core with numpy (timing reference only; parity is tests/test_recent.py)."""
    idx = np.nonzero(created >= since)[0]
    by_order = idx[np.argsort(order[idx], kind="stable")][::-1]
    asc = by_order[np.argsort(created[by_order], kind="stable")]
    return asc[::-1][:limit]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--claims", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=100)
    args = ap.parse_args()
    n = args.claims
    rng = np.random.default_rng(5)
    ix = _hip.Index()
    ix.reserve(n)
    created = np.empty(n, dtype=np.int64)
    for s in range(0, n, 1_000_000):
        m = min(1_000_000, n - s)
        v = rng.standard_normal((m, 64))
        # publish order = time order, several claims per millisecond (ties) and some jitter
        ms = 1_700_000_000_000 + (np.arange(s, s + m) // 3) + rng.integers(0, 5, size=m)
        created[s:s + m] = ms
        ix.add_vectors(v, ms, np.zeros(m, dtype=np.uint64), rng.integers(0, 256, size=(m, 32), dtype=np.uint8))
    order = np.arange(n, dtype=np.uint64)
    st = _hip.Stats()
    out = {"device": _hip.device_name(), "claims": n, "algorithmic_bytes_per_claim": 8}
    p90 = int(np.percentile(created, 90))
    for name, since, limit in [("limit100", _hip.SINCE_NONE, 100), ("limit100_since_p90", p90, 100),
                               ("limit1000", _hip.SINCE_NONE, 1000), ("limit64", _hip.SINCE_NONE, 64)]:
        ids, cr, matched = ix.recent(since, limit)
        exp = numpy_recent(created, order, since, limit)
        assert ids.tolist() == exp.tolist() and cr.tolist() == created[exp].tolist()
        wall, scan = [], []
        for r in range(args.reps):
            st.want_timing = 1 if r % 4 == 0 else 0
            t0 = time.perf_counter()
            ix.recent(since, limit, st)
            wall.append(time.perf_counter() - t0)
            if r % 4 == 0:
                scan.append(st.scan_ms * 1e-3 / st.scan_launches)
        t0 = time.perf_counter()
        numpy_recent(created, order, since, limit)
        t_cpu = time.perf_counter() - t0
        s = float(np.median(scan))
        out[name] = {"limit": limit, "matched": matched, "passes": st.scan_launches,
                     "p50_ms": float(np.percentile(wall, 50) * 1e3), "p99_ms": float(np.percentile(wall, 99) * 1e3),
                     "scan_us_per_pass": s * 1e6, "scan_GBps": n * 8 / s / 1e9, "frac_of_8TBps": n * 8 / s / 8e12,
                     "cpu_numpy_1core_ms": t_cpu * 1e3}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
