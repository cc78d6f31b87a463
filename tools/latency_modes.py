#!/usr/bin/env python3
"""Wall-clock latency of one single-query search under the two completion modes (spin_us = 0: block in
hipStreamSynchronize; > 0: poll the completion ticket in mapped host memory), with and without event timing.

Usage: python3 tools/latency_modes.py [--claims 1000000,10000000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth  # noqa: E402
from tools.run_configs import build_index  # noqa: E402


def measure(ix, qv, k, timing, reps=300):
    st = _hip.Stats()
    st.want_timing = timing
    for i in range(10):
        ix.search_shard(qv[i % len(qv)], k, _hip.SINCE_NONE, 0, st)
    wall = []
    for i in range(reps):
        t0 = time.perf_counter()
        ix.search_shard(qv[i % len(qv)], k, _hip.SINCE_NONE, 0, st)
        wall.append(time.perf_counter() - t0)
    w = np.array(wall) * 1e6
    return {"p50_us": float(np.percentile(w, 50)), "p90_us": float(np.percentile(w, 90)), "p99_us": float(np.percentile(w, 99)),
            "min_us": float(w.min()), "scan_us": st.scan_ms * 1e3 if timing else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--claims", default="1000000,10000000")
    args = ap.parse_args()
    qv = _hip.embed_batch(synth.make_queries(64, seed=4321))
    out = {"device": _hip.device_name()}
    for n in [int(x) for x in args.claims.split(",")]:
        ix, _ = build_index(n)
        res = {}
        for spin in (0, 5000):
            _hip.set_option("spin_us", spin)
            for timing in (0, 1, 2):
                res[f"spin_us={spin},timing={timing}"] = measure(ix, qv, 8, timing)
        _hip.set_option("scan_mode", 1)  # opt-in: scan a 128 B/claim fp16 copy, re-rank in fp64 as always
        for timing in (0, 1):
            res[f"scan_mode=1,spin_us=5000,timing={timing}"] = measure(ix, qv, 8, timing)
        _hip.set_option("scan_mode", 0)
        out[str(n)] = res
        ix.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
