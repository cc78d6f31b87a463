#!/usr/bin/env python3
"""Single-query latency / scan-time A/B on one index: for each (option, value) setting, p50 of the whole call (no device
events) and the scan kernel's HIP-event time, at top_k 8 and 64, plain and through the TEXT entry (compile_context's call).
Usage: python3 tools/lat_ab.py <claims> <option> <v1,v2,...> [reps]      e.g.  lat_ab.py 1000000 scan_fuse 0,1,2"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth  # noqa: E402

n = int(sys.argv[1])
opt = sys.argv[2]
vals = [int(v) for v in sys.argv[3].split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
gap_us = float(os.environ.get("LAT_GAP_US", "0"))  # idle time between two calls (a serving loop is not back to back)


def idle():
    if gap_us > 0:
        t_end = time.perf_counter() + gap_us * 1e-6
        while time.perf_counter() < t_end:
            pass
ix = _hip.Index()
ix.reserve(n)
done = ci = 0
while done < n:
    m = min(1_000_000, n - done)
    ch = synth.make_chunk(done, m, seed=1234, chunk_index=ci, real_sha_keys=False)
    ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    ix.retract_many(ids[ch.retracted])
    done += m
    ci += 1
queries = synth.make_queries(64, seed=4321)
qv = _hip.embed_batch(queries)
out = {"claims": n, "live": ix.size(), "option": opt, "gap_us_between_calls": gap_us, "runs": []}
ref = {}
for rnd in range(2):  # every setting twice, interleaved: box drift shows up as a spread between the rounds
    for v in vals:
        _hip.set_option(opt, v)
        rec = {"value": v, "round": rnd}
        for k in (8, 64):
            st = _hip.Stats()
            for i in range(20):
                ix.search_shard(qv[i % 64], k, _hip.SINCE_NONE, 0)
            lat = []
            for i in range(reps):
                t0 = time.perf_counter()
                r = ix.search_shard(qv[i % 64], k, _hip.SINCE_NONE, 0)
                lat.append(time.perf_counter() - t0)
                idle()
                key = (k, i % 64)
                sig = (r[0].tobytes(), r[1].tobytes())
                assert ref.setdefault(key, sig) == sig, "results differ between settings"
            tl = []
            for i in range(reps):
                t0 = time.perf_counter()
                ix.search_text(queries[i % 64], k)
                tl.append(time.perf_counter() - t0)
                idle()
            st.want_timing = 1
            scan = []
            for i in range(60):
                ix.search_shard(qv[i % 64], k, _hip.SINCE_NONE, 0, st)
                scan.append(st.scan_ms)
            rec[f"k{k}"] = {"p50_us": round(float(np.percentile(lat, 50)) * 1e6, 1), "p99_us": round(float(np.percentile(lat, 99)) * 1e6, 1),
                            "text_p50_us": round(float(np.percentile(tl, 50)) * 1e6, 1),
                            "scan_us": round(float(np.median(scan)) * 1e3, 1), "candidates": int(st.candidates)}
        out["runs"].append(rec)
print(json.dumps(out))
