#!/usr/bin/env python3
"""A/B of lbh_index_add's text-slice cuts (option "add_cuts") on pre-packed pinned 1 M-claim batches, memo warm and off."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth

cuts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,20611,30711,40812,10307").split(",")]
chunk = 1_000_000
chs = [synth.make_chunk(ci * chunk, chunk, seed=1234, chunk_index=ci, real_sha_keys=False) for ci in range(3)]
pins = [[_hip.pinned_like(a) for a in (ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)] for ch in chs]
out = []
for memo in (20, 0):
    _hip.set_option("k1_memo_log2", memo)
    for rnd in range(2):
        for cut in cuts:
            _hip.set_option("add_cuts", cut)
            ix = _hip.Index()
            ix.reserve(4 * chunk)
            calls = []
            for rep in range(4):
                p = pins[rep % 3]
                t0 = time.perf_counter()
                ix.add_texts(*(x.a for x in p))
                calls.append(time.perf_counter() - t0)
            ix.close()
            out.append({"memo_log2": memo, "round": rnd, "add_cuts": cut, "call_ms": [round(1e3 * t, 3) for t in calls],
                        "claims_per_s_best": chunk / min(calls[1:])})
_hip.set_option("add_cuts", 0)
print(json.dumps(out, indent=1))
