#!/usr/bin/env python3
"""lbh_index_add from pageable numpy arrays (1 M-claim text batches) against the number of staging threads."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth
chs = [synth.make_chunk(ci * 1_000_000, 1_000_000, seed=1234, chunk_index=ci, real_sha_keys=False) for ci in range(2)]
out = []
for rnd in range(2):
    for nh in (1, 2, 3, 4, 6, 8):
        _hip.set_option("stage_helpers", nh)
        ix = _hip.Index(); ix.reserve(5_000_000)
        calls = []
        for rep in range(5):
            ch = chs[rep % 2]
            t0 = time.perf_counter(); ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys); calls.append(time.perf_counter() - t0)
        ix.close()
        out.append({"round": rnd, "stage_helpers": nh, "call_ms": [round(1e3 * t, 2) for t in calls]})
        print(out[-1], flush=True)
_hip.set_option("stage_helpers", 2)
