#!/usr/bin/env python3
"""lbh_index_add on 1 M-claim batches from pinned buffers and from ordinary (pageable) numpy arrays, texts and vectors."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip
from tools.run_configs import ingest_prepacked

out = {"texts": ingest_prepacked()}
rng = np.random.default_rng(0)
v = rng.standard_normal((1_000_000, 64)); v /= np.linalg.norm(v, axis=1, keepdims=True)
keys = rng.integers(0, 256, size=(1_000_000, 32), dtype=np.uint8)
created = np.arange(1_000_000, dtype=np.int64)
ix = _hip.Index(); ix.reserve(4_000_000)
calls = []
for rep in range(4):
    t0 = time.perf_counter(); ix.add_vectors(v, created, None, keys); calls.append(time.perf_counter() - t0)
out["vectors_pageable_call_ms"] = [round(1e3 * t, 3) for t in calls]
out["vectors_note"] = "1 M x 64 fp64 vectors = 512 MB + 40 MB of metadata per call; the link alone needs 9.7 ms"
ix.close()
print(json.dumps(out, indent=1))
