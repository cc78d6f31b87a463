#!/usr/bin/env python3
"""Three 1 M-claim lbh_index_add calls on pre-packed pinned buffers, for a rocprofv3 --kernel-trace --memory-copy-trace run
(tools/scripts/ingest_trace.sh): where the time of one ingest call goes (copies vs K1 vs host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ix = _hip.Index()
ix.reserve(4 * chunk)
for ci in range(4):
    ch = synth.make_chunk(ci * chunk, chunk, seed=1234, chunk_index=ci, real_sha_keys=False)
    pins = [_hip.pinned_like(a) for a in (ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)]
    t0 = time.perf_counter()
    ix.add_texts(*(p.a for p in pins))
    dt = time.perf_counter() - t0
    print(f"call {ci}: {dt * 1e3:.2f} ms = {chunk / dt / 1e6:.1f} M claims/s, text {len(ch.blob) / chunk:.0f} B/claim", flush=True)
    for p in pins:
        p.close()
ix.close()

# the bench's own loop shape: ordinary numpy arrays, 1 % of each chunk retracted right after
import numpy as np
ix = _hip.Index()
ix.reserve(4 * chunk)
for ci in range(4):
    ch = synth.make_chunk(ci * chunk, chunk, seed=1234, chunk_index=ci, real_sha_keys=False)
    t0 = time.perf_counter()
    ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    t1 = time.perf_counter()
    dead = ids[ch.retracted]
    t2 = time.perf_counter()
    ix.retract_many(dead)
    t3 = time.perf_counter()
    print(f"pageable call {ci}: add {1e3 * (t1 - t0):.2f} ms, pick ids {1e3 * (t2 - t1):.2f} ms, retract_many({len(dead)}) {1e3 * (t3 - t2):.2f} ms", flush=True)
ix.close()
