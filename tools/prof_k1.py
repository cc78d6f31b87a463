#!/usr/bin/env python3
"""Fixed K1 workload for rocprofv3: chunks of the bench corpus (1 M claims each, text resident in HBM), `reps` launches
of embed_kernel per chunk through lbh_embed_batch_dev.  With the token memo on (default) launch 1 of a process fills the
table and the later ones read it.  Usage: python3 tools/prof_k1.py [claims] [reps] [chunks]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (buffer ownership only)
from lbm_amd import _hip, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
_hip.context()
e64 = torch.empty((n, 64), dtype=torch.float64, device="cuda")
e32 = torch.empty((n, 64), dtype=torch.float32, device="cuda")
for ci in range(chunks):
    ch = synth.make_chunk(ci * n, n, seed=1234, chunk_index=ci, real_sha_keys=False)
    d_blob = torch.from_numpy(np.concatenate([ch.blob, np.zeros(64, np.uint8)])).cuda()
    d_off = torch.from_numpy(ch.offsets).cuda()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _hip.embed_batch_dev(d_blob.data_ptr(), d_off.data_ptr(), n, int(ch.offsets[-1]), e64.data_ptr(), e32.data_ptr())
        ts.append(time.perf_counter() - t0)
    print("k1 chunk", ci, n, "claims", int(ch.lens.sum()), "tokens", int(ch.offsets[-1]), "text bytes; ms per launch:", [round(t * 1e3, 3) for t in ts])
