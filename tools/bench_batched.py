#!/usr/bin/env python3
"""Config 3: N claims x Q batched queries on one MI355X (K3, MFMA path).
Usage: python3 tools/bench_batched.py [claims] [queries] [top_k] [reps]"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ix = _hip.Index(); ix.reserve(n)
done = 0; ci = 0
while done < n:
    m = min(1_000_000, n - done)
    ch = synth.make_chunk(done, m, seed=1234, chunk_index=ci, real_sha_keys=False)
    ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    done += m; ci += 1
qv = _hip.embed_batch(synth.make_queries(nq, seed=4321))
st = _hip.Stats(); st.want_timing = 1
ix.search(qv, k, _hip.SINCE_NONE, 0, st)
plain = _hip.Stats()  # wall time is taken on calls WITHOUT device events (they add a gap in front of the first scan kernel)
wall, wall_timed, scan, fb = [], [], [], 0
for r in range(reps):
    t0 = time.perf_counter()
    ix.search(qv, k, _hip.SINCE_NONE, 0, plain)
    wall.append(time.perf_counter() - t0); fb += plain.fallbacks
    t0 = time.perf_counter()
    ix.search(qv, k, _hip.SINCE_NONE, 0, st)
    wall_timed.append(time.perf_counter() - t0); scan.append(st.scan_ms / 1e3); fb += st.fallbacks
w, wt, s = float(np.median(wall)), float(np.median(wall_timed)), float(np.median(scan))
flops = 2.0 * 64 * nq * n
bpc = st.bytes_per_row
print(json.dumps({"claims": n, "queries": nq, "top_k": k, "path": st.path, "fallbacks": fb,
                  "wall_ms": w * 1e3, "wall_ms_with_event_timing": wt * 1e3, "scan_ms": s * 1e3, "pairs_per_s": n * nq / w,
                  "scan_tflops": flops / s / 1e12, "frac_of_dense_f16_mfma_peak_2500": flops / s / 2.5e15,
                  "bytes_per_claim": bpc, "scan_GBps": n * bpc / s / 1e9}))
