#!/usr/bin/env python3
"""Wave-level timeline of one batched search (instrumented build: make -C lbm_amd/csrc ablate A=10, then
LBM_HIP_LIB=lbm_amd/csrc/abl/liblbhip_abl10.so python3 tools/scripts/k3_phases.py): for each of the three scan
launches the span from the first workgroup's start to the last one's end, and the gaps between launches."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbm_amd import _hip, synth
from tools.run_configs import build_index
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ix, _ = build_index(n)
qv = _hip.embed_batch(synth.make_queries(256, seed=4321))
st = _hip.Stats(); st.want_timing = 1
for _ in range(4):
    ix.search(qv, 8, _hip.SINCE_NONE, 0, st)
lib = ctypes.CDLL(os.environ["LBM_HIP_LIB"])
out = np.zeros((3, 256, 5), dtype=np.int64)
assert lib.lbh_debug_k3_times(out.ctypes.data_as(ctypes.c_void_p)) == 0
t0 = out[0, :, 0].min()
prev = None
print(f"event-timed scan (all launches): {st.scan_ms * 1e3:.1f} us")
for p, name in enumerate(("sample n/256", "sample n/16", "full")):
    s, e = out[p, :, 0], out[p, :, 1]
    line = f"{name:13s} first start +{(s.min() - t0) / 100:8.2f} us  start spread {(s.max() - s.min()) / 100:5.2f}  last end +{(e.max() - t0) / 100:8.2f}  wave span {(e.max() - s.min()) / 100:8.2f}  per-wg min/med/max {(e - s).min() / 100:.1f}/{np.median(e - s) / 100:.1f}/{(e - s).max() / 100:.1f}"
    if prev is not None:
        line += f"  gap since previous launch's last end {(s.min() - prev) / 100:6.2f}"
    prev = e.max()
    print(line)
    pro, loop = out[p, :, 2] - s, out[p, :, 4] - out[p, :, 3]
    print(f"              median per workgroup: prologue {np.median(pro) / 100:.1f}  loop {np.median(loop) / 100:.1f}  flush {np.median(e - out[p, :, 4]) / 100:.1f} us")
