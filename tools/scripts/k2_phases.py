#!/usr/bin/env python3
"""Timeline of the single-query scan's blocks (instrumented build of the library, LBH_K2_TIMING in search.hip):
   make -C lbm_amd/csrc k2timing  ->  abl/liblbhip_k2t.so;   LBM_HIP_LIB=.../liblbhip_k2t.so python3 tools/scripts/k2_phases.py [claims]
Stamps per block (first wave): 0 start, 1 main loop done, 2 final wave compaction done, 3 block barrier passed, 4 merged list
sorted + shared cut applied, 5 candidates written."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbm_amd import _hip
lib = ctypes.CDLL(os.environ["LBM_HIP_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
TOPK = int(os.environ.get("K2_TOPK", "8"))
rng = np.random.default_rng(0)
ix = _hip.Index()
for s in range(0, n, 1_000_000):
    m = min(1_000_000, n - s)
    v = rng.standard_normal((m, 64)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    ix.add_vectors(v, np.arange(s, s + m, dtype=np.int64), None, rng.integers(0, 256, size=(m, 32), dtype=np.uint8))
q = rng.standard_normal(64); q /= np.linalg.norm(q)
st = _hip.Stats(); st.want_timing = 1
for _ in range(5):
    ix.search_shard(q, TOPK, _hip.SINCE_NONE, 0, st)
out = np.zeros((2048, 6), dtype=np.int64)
lib.lbh_debug_k2_times(out.ctypes.data_as(ctypes.c_void_p))
g = int(os.environ.get("K2_BLOCKS", "0")) or min(256 if n <= 3_500_000 else 512, (n + 255) // 256)
t = out[:g].astype(np.float64) / 100.0  # us
t0 = t[:, 0].min()
names = ["start", "loop done", "wave compact", "barrier", "sorted+cut", "written"]
print(f"claims {n}, blocks {g}, scan event time {st.scan_ms * 1e3:.1f} us")
for i, nm in enumerate(names):
    c = t[:, i] - t0
    print(f"  {nm:14s} first {c.min():7.2f}  median {np.median(c):7.2f}  last {c.max():7.2f} us after the first block's start")
d = np.diff(t, axis=1)
if os.environ.get("K2_DUMP"):
    np.save(os.environ["K2_DUMP"], t - t0)
print("  per-block phase medians (us):", {names[i + 1]: round(float(np.median(d[:, i])), 2) for i in range(5)})
# Are the late blocks the same ones every time?  Loop-end offsets of each block over repeated searches: their correlation
# from run to run and their mean by blockIdx % 8 (the XCD a block is dispatched to).
runs = []
for _ in range(8):
    ix.search_shard(q, TOPK, _hip.SINCE_NONE, 0, st)
    lib.lbh_debug_k2_times(out.ctypes.data_as(ctypes.c_void_p))
    r = out[:g].astype(np.float64) / 100.0
    runs.append(r[:, 1] - r[:, 0].min())
runs = np.array(runs)
cc = np.corrcoef(runs)
print(f"  loop-end offsets over 8 searches: run-to-run correlation median {np.median(cc[np.triu_indices(8, 1)]):.2f}")
mean = runs.mean(axis=0)
print("  mean loop end by blockIdx % 8 (us):", [round(float(mean[x::8].mean()), 1) for x in range(8)])
print("  mean loop end by blockIdx // 32 (us):", [round(float(mean[x * 32:(x + 1) * 32].mean()), 1) for x in range(g // 32)])
late = np.argsort(mean)[-16:]
print("  16 latest blocks on average:", sorted(int(b) for b in late), "their mean", round(float(mean[late].mean()), 1),
      "all-block mean", round(float(mean.mean()), 1), "max of per-run last", round(float(runs.max(axis=1).mean()), 1))
# the select kernel behind it (same clock): phases of its thread 0, relative to the scan's first block start and to its own start
k4 = np.zeros(16, dtype=np.int64)
if hasattr(lib, "lbh_debug_k4_times"):
    acc = []
    for _ in range(8):
        ix.search_shard(q, TOPK, _hip.SINCE_NONE, 0, st)
        lib.lbh_debug_k2_times(out.ctypes.data_as(ctypes.c_void_p))
        lib.lbh_debug_k4_times(k4.ctypes.data_as(ctypes.c_void_p))
        r = out[:g].astype(np.float64) / 100.0
        k = k4.astype(np.float64) / 100.0
        acc.append(np.concatenate([[r[:, 5].max() - r[:, 0].min(), k[0] - r[:, 0].min()], k[[1, 2, 3, 4, 6, 7]] - k[0]]))
    m = np.median(np.array(acc), axis=0)
    print("  select kernel (medians of 8): last scan block written %.2f, select starts %.2f us after the scan's start;" % (m[0], m[1]))
    print("    then +us: counters read %.2f, candidates cut %.2f, exact scores %.2f, sorted %.2f, result written %.2f, published %.2f"
          % tuple(m[2:]))
    if k4[11]:
        print("    fused tail, after the ticket: loads back +%.2f us, cuts known +%.2f us" % ((k4[11] - k4[0]) / 100.0, (k4[12] - k4[0]) / 100.0))
        if k4[13]:
            print("    the k-th-largest bound computed twice more through one copy of the code: %.2f us (code fetched), %.2f us (cached)" % ((k4[13] - k4[12]) / 100.0, (k4[14] - k4[13]) / 100.0))
    print("    candidates in the flat list E = %d, kept for the exact re-rank M = %d (last search); staged by a fused scan's last block after its pre-cut: %d" % (k4[8], k4[9], k4[10]))
