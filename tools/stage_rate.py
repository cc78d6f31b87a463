import sys, time, threading, numpy as np
sys.path.insert(0, "/root/repo")
from lbm_amd import _hip
n = 168_000_000
src = np.random.default_rng(0).integers(0, 255, size=n, dtype=np.uint8)
pin = _hip.PinnedArray(n, np.uint8)
for nt in (1, 2, 4, 8):
    best = 1e9
    for rep in range(4):
        parts = np.array_split(np.arange(n), nt)
        bounds = [(int(p[0]), int(p[-1]) + 1) for p in parts]
        def work(a, b):
            pin.a[a:b] = src[a:b]
        ts = [threading.Thread(target=work, args=b) for b in bounds]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        best = min(best, time.perf_counter() - t0)
    print("threads", nt, "host memcpy into pinned: %.1f GB/s (%.2f ms for 168 MB)" % (n / best / 1e9, best * 1e3))
