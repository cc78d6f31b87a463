#!/usr/bin/env python3
"""Where a claim's time goes inside K1 (instrumented build of the library, see LBH_K1_TIMING in embed.hip):
   cd lbm_amd/csrc && hipcc $(CXXFLAGS) -DLBH_K1_TIMING -c embed.hip -o abl/embed_t.o && hipcc -shared ... -o abl/liblbhip_k1t.so
   LBM_HIP_LIB=.../liblbhip_k1t.so python3 tools/scripts/k1_phases.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbm_amd import _hip, synth
lib = ctypes.CDLL(os.environ["LBM_HIP_LIB"])
out = np.zeros(8, dtype=np.uint64)
ch = synth.make_chunk(0, 1_000_000, seed=1234, chunk_index=0, real_sha_keys=False)
_hip.embed_batch_packed(ch.blob[: int(ch.offsets[1000])], ch.offsets[:1001])  # warm-up
lib.lbh_debug_k1_times(None, 1)
import time
t0 = time.perf_counter(); _hip.embed_batch_packed(ch.blob, ch.offsets); t1 = time.perf_counter()
lib.lbh_debug_k1_times(out.ctypes.data_as(ctypes.c_void_p), 0)
n = int(out[4])
print(f"claims {n}; hash batches {int(out[5])}; embed_batch incl. copies {1e3 * (t1 - t0):.1f} ms")
print(f"per claim and wave (us): up to normalisation {out[0] / n / 100:.2f}  of which hashing {out[1] / n / 100:.2f}, accumulation {out[2] / n / 100:.2f}, "
      f"classifier + loads {(int(out[0]) - int(out[1]) - int(out[2])) / n / 100:.2f};  normalise + store {out[3] / n / 100:.2f}")
print(f"hash batches on the one-block fast path {int(out[6])}, served whole by the token memo {int(out[7])}")
for rep in range(2):  # the same text again: the memo is warm now
    lib.lbh_debug_k1_times(None, 1)
    t0 = time.perf_counter(); _hip.embed_batch_packed(ch.blob, ch.offsets); t1 = time.perf_counter()
    lib.lbh_debug_k1_times(out.ctypes.data_as(ctypes.c_void_p), 0)
    n = int(out[4])
    print(f"again: hash batches {int(out[5])}, fast path {int(out[6])}, from the memo {int(out[7])}; per claim and wave (us): hashing {out[1] / n / 100:.2f}, "
          f"accumulation {out[2] / n / 100:.2f}, rest {(int(out[0]) - int(out[1]) - int(out[2])) / n / 100:.2f}, normalise + store {out[3] / n / 100:.2f}; call {1e3 * (t1 - t0):.1f} ms")
