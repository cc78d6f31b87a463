import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from lbm_amd import _hip, synth
n = 2_000_000
ix = _hip.Index(); ix.reserve(n)
for ci in range(n // 1_000_000):
    ch = synth.make_chunk(ci * 1_000_000, 1_000_000, seed=1234, chunk_index=ci, real_sha_keys=False)
    ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
qv = _hip.embed_batch(synth.make_queries(256, seed=4321))
print("max |q_i|", np.abs(qv).max(), "mean l1", np.abs(qv).sum(1).mean())
st = _hip.Stats()
for mode in (2, 3):
    _hip.set_option("bq_mode", mode)
    ids, sc, cnt = ix.search(qv, 8, _hip.SINCE_NONE, 0, st)
    print("mode", mode, "margin", st.margin, "max_err", st.max_score_err, "fallbacks", st.fallbacks, "candidates", st.candidates, "bytes", st.bytes_per_row)
    k8 = sc[:, 7]
    print("  8th best score: min %.3f median %.3f max %.3f" % (k8.min(), np.median(k8), k8.max()))
