#!/usr/bin/env python3
"""Config 5 at its stated size on ONE MI355X: bulk re-vectorisation of 100 M claims (82 GB of index) through the text path
(K1), 1 % of them retracted again through lbh_index_retract_many, then filtered and unfiltered queries, with parity shown
the way tests/test_gpu_fullsize.py shows it at 10 M -- sampled embeddings against the oracle bit for bit, and
size-independent properties of the ranking (prefix property across the fast / exact paths, exact scores recomputed from
the stored vectors, filters honoured, batched == single query) -- plus the timing of every leg.

Usage: python3 tools/run_100m.py [claims] > profiles/rNN_100m_single_gpu.json      (about 4 minutes, mostly host-side
text generation)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip as hip, synth  # noqa: E402
from oracle import lb_oracle as orc  # noqa: E402  (the checker; this script is a parity run, not product code)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
CHUNK = 1_000_000
rng = np.random.default_rng(2025)
out = {"device": hip.device_name(), "claims": N, "checks": {}}

ix = hip.Index()
ix.reserve(N)
t_gen = t_add = t_retract = 0.0
sampled = 0
keep = {}
n_chunks = (N + CHUNK - 1) // CHUNK
t_start = time.perf_counter()
for ci in range(n_chunks):
    m = min(CHUNK, N - ci * CHUNK)
    t0 = time.perf_counter()
    ch = synth.make_chunk(ci * CHUNK, m, seed=1234, chunk_index=ci, real_sha_keys=False)
    t1 = time.perf_counter()
    ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    t2 = time.perf_counter()
    ix.retract_many(ids[ch.retracted])
    t3 = time.perf_counter()
    t_gen += t1 - t0
    t_add += t2 - t1
    t_retract += t3 - t2
    live_idx = np.nonzero(~ch.retracted)[0]
    pick = rng.choice(live_idx, size=200, replace=False)
    pick[:2] = live_idx[[0, -1]]
    want = orc.embed_batch([ch.text(int(i)) for i in pick])
    got = ix.get_embeddings(ids[pick])
    assert got.tobytes() == want.tobytes(), f"K1 parity failed in chunk {ci}"
    sampled += len(pick)
    if ci in (0, n_chunks // 2, n_chunks - 1):
        keep[ci] = (ch.text(4242 % m), int(ids[4242 % m]), bool(ch.retracted[4242 % m]))
    if ci % 10 == 9:
        print(f"[run_100m] {ci + 1} / {n_chunks} chunks, {time.perf_counter() - t_start:.0f} s", file=sys.stderr, flush=True)
live = ix.size()
out["vectorise"] = {"chunks": n_chunks, "host_text_generation_s": t_gen, "add_texts_s": t_add, "retract_many_s": t_retract,
                    "claims_per_s_incl_h2d": N / t_add, "retracted": N - live, "live": live,
                    "index_bytes": live * 824}
out["checks"]["k1_sampled_rows_bit_exact_vs_oracle"] = sampled

base = synth.BASE_MS
p50, p90 = base + 1000 * (N // 2), base + 1000 * (9 * N // 10)
queries = synth.make_queries(16, seed=4321) + [v[0] for v in keep.values()]
qv = hip.embed_batch(queries)
assert qv.tobytes() == orc.embed_batch(queries).tobytes()
st = hip.Stats()


def timed(fn, reps=20):
    for _ in range(3):
        fn(0)
    wall, scan = [], []
    for i in range(reps):
        st.want_timing = 1
        t0 = time.perf_counter()
        fn(i)
        wall.append(time.perf_counter() - t0)
        scan.append(st.scan_ms)
    return float(np.median(wall) * 1e3), float(np.median(scan))


legs = {}
n_prop = 0
for name, since, tm in (("unfiltered", hip.SINCE_NONE, 0), ("since_p50", p50, 0), ("since_p90", p90, 0), ("tags", hip.SINCE_NONE, 0xF0F0),
                        ("since_p50_and_tags", p50, 0x00FF)):
    for mode in (0, 1):
        hip.set_option("scan_mode", mode)
        w, s = timed(lambda i: ix.search_shard(qv[i % 16], 8, since, tm, st))
        bpr = st.bytes_per_row
        legs[f"{name}/{'fp16_copy' if mode else 'fp32'}"] = {
            "p50_ms": w, "scan_us": s * 1e3, "bytes_per_claim": bpr, "scan_GBps": live * bpr / (s * 1e-3) / 1e9,
            "frac_of_8TBps": live * bpr / (s * 1e-3) / 8e12, "claims_per_s": live / (w * 1e-3), "path": st.path}
    hip.set_option("scan_mode", 0)
    # properties (every query): fast top-8 is a prefix of fast top-64 is a prefix of the paged top-200 whose pages 2.. come
    # from the exact fp64 scan; scores are the reference's dot product of the stored vectors; every result passes the filters
    for j in range(len(queries)):
        i8, s8, _ = ix.search_shard(qv[j], 8, since, tm, st)
        assert st.path == 0 and st.fallbacks == 0
        i64, s64, _ = ix.search_shard(qv[j], 64, since, tm)
        i200, s200, _ = ix.search_shard(qv[j], 200, since, tm, st)
        assert st.path == 1
        assert i64[:8].tolist() == i8.tolist() and i200[:64].tolist() == i64.tolist() and s200[:64].tobytes() == s64.tobytes()
        assert np.all(s200[:-1] >= s200[1:]) and len(i200) == 200
        vec = ix.get_embeddings(i200[:16])
        assert [orc.cosine(qv[j], vec[r]) for r in range(16)] == s200[:16].tolist()
        hip.set_option("scan_mode", 1)
        h8, hs8, _ = ix.search_shard(qv[j], 8, since, tm)
        hip.set_option("scan_mode", 0)
        assert h8.tolist() == i8.tolist() and hs8.tobytes() == s8.tobytes()
        if since != hip.SINCE_NONE:
            assert np.all(i200 >= (since - base + 999) // 1000)  # claim id i was created at base + 1000 i
        n_prop += 1
out["single_query"] = legs
out["checks"]["ranking_property_checks_passed"] = n_prop

# the claims whose own text is the query rank first (or tie with an exact duplicate of the text)
for ci, (text, cid, retracted) in keep.items():
    if retracted:
        continue
    i8, s8, _ = ix.search_shard(hip.embed_batch([text])[0], 8, hip.SINCE_NONE, 0)
    assert cid in i8[s8 == s8[0]].tolist(), ci
out["checks"]["self_queries_rank_first"] = len(keep)

# config 3 shape on the 100 M corpus: 256 batched queries == 256 single queries
q256 = hip.embed_batch(synth.make_queries(256, seed=99))
w, s = timed(lambda i: ix.search(q256, 8, hip.SINCE_NONE, 0, st), reps=5)
bi, bs, bc = ix.search(q256, 8, hip.SINCE_NONE, 0, st)
fb = st.fallbacks
for j in range(256):
    wi, ws, _ = ix.search_shard(q256[j], 8, hip.SINCE_NONE, 0)
    assert bi[j, : int(bc[j])].tolist() == wi.tolist() and bs[j, : int(bc[j])].tobytes() == ws.tobytes(), j
out["batched_256"] = {"wall_ms": w, "scan_ms": s, "pairs_per_s": live * 256 / (w * 1e-3), "fallbacks": fb, "bytes_per_claim": st.bytes_per_row,
                      "equal_to_single_query_for_all_256": True}
wf, sf = timed(lambda i: ix.search(q256, 8, p50, 0x0F0F, st), reps=5)
bi, bs, bc = ix.search(q256, 8, p50, 0x0F0F, st)
for j in range(0, 256, 5):
    wi, ws, _ = ix.search_shard(q256[j], 8, p50, 0x0F0F)
    assert bi[j, : int(bc[j])].tolist() == wi.tolist() and bs[j, : int(bc[j])].tobytes() == ws.tobytes(), j
out["batched_256_since_p50_and_tags"] = {"wall_ms": wf, "scan_ms": sf}

# N3 on the same corpus
ids, created, matched = ix.recent(p90, 64, st)
assert matched > 0 and np.all(created[:-1] >= created[1:]) and np.all(created >= p90)
out["recent_64_since_p90"] = {"matched": matched}
print(json.dumps(out, indent=1))
