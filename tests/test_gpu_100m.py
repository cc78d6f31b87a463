"""BASELINE.json configs 4 and 5 at their STATED size on the one GPU of the box: 100 M synthetic claims through the TEXT
path (K1), held twice in HBM -- once as ONE index (82 GB) and once as the EIGHT 12.5 M-claim shards that eight ranks would
hold (another 82 GB; 288 GB fits both).  That is config 4's exact partition minus the xGMI hop, and config 5 (bulk
re-vectorisation + filtered queries) whole.  The oracle cannot score 100 M claims in test time, so parity rests on

  * K1: 20 000 sampled rows (200 per chunk, chunk edges included) bit-equal to the oracle's embed of the same text;
  * for 16 queries x k in {8, 64} x {no filter, since = P50, tag mask}: the merge (lbh_merge_topk, the reference's
    comparator, lb/context_graph.py:120) of the eight shards' local top-k lists (lbh_index_search_shard) == the single
    index's answer == the exact fp64 path's answer (option force_exact) -- ids, fp64 score bits, hash keys;
  * the k winners' scores are the oracle's sequential fp64 dot (lb/latent.py:63) of the stored vectors;
  * 256 batched queries (K3, one pass over the 100 M-claim matrix) == the 256 single queries.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import lb_oracle as orc

pytestmark = pytest.mark.gpu

N = 100_000_000
CHUNK = 1_000_000
SHARDS = 8
PER_SHARD = N // SHARDS
WORKERS = 14  # generation threads (numpy releases the GIL for the bulk of a chunk); the box gives the job 16 cores


@pytest.fixture(scope="module")
def corpus():
    from lbm_amd import _hip as hip, synth
    one = hip.Index()
    one.reserve(N)
    shards = [hip.Index() for _ in range(SHARDS)]
    for s in shards:
        s.reserve(PER_SHARD)
    rng = np.random.default_rng(100)
    sample_ids, sample_texts = [], []
    n_chunks = N // CHUNK
    gate = threading.Semaphore(WORKERS + 6)  # chunks generated but not yet loaded: bounds the host memory they hold

    def gen(ci):
        gate.acquire()
        return synth.make_chunk(ci * CHUNK, CHUNK, seed=1234, chunk_index=ci, real_sha_keys=False)

    with ThreadPoolExecutor(WORKERS) as ex:
        futs = [ex.submit(gen, ci) for ci in range(n_chunks)]
        for ci, f in enumerate(futs):
            ch = f.result()
            futs[ci] = None
            ids = one.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
            assert int(ids[0]) == ci * CHUNK and int(ids[-1]) == (ci + 1) * CHUNK - 1  # claim id == position in the corpus
            one.retract_many(ids[ch.retracted])
            # the same claims into the shard(s) this chunk belongs to (12.5 M per shard: four chunks straddle a boundary)
            lo = ci * CHUNK
            while lo < (ci + 1) * CHUNK:
                s = lo // PER_SHARD
                hi = min((ci + 1) * CHUNK, (s + 1) * PER_SHARD)
                a, b = lo - ci * CHUNK, hi - ci * CHUNK
                lids = shards[s].add_texts(ch.blob, ch.offsets[a: b + 1], ch.created_ms[a:b], ch.tagmask[a:b], ch.keys[a:b])
                assert int(lids[0]) == lo - s * PER_SHARD
                shards[s].retract_many(lids[ch.retracted[a:b]])
                lo = hi
            livepos = np.nonzero(~ch.retracted)[0]
            pick = rng.choice(livepos, size=200, replace=False)
            pick[:4] = livepos[[0, 1, -2, -1]]  # chunk edges: first / last K1 groups of a launch
            sample_ids.append(ids[pick])
            sample_texts.extend(ch.text(int(i)) for i in pick)
            del ch
            gate.release()
    yield {"hip": hip, "one": one, "shards": shards, "sample_ids": np.concatenate(sample_ids), "sample_texts": sample_texts}
    one.close()
    for s in shards:
        s.close()


def test_k1_sampled_rows_equal_the_oracle_100m(corpus):
    hip, one, shards = corpus["hip"], corpus["one"], corpus["shards"]
    live = one.size()
    assert N * 0.9895 < live < N * 0.9905  # 1 % retracted
    assert sum(s.size() for s in shards) == live
    ids = corpus["sample_ids"]
    assert ids.shape[0] == 20_000
    want = orc.embed_batch(corpus["sample_texts"])
    assert one.get_embeddings(ids).tobytes() == want.tobytes()
    # the shards' copies of the same rows (local id = global id - shard base)
    for s in range(SHARDS):
        m = (ids // PER_SHARD) == s
        assert shards[s].get_embeddings(ids[m] - np.uint64(s * PER_SHARD)).tobytes() == want[m].tobytes(), s


def _merged(hip, shards, q, k, since, tm):
    from lbm_amd.sharded import pack_local
    packs = []
    for s, ix in enumerate(shards):
        li, ls, lk = ix.search_shard(q, k, since, tm)
        packs.append(pack_local(ls, li + np.uint64(s * PER_SHARD), lk, k))
    e = hip.merge_topk(np.stack(packs), k)
    return e[:, 1].copy().view(np.uint64), e[:, 0].copy().view(np.float64), np.ascontiguousarray(e[:, 2:6]).view(np.uint64)


def test_eight_shard_merge_equals_single_index_equals_exact_path_100m(corpus):
    from lbm_amd import synth
    hip, one, shards = corpus["hip"], corpus["one"], corpus["shards"]
    queries = synth.make_queries(14, seed=4321) + [corpus["sample_texts"][3], corpus["sample_texts"][-7]]
    qv = hip.embed_batch(queries)
    assert qv.tobytes() == orc.embed_batch(queries).tobytes()
    p50 = synth.BASE_MS + 1000 * (N // 2)
    st = hip.Stats()
    cases = 0
    for j, q in enumerate(qv):
        for k in (8, 64):
            for since, tm in ((hip.SINCE_NONE, 0), (p50, 0), (hip.SINCE_NONE, 0x00FF00FF00FF00FF)):
                wi, ws, wk = one.search_shard(q, k, since, tm, st)
                assert st.path == 0 and st.rows_scanned == one.size() and len(wi) == k
                gi, gs, gkw = _merged(hip, shards, q, k, since, tm)
                assert gi.tolist() == wi.tolist(), (j, k, since, tm)
                assert gs.tobytes() == ws.tobytes()
                assert gkw.astype(">u8").view(np.uint8).reshape(-1, 32).tobytes() == wk.tobytes()
                hip.set_option("force_exact", 1)
                try:
                    xi, xs, xk = one.search_shard(q, k, since, tm, st)
                finally:
                    hip.set_option("force_exact", 0)
                assert st.path == 1 and st.bytes_per_row >= 512
                assert xi.tolist() == wi.tolist() and xs.tobytes() == ws.tobytes() and xk.tobytes() == wk.tobytes(), (j, k, since, tm)
                assert np.all(ws[:-1] >= ws[1:])
                if since != hip.SINCE_NONE:
                    assert np.all(wi >= N // 2)  # created_ms = BASE + 1000 * claim id in this corpus
                cases += 1
        # the winners' scores are the reference's sequential fp64 dot of the stored vectors
        wi, ws, _ = one.search_shard(q, 8)
        vecs = one.get_embeddings(wi)
        assert [orc.cosine(q, vecs[r]) for r in range(8)] == ws.tolist()
    assert cases == 96
    # the claims whose own text is the query come first (or tie with an exact duplicate of that text)
    for t, cid in ((corpus["sample_texts"][3], corpus["sample_ids"][3]), (corpus["sample_texts"][-7], corpus["sample_ids"][-7])):
        i8, s8, _ = one.search_shard(hip.embed_batch([t])[0], 8)
        assert int(cid) in i8[s8 == s8[0]].tolist()


def test_256_batched_queries_equal_single_queries_100m(corpus):
    from lbm_amd import synth
    hip, one = corpus["hip"], corpus["one"]
    qv = hip.embed_batch(synth.make_queries(256, seed=99))
    st = hip.Stats()
    p50 = synth.BASE_MS + 1000 * (N // 2)
    for k, since, tm, stride in ((8, hip.SINCE_NONE, 0, 1), (16, p50, 0xF0F0, 8)):
        bi, bs, bc = one.search(qv, k, since, tm, st)
        assert st.path == 2 and st.rows_scanned == one.size()
        for j in range(0, 256, stride):
            wi, ws, _ = one.search_shard(qv[j], k, since, tm)
            assert bi[j, : int(bc[j])].tolist() == wi.tolist() and bs[j, : int(bc[j])].tobytes() == ws.tobytes(), (k, j)
