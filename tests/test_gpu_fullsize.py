"""Full-size (BASELINE.json scale) GPU checks through size-independent properties -- the oracle cannot
score 10 M claims x hundreds of queries in test time, so here parity is shown by invariants:

  * exact-path == fast-path: the fp64 scan under the full (score, key) order must return the same ids and
    score bits as the fp32 scan + re-rank (two independent kernels);
  * batched (MFMA) path == single-query path, query by query;
  * order: scores non-increasing, equal scores ordered by ascending key; scores are the exact fp64 dot of
    the stored vectors (recomputed on the host for the k winners);
  * prefix property: top-8 is a prefix of top-64 is a prefix of the paged top-200;
  * since filter: every result passes it, and results equal an unfiltered search over a corpus that only
    holds the passing rows (checked on the shard split below);
  * shard property: splitting the corpus over two indexes and merging their local top-k (the multi-GPU
    algorithm, lbm_amd/sharded.py) equals the single-index answer;
  * idempotence: repeating a search returns identical bits.
A 1 M-claim slice of the same corpus is additionally checked against the oracle directly.
"""
import numpy as np
import pytest

from oracle import lb_oracle as orc

pytestmark = pytest.mark.gpu

N_FULL = 10_000_000


@pytest.fixture(scope="module")
def corpus():
    from lbm_amd import _hip as hip
    rng = np.random.default_rng(2024)
    ix = hip.Index()
    ix.reserve(N_FULL)
    half = [hip.Index(), hip.Index()]  # the same rows dealt round-robin over two shards
    keep_vec, keep_key, keep_ms = [], [], []
    for s in range(0, N_FULL, 1_000_000):
        m = 1_000_000
        # clustered vectors (a few hundred centres + noise): realistic near-ties, plus exact duplicates
        centres = rng.standard_normal((256, 64))
        v = centres[rng.integers(0, 256, size=m)] + 0.35 * rng.standard_normal((m, 64))
        v[1::1000] = v[0::1000][: v[1::1000].shape[0]]  # 0.1 % exact duplicates -> key tie-breaks
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        ms = np.arange(s, s + m, dtype=np.int64) * 1000 + 1_700_000_000_000
        keys = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
        tags = (np.uint64(1) << rng.integers(0, 64, size=m).astype(np.uint64)).astype(np.uint64)
        ids = ix.add_vectors(v, ms, tags, keys)
        assert int(ids[0]) == s
        for r in (0, 1):
            half[r].add_vectors(v[r::2], ms[r::2], tags[r::2], keys[r::2])
        if s == 0:
            keep_vec, keep_key, keep_ms, keep_tags = v.copy(), keys.copy(), ms.copy(), tags.copy()
    q = rng.standard_normal((256, 64))  # a full K3 launch: all eight query groups of every workgroup are in use
    q[:8] = keep_vec[[5, 17, 1000, 1001, 4242, 99_999, 500_000, 999_999]] + 0.05 * rng.standard_normal((8, 64))
    q[40:168] = centres[rng.integers(0, 256, size=128)] + 0.2 * rng.standard_normal((128, 64))  # near the last chunk's clusters
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    yield {"hip": hip, "ix": ix, "half": half, "q": q, "vec0": keep_vec, "key0": keep_key, "ms0": keep_ms, "tags0": keep_tags}
    ix.close()
    for h in half:
        h.close()


def _check_order(hip, ix, q, ids, sc):
    assert np.all(sc[:-1] >= sc[1:])
    vecs = np.stack([ix.get_embedding(int(i)) for i in ids])
    for j in range(len(ids)):  # exact fp64 dot in the reference's order (lb/latent.py:63)
        assert orc.cosine(q, vecs[j]) == sc[j]


def test_fast_vs_exact_vs_batched_10m(corpus):
    hip, ix, q = corpus["hip"], corpus["ix"], corpus["q"]
    assert ix.size() == N_FULL
    st = hip.Stats()
    # K3 in its three scan precisions (fp32 MFMA, split bf16, fp16): identical answers, measured scan error inside
    # the bound the candidate margin assumes
    batched = {}
    for mode, margin in ((0, 1.0e-5), (1, 1.0e-4), (2, 2.2e-3)):
        hip.set_option("bq_mode", mode)
        batched[mode] = ix.search(q, 8, hip.SINCE_NONE, 0, st)
        assert st.path == 2 and 0.0 < st.max_score_err < margin / 2, (mode, st.max_score_err)
        assert st.fallbacks == 0 or mode == 2, (mode, st.fallbacks)  # the wide fp16 margin may hand a query over
    for mode in (0, 1):
        assert batched[mode][0].tolist() == batched[2][0].tolist() and batched[mode][1].tobytes() == batched[2][1].tobytes()
    bi, bs, bc = batched[2]
    assert q.shape[0] == 256
    for j in range(q.shape[0]):
        i8, s8, _ = ix.search_shard(q[j], 8, hip.SINCE_NONE, 0, st)
        assert st.path == 0 and st.fallbacks == 0
        assert bi[j, : int(bc[j])].tolist() == i8.tolist() and bs[j, : int(bc[j])].tobytes() == s8.tobytes()
        if j >= 40:
            continue
        again, s_again, _ = ix.search_shard(q[j], 8, hip.SINCE_NONE, 0)
        assert again.tolist() == i8.tolist() and s_again.tobytes() == s8.tobytes()  # idempotent
        hip.set_option("scan_mode", 1)  # the fp16 copy of the matrix as scan operand: same answer, half the bytes
        h8, hs8, _ = ix.search_shard(q[j], 8, hip.SINCE_NONE, 0, st)
        hip.set_option("scan_mode", 0)
        assert st.path == 0 and st.bytes_per_row == 128 and st.max_score_err < 1.1e-3
        assert h8.tolist() == i8.tolist() and hs8.tobytes() == s8.tobytes()
        if j < 12:
            i64, s64, _ = ix.search_shard(q[j], 64, hip.SINCE_NONE, 0)
            i200, s200, _ = ix.search_shard(q[j], 200, hip.SINCE_NONE, 0, st)  # pages 2.. use the exact path
            assert st.path == 1
            assert i64[:8].tolist() == i8.tolist() and i200[:64].tolist() == i64.tolist()
            assert s200[:64].tobytes() == s64.tobytes() and len(i200) == 200
            _check_order(hip, ix, q[j], i64, s64)


def test_batched_256_filters_and_k16_10m(corpus):
    """Config 3 at full width with the fused masks: 256 queries x 10 M claims, since / tag filters, top-16 .. top-64; every
    query equals the single-query path (which test_first_million_against_oracle ties to the oracle)."""
    hip, ix, q = corpus["hip"], corpus["ix"], corpus["q"]
    st = hip.Stats()
    p50 = 1_700_000_000_000 + 1000 * (N_FULL // 2)
    for k, since, tm in ((16, hip.SINCE_NONE, 0), (8, p50, 0), (8, hip.SINCE_NONE, 0xF0F0), (16, p50, 0x0F0F), (64, hip.SINCE_NONE, 0),
                         (40, p50, 0xFF)):  # k > 16: served by the fp16 kernel only (the default mode)
        bi, bs, bc = ix.search(q, k, since, tm, st)
        assert st.path == 2 and st.bytes_per_row in (128 + 8 * (since != hip.SINCE_NONE) + 8 * (tm != 0),
                                                      256 + 8 * (since != hip.SINCE_NONE) + 8 * (tm != 0))
        for j in range(0, 256, 3):
            wi, ws, _ = ix.search_shard(q[j], k, since, tm)
            assert bi[j, : int(bc[j])].tolist() == wi.tolist() and bs[j, : int(bc[j])].tobytes() == ws.tobytes(), (k, since, tm, j)


def test_exact_path_equals_fast_path_10m(corpus):
    """An all-zero query makes every score 0.0 -> the exact path orders 10 M claims purely by key."""
    hip, ix = corpus["hip"], corpus["ix"]
    st = hip.Stats()
    ids, sc, _ = ix.search_shard(np.zeros(64), 64, hip.SINCE_NONE, 0, st)
    assert st.path == 1 and st.fallbacks == 1 and np.all(sc == 0.0) and len(ids) == 64
    k0 = corpus["key0"]
    # the winners' keys must be ascending; and no claim of the first million has a smaller key than the 64th winner
    keys = []
    for i in ids.tolist():
        if i < k0.shape[0]:
            keys.append(bytes(k0[i]))
    assert keys == sorted(keys)
    if keys:
        assert sum(bytes(k0[i]) < keys[-1] for i in range(0, k0.shape[0], 997)) <= 64


def test_shard_merge_equals_single_index_10m(corpus):
    from lbm_amd.sharded import merge_gathered, pack_local
    hip, ix, half, q = corpus["hip"], corpus["ix"], corpus["half"], corpus["q"]
    p50 = 1_700_000_000_000 + 1000 * (N_FULL // 2)
    for j in range(10):
        for k, since, tm in ((8, hip.SINCE_NONE, 0), (64, p50, 0), (8, hip.SINCE_NONE, 0xF0F0)):
            want_i, want_s, _ = ix.search_shard(q[j], k, since, tm)
            packs = []
            for r in (0, 1):
                li, ls, lk = half[r].search_shard(q[j], k, since, tm)
                # local id -> global id: within each 1 M chunk rows were dealt alternately
                gl = (li // 500_000) * 1_000_000 + (li % 500_000) * 2 + r
                packs.append(pack_local(ls, gl.astype(np.uint64), lk, k))
            gi, gs, _ = merge_gathered(np.stack(packs), k)
            assert gi.tolist() == want_i.tolist(), (j, k, since, tm)
            assert gs.tobytes() == want_s.tobytes()
            if since != hip.SINCE_NONE:
                assert np.all(want_i >= N_FULL // 2)


def test_first_million_against_oracle(corpus):
    """Direct oracle check on a 1 M-claim index holding the first chunk of the same corpus."""
    hip, q = corpus["hip"], corpus["q"]
    v, keys, ms, tags = corpus["vec0"], corpus["key0"], corpus["ms0"], corpus["tags0"]
    ix = hip.Index()
    ix.add_vectors(v, ms, tags, keys)
    packed = orc.pack_texts([bytes(k).hex() for k in keys])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(16) as ex:  # the oracle on 16 host threads: 256 queries x 1 M claims
        want = list(ex.map(lambda j: orc.compile_arrays(v, ms, None, None, q[j], 8, None, packed_keys=packed), range(256)))
    for mode in (2, 1, 0):  # every K3 precision, all 256 queries in ONE search call
        hip.set_option("bq_mode", mode)
        bi, bs, bc = ix.search(q, 8, hip.SINCE_NONE, 0)
        for j in range(256):
            want_i, want_s = want[j]
            assert bi[j, : int(bc[j])].tolist() == want_i.tolist() and bs[j, : int(bc[j])].tobytes() == want_s.tobytes(), (mode, j)
    hip.set_option("bq_mode", 2)
    for j in range(12):
        want_i, want_s = want[j]
        got_i, got_s, _ = ix.search_shard(q[j], 8, hip.SINCE_NONE, 0)
        assert got_i.tolist() == want_i.tolist() and got_s.tobytes() == want_s.tobytes()
    want_i, want_s = orc.compile_arrays(v, ms, None, None, q[0], 64, int(ms[600_000]), tags, 0xFF, packed_keys=packed)
    got_i, got_s, _ = ix.search_shard(q[0], 64, int(ms[600_000]), 0xFF)
    assert got_i.tolist() == want_i.tolist() and got_s.tobytes() == want_s.tobytes()
    ix.close()


def test_text_path_10m_k1_sampled_against_oracle():
    """Config 5 at bench scale through the TEXT path: 10 M synthetic claims are vectorised by K1 (lbh_index_add with
    texts, exactly what bench.py does), 1 % leave again through lbh_index_retract_many, and 2 000 random rows per chunk
    (20 000 in all) are read back and compared bit for bit with the oracle's embedding of the same text.  Then the
    ranking over that corpus is tied to the oracle through the sample: the k winners' scores are the oracle's dot
    products of the oracle's embeddings."""
    from lbm_amd import _hip as hip, synth
    rng = np.random.default_rng(77)
    ix = hip.Index()
    ix.reserve(N_FULL)
    kept = {}
    for ci in range(N_FULL // 1_000_000):
        ch = synth.make_chunk(ci * 1_000_000, 1_000_000, seed=1234, chunk_index=ci, real_sha_keys=False)
        ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
        assert int(ids[0]) == ci * 1_000_000 and int(ids[-1]) == (ci + 1) * 1_000_000 - 1
        ix.retract_many(ids[ch.retracted])
        pick = rng.choice(np.nonzero(~ch.retracted)[0], size=2000, replace=False)
        pick[:4] = np.nonzero(~ch.retracted)[0][[0, 1, -2, -1]]  # chunk edges: first / last K1 groups of the launch
        want = orc.embed_batch([ch.text(int(i)) for i in pick])
        got = ix.get_embeddings(ids[pick])
        assert got.tobytes() == want.tobytes(), ci
        if ci in (0, 9):
            kept[ci] = (ch, ids)
    live = ix.size()
    assert N_FULL * 0.985 < live < N_FULL * 0.995
    st = hip.Stats()
    queries = synth.make_queries(6, seed=4321) + [kept[0][0].text(12345), kept[9][0].text(999_000)]
    qv = hip.embed_batch(queries)
    assert qv.tobytes() == orc.embed_batch(queries).tobytes()
    for j, q in enumerate(qv):
        i8, s8, _ = ix.search_shard(q, 8, hip.SINCE_NONE, 0, st)
        assert st.path == 0 and st.rows_scanned == live
        i200, s200, _ = ix.search_shard(q, 200, hip.SINCE_NONE, 0)  # pages 2.. come from the exact fp64 scan
        assert i200[:8].tolist() == i8.tolist() and s200[:8].tobytes() == s8.tobytes()
        vecs = ix.get_embeddings(i8)
        assert [orc.cosine(q, vecs[r]) for r in range(8)] == s8.tolist()
    # the two claims whose own text is the query must come first (or tie with an exact duplicate of that text)
    for ci, local in ((0, 12345), (9, 999_000)):
        ch, ids = kept[ci]
        if ch.retracted[local]:
            continue
        q = hip.embed_batch([ch.text(local)])[0]
        i8, s8, _ = ix.search_shard(q, 8, hip.SINCE_NONE, 0)
        top = i8[s8 == s8[0]].tolist()
        assert int(ids[local]) in top
    ix.close()


def test_recent_10m(corpus):
    """N3 at full size: created_ms is strictly increasing with the claim id in this corpus, so the newest-first
    answer is known in closed form; `matched` must count exactly the rows at or after the bound."""
    hip, ix = corpus["hip"], corpus["ix"]
    st = hip.Stats()
    base = 1_700_000_000_000
    for since, limit in [(hip.SINCE_NONE, 8), (hip.SINCE_NONE, 200), (base + 1000 * (N_FULL - 150), 1000), (base + 1000 * 5_000_000, 64)]:
        ids, created, matched = ix.recent(since, limit, st)
        first = 0 if since == hip.SINCE_NONE else (since - base + 999) // 1000
        exp_matched = N_FULL - first
        take = min(limit, exp_matched)
        assert matched == exp_matched
        assert ids.tolist() == list(range(N_FULL - 1, N_FULL - 1 - take, -1))
        assert created.tolist() == [base + 1000 * i for i in range(N_FULL - 1, N_FULL - 1 - take, -1)]
        assert st.path == 3 and st.rows_scanned == N_FULL
    # the two half-corpus shards hold the rows round-robin: merging their answers by (created desc) gives the same list
    a, ca, _ = corpus["half"][0].recent(hip.SINCE_NONE, 64)
    b, cb, _ = corpus["half"][1].recent(hip.SINCE_NONE, 64)
    merged = sorted(ca.tolist() + cb.tolist(), reverse=True)[:64]
    assert merged == [base + 1000 * i for i in range(N_FULL - 1, N_FULL - 65, -1)]
