"""Round-3 GPU tests: the batched claim-sharded search (lbh_index_search_sharded_many), failure containment and the
bounded wait of the sharded searches, lbh_comm_allgather, and the claim-sharded graph's batched / recency / snapshot
entries on the real backend -- all with a one-rank RCCL communicator (the box has one GPU).  Bit-exact, through the C ABI."""
import numpy as np
import pytest

from oracle import lb_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from lbm_amd import _hip
    assert "gfx950" in _hip.device_name()
    return _hip


@pytest.fixture(scope="module")
def shard(hip):
    from lbm_amd import synth
    ch = synth.make_chunk(0, 120_000, seed=19, real_sha_keys=False, dup_frac=0.01)
    ix = hip.Index()
    ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    ix.retract_many(ids[ch.retracted])
    comm = hip.Comm(hip.comm_unique_id(), 0, 1)
    yield ch, ix, comm
    comm.close()
    ix.close()


def test_sharded_many_equals_index_search_single_rank(hip, shard):
    """Local K3 pass -> per-query lists -> ONE all-gather -> merges must give exactly lbh_index_search's answers (and the
    single-query sharded entry's keys), for batches that take the matrix-core path, the paged path (k > 64), the
    all-ties query that falls back to the exact scan, filters, and batches of one and two queries."""
    from lbm_amd import synth
    ch, ix, comm = shard
    qv = hip.embed_batch(synth.make_queries(37, seed=5) + [ch.text(5), ""])
    p50 = int(ch.created_ms[ch.n // 2])
    st = hip.Stats()
    for nq in (39, 1, 2, 3):
        for k, since, tm in ((8, hip.SINCE_NONE, 0), (64, p50, 0), (5, hip.SINCE_NONE, 0xF0F0), (100, hip.SINCE_NONE, 0), (0, p50, 0)):
            q = qv[-nq:]
            wi, ws, wc = ix.search(q, k, since, tm)
            gi, gs, gk, gc = ix.search_sharded_many(comm, q, k, since, tm, st, gid_mul=1, gid_add=0)
            assert gc.tolist() == wc.tolist(), (nq, k)
            for j in range(nq):
                c = int(wc[j])
                assert gi[j, :c].tolist() == wi[j, :c].tolist(), (nq, k, j)
                assert gs[j, :c].tobytes() == ws[j, :c].tobytes()
                if k and j in (0, nq - 1):
                    _, _, sk = ix.search_sharded(comm, q[j], k, since, tm)
                    assert gk[j, :c].tobytes() == sk.tobytes()
            if nq >= 3 and 0 < k <= 64:
                assert st.path == 2 and st.rows_scanned >= ix.size()  # the batch really took the matrix-core pass
    gi, _, _, _ = ix.search_sharded_many(comm, qv[:4], 8, gid_mul=8, gid_add=5)
    wi, _, _ = ix.search(qv[:4], 8)
    assert gi.tolist() == (wi * 8 + 5).tolist()


def test_sharded_many_against_the_oracle(hip, shard):
    """... and once against the CPU oracle itself (hash keys included), so the chain of equalities ends in the reference."""
    from lbm_amd import synth
    ch, ix, comm = shard
    n = 20_000
    small = hip.Index()
    small.add_texts(ch.blob[: ch.offsets[n] + 1], ch.offsets[: n + 1], ch.created_ms[:n], ch.tagmask[:n], ch.keys[:n])
    emb = orc.embed_batch_packed(ch.blob[: ch.offsets[n] + 1], ch.offsets[: n + 1])
    hexes = [bytes(r).hex() for r in ch.keys[:n]]
    qs = synth.make_queries(6, seed=8) + [""]
    qv = hip.embed_batch(qs)
    for k, since, tm in ((8, None, 0), (64, int(ch.created_ms[n // 3]), 0xFF)):
        gi, gs, gk, gc = small.search_sharded_many(comm, qv, k, hip.SINCE_NONE if since is None else since, tm, gid_mul=1, gid_add=0)
        for j, q in enumerate(qs):
            oi, osc = orc.compile_arrays(emb, ch.created_ms[:n], np.zeros(n, np.uint8), hexes, orc.embed_array(q), k, since, ch.tagmask[:n], tm)
            c = int(gc[j])
            assert gi[j, :c].tolist() == oi.tolist() and gs[j, :c].tobytes() == osc.tobytes(), (k, j)
            assert gk[j, :c].tobytes() == ch.keys[:n][oi].tobytes()
    small.close()


def test_local_failure_still_reaches_the_all_gather(hip, shard):
    """Failure containment: a rank whose LOCAL search fails must not leave before the all-gather (its peers would wait
    inside it for ever).  The injected fault makes this rank's local part fail; the call still runs the exchange -- the
    list's count word carries the error -- returns an error, and the communicator stays usable."""
    ch, ix, comm = shard
    q = hip.embed_batch(["w00001 w00002"])[0]
    want = ix.search_shard(q, 8)
    for call in (lambda: ix.search_sharded(comm, q, 8), lambda: ix.search_sharded(comm, q, 100),
                 lambda: ix.search_sharded_many(comm, np.stack([q] * 5), 8)):
        hip.set_option("inject_fault", 1)
        with pytest.raises(hip.HipBackendError, match="injected fault"):
            call()
        gi, gs, _ = ix.search_sharded(comm, q, 8)  # next call: business as usual
        assert gi.tolist() == want[0].tolist() and gs.tobytes() == want[1].tobytes()


def test_a_missing_peer_ends_in_a_timeout_not_a_hang(hip, shard):
    """A peer that never reaches the all-gather is played by a host function that holds this rank's stream up for four
    times the timeout: the call must come back with an error after ~coll_timeout_ms, and the communicator must refuse
    further collectives (the queued one is still unfinished behind it)."""
    import time
    ch, ix, _ = shard
    comm = hip.Comm(hip.comm_unique_id(), 0, 1)  # its own communicator: it ends up broken
    q = hip.embed_batch(["w00003"])[0]
    hip.set_option("coll_timeout_ms", 150)
    try:
        hip.set_option("inject_fault", 2)
        t0 = time.perf_counter()
        with pytest.raises(hip.HipBackendError, match="no completion after 150 ms"):
            ix.search_sharded(comm, q, 8)
        assert 0.1 < time.perf_counter() - t0 < 0.55
        with pytest.raises(hip.HipBackendError, match="unusable"):
            ix.search_sharded(comm, q, 8)
        with pytest.raises(hip.HipBackendError, match="unusable"):
            ix.search_sharded_many(comm, np.stack([q] * 4), 8)
        time.sleep(0.7)  # the sleeping host function returns, the stream drains: the index itself is fine
        assert ix.search_shard(q, 8)[0].shape[0] == 8
    finally:
        hip.set_option("coll_timeout_ms", 60000)
        hip.set_option("inject_fault", 0)
        comm.close()


def test_comm_allgather_bytes(hip, shard):
    _, _, comm = shard
    for payload in (b"", b"x", bytes(range(256)) * 300):
        assert comm.allgather(payload) == [payload]


def test_sharded_graph_batched_recent_snapshot_single_rank(hip, golden_synth):
    """ShardedContextGraph on the real backend (one-rank RCCL communicator): compile_many through
    lbh_index_search_sharded_many, the winners' text through lbh_comm_allgather, the recency query and the gathered
    snapshot -- against the oracle graph."""
    from lbm_amd.sharded import ShardedContextGraph
    d = golden_synth
    comm = hip.Comm(hip.comm_unique_id(), 0, 1)
    try:
        g, og = ShardedContextGraph(0, 1, comm=comm), orc.OracleGraph()
        for i in range(600):
            tags = ["t%02d" % j for j in range(64) if (d["tagmask"][i] >> j) & 1]
            for gg in (g, og):
                gg.add_claim(d["hashes"][i], d["texts"][i], tags, created_ms=d["created_ms"][i // 3 * 3])
        for i in range(0, 600, 37):
            g.retract(d["hashes"][i]); og.retract(d["hashes"][i])
        qs = ["w00001 w00004", "w00002", "", d["texts"][7], "w00010 w00003 w00001"]
        for kw in ({"top_k": 8}, {"top_k": 70, "since_ms": d["created_ms"][300]}, {"top_k": 5, "tags": ["t03", "t07"]}, {"top_k": 0}):
            assert g.compile_many(qs, **kw) == [og.compile(q, **kw) for q in qs], kw
            assert g.compile(qs[0], **kw) == og.compile(qs[0], **kw)
        for since, limit in ((0, 100), (d["created_ms"][300], 7), (0, None), (0, -590), (10 ** 18, 5)):
            want = orc.recent_claims(og.claims, since, limit)
            assert g.recent_claims(since, limit) == want, (since, limit)
        assert g.snapshot() == og.snapshot()
    finally:
        comm.close()


def test_k1_token_memo_is_exact_cold_warm_and_overfull(hip):
    """K1's token memo (token -> its two SHA-256 digests, kept in HBM across launches; lb/latent.py:47-49 is a pure
    function of the lower-cased token) must never change a bit: cold table, warm table, a table far too small for the
    vocabulary (every bucket full: misses fall back to the compression), tokens that differ only in case (one entry),
    tokens the memo does not serve (two-block messages, tokens beyond a claim's prefetched head), and the memo switched off."""
    from lbm_amd import synth
    rng = np.random.default_rng(5)
    ch = synth.make_chunk(0, 60_000, seed=77, real_sha_keys=False)
    texts = [ch.text(i) for i in range(0, 60_000, 12)]
    alphabet = list("abcdefgXYZ0189_") + [" ", " ", "-", "é"]
    for i in range(2500):  # a large, flat vocabulary of mixed-case tokens, some longer than one SHA block, some past byte 128
        n_tok = int(rng.integers(0, 30))
        toks = ["".join(rng.choice(alphabet, size=int(rng.integers(1, 12)))) for _ in range(n_tok)]
        if i % 50 == 0:
            toks.insert(int(rng.integers(0, n_tok + 1)), "L" * int(rng.integers(50, 70)))
        texts.append(" ".join(toks))
    texts += ["Hello hello HELLO hElLo", "hello", "", "w00001 W00001"]
    want = orc.embed_batch(texts)
    try:
        for lg in (20, 8, 0):
            hip.set_option("k1_memo_log2", lg)
            for rep in range(3):  # launch 1 fills the table, launches 2 and 3 read it
                got = hip.embed_batch(texts)
                assert got.tobytes() == want.tobytes(), (lg, rep)
            # a different batch against the table the first one left behind
            other = [t.upper() for t in texts[:3000]] + texts[3000:6000]
            assert hip.embed_batch(other).tobytes() == orc.embed_batch(other).tobytes(), lg
        # through the index (sliced ingest: several launches per call, each its own epoch)
        hip.set_option("k1_memo_log2", 20)
        ix = hip.Index()
        for _ in range(2):
            ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
            pick = rng.choice(ch.n, size=1500, replace=False)
            assert ix.get_embeddings(ids[pick]).tobytes() == orc.embed_batch([ch.text(int(i)) for i in pick]).tobytes()
        ix.close()
    finally:
        hip.set_option("k1_memo_log2", 20)


def test_cross_block_bound_for_large_k_is_exact(hip):
    """K2's cross-block bound (top_k >= scan_xb_k: at a compaction a wave takes the k-th largest of the blocks' best scores
    as its threshold) must not change a result: against the exact fp64 path and against the kernel with the bound off, on
    duplicate-heavy corpora of several sizes (grids below and above k blocks), k from the switch-over to the page size,
    filters that leave few valid rows per block, and -- because the slots live across launches -- queries whose scores
    are orders of magnitude BELOW the previous launch's (a stale slot taken for a current one would cut every row)."""
    from lbm_amd import synth
    st = hip.Stats()
    try:
        for n in (3_000, 40_000, 700_000):
            ch = synth.make_chunk(0, n, seed=23 + n % 7, real_sha_keys=False, dup_frac=0.03)
            ix = hip.Index()
            ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
            qv = hip.embed_batch(synth.make_queries(3, seed=3) + [ch.text(11)])
            p50, p97 = int(ch.created_ms[n // 2]), int(ch.created_ms[n * 97 // 100])
            cases = [(64, hip.SINCE_NONE, 0), (24, hip.SINCE_NONE, 0), (33, p97, 0), (64, hip.SINCE_NONE, 1 << 40),
                     (50, p50, 0b11), (64, p97, 1 << 41)]
            for qi, scale in ((0, 1.0), (0, 1e-3), (1, 1.0), (2, 40.0), (3, 1e-2), (3, 1.0)):
                q = qv[qi] * scale
                for k, since, tm in cases:
                    got = [a.copy() for a in ix.search_shard(q, k, since, tm, st)]  # (the arrays are reused per k)
                    if since == hip.SINCE_NONE and tm == 0 and n >= 40_000:
                        assert st.path == 0, (n, k, st.path)  # the fast path served it
                    hip.set_option("scan_xb_k", 0)
                    off = [a.copy() for a in ix.search_shard(q, k, since, tm)]
                    hip.set_option("scan_xb_k", 24)
                    hip.set_option("force_exact", 1)
                    want = [a.copy() for a in ix.search_shard(q, k, since, tm)]
                    hip.set_option("force_exact", 0)
                    for other in (off, want):
                        assert got[0].tolist() == other[0].tolist(), (n, qi, scale, k, since, tm)
                        assert got[1].tobytes() == other[1].tobytes() and got[2].tobytes() == other[2].tobytes()
            ix.close()
    finally:
        hip.set_option("scan_xb_k", 24)
        hip.set_option("force_exact", 0)


def test_int8_copy_follows_appends_tombstones_and_a_change_of_scale(hip):
    """bq_mode 3 keeps an int8 copy of the matrix (one scale for the index) next to the fp64 master: it must follow appended
    rows, single and batched swap-deletes, and be quantised again when a new row's largest component exceeds the scale --
    after every step the int8 batched search must equal the fp16 one (both are exact: same ids, same fp64 score bits)
    and must have been served by the int8 kernel."""
    rng = np.random.default_rng(11)

    def unit(m):
        v = rng.standard_normal((m, 64))
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    def keys(m):
        return rng.integers(0, 256, size=(m, 32), dtype=np.uint8)

    ix = hip.Index()
    ids = ix.add_vectors(unit(30_000), np.arange(30_000, dtype=np.int64), None, keys(30_000))
    qv = unit(9)
    st = hip.Stats()

    def check(tag):
        hip.set_option("bq_mode", 3)
        a = [x.copy() for x in ix.search(qv, 8, hip.SINCE_NONE, 0, st)]
        assert st.path == 2 and st.bytes_per_row == 64 and 0.0 < st.margin < 0.09 and st.max_score_err < st.margin / 2, (tag, st.bytes_per_row, st.margin)
        hip.set_option("bq_mode", 2)
        b = ix.search(qv, 8, hip.SINCE_NONE, 0, st)
        assert st.bytes_per_row == 128
        assert a[0].tolist() == b[0].tolist() and a[1].tobytes() == b[1].tobytes() and a[2].tolist() == b[2].tolist(), tag

    try:
        check("built")
        ids2 = ix.add_vectors(unit(5_000), np.arange(5_000, dtype=np.int64), None, keys(5_000))
        check("appended")
        ix.retract_many(np.concatenate([ids[100:4_000:3], ids2[::5]]))
        check("batched swap-delete")
        for i in (7, 29_999, 20_000):
            ix.retract(int(ids[i]))
        check("single swap-deletes")
        spike = np.zeros((2, 64))
        spike[0, 5] = 1.0  # a unit vector with ONE component: far above the scale the random rows set (~0.45 / 127)
        spike[1, :2] = [0.8, -0.6]
        ix.add_vectors(spike, np.zeros(2, dtype=np.int64), None, keys(2))
        check("scale raised, copy quantised again")
        ix.add_vectors(unit(3_000), np.zeros(3_000, dtype=np.int64), None, keys(3_000))
        check("appended under the new scale")
    finally:
        hip.set_option("bq_mode", 2)
        ix.close()


def test_select_flavours_agree_two_kernels_fused_and_fused_on_a_large_grid(hip):
    """The single-query scan ends in one of three ways: its own select kernel (scan_fuse 0), the select step run by the
    scan's last block (1, the default for small and medium indexes), and the same on a grid of more blocks than the last
    block has threads (scan_fuse 2 with four blocks per CU: 1024 blocks, more than the 512 slots of the k-th-largest
    bound).  All must return what the exact fp64 path returns."""
    from lbm_amd import synth
    ch = synth.make_chunk(0, 600_000, seed=41, real_sha_keys=False, dup_frac=0.02)
    ix = hip.Index()
    ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    qv = hip.embed_batch(synth.make_queries(4, seed=9) + [ch.text(5), ""])
    p90 = int(ch.created_ms[ch.n * 9 // 10])
    cases = [(8, hip.SINCE_NONE, 0), (64, hip.SINCE_NONE, 0), (20, p90, 0), (8, hip.SINCE_NONE, 1 << 3), (64, p90, 0b101)]
    try:
        hip.set_option("force_exact", 1)
        want = {(j, c): [a.copy() for a in ix.search_shard(qv[j], *c)] for j in range(len(qv)) for c in cases}
        hip.set_option("force_exact", 0)
        for fuse, bpc in ((0, 0), (1, 0), (2, 0), (2, 4), (0, 4)):
            hip.set_option("scan_fuse", fuse)
            hip.set_option("scan_blocks_per_cu", bpc)
            for (j, c), w in want.items():
                got = ix.search_shard(qv[j], *c)
                assert got[0].tolist() == w[0].tolist() and got[1].tobytes() == w[1].tobytes() and got[2].tobytes() == w[2].tobytes(), (fuse, bpc, j, c)
    finally:
        hip.set_option("scan_fuse", 1)
        hip.set_option("scan_blocks_per_cu", 0)
        hip.set_option("force_exact", 0)
        ix.close()


def test_small_and_bulk_ingest_paths_give_the_same_index(hip):
    """lbh_index_add takes batches of up to 256 claims / 40 KB of text through one device-mapped block and two launches,
    anything larger through the sliced bulk path (pageable buffers: the staging ring).  Either way the rows must carry the
    oracle's embedding bits, their timestamps, tag masks, keys and ids -- checked at the sizes where the path switches,
    with empty texts, no tag masks, and a small batch whose text alone is too long for the block."""
    from lbm_amd import synth
    rng = np.random.default_rng(3)
    ch = synth.make_chunk(0, 3000, seed=5, real_sha_keys=False)
    ix = hip.Index()
    all_ids, all_texts, all_created, all_tags, all_keys = [], [], [], [], []
    pos = 0
    for m in (1, 63, 64, 65, 255, 256, 257, 2, 300, 1):
        texts = [ch.text(pos + i) if (pos + i) % 17 else "" for i in range(m)]
        if m == 2:
            texts[1] = " ".join("w%05d" % int(x) for x in rng.integers(0, 90000, size=7000))  # ~49 KB: too long for the block
        blob = np.frombuffer("".join(texts).encode() + b"\x00", dtype=np.uint8).copy()
        offs = np.zeros(m + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(t.encode()) for t in texts])
        created = ch.created_ms[pos:pos + m].copy()
        tags = None if m in (64, 257) else ch.tagmask[pos:pos + m].copy()
        keys = ch.keys[pos:pos + m].copy()
        ids = ix.add_texts(blob, offs, created, tags, keys)
        assert ids.tolist() == list(range(len(all_texts), len(all_texts) + m))
        all_ids += ids.tolist(); all_texts += texts; all_created += created.tolist()
        all_tags += (np.zeros(m, np.uint64) if tags is None else tags).tolist(); all_keys.append(keys)
        pos += m
    n = len(all_texts)
    want = orc.embed_batch(all_texts)
    got = ix.get_embeddings(np.array(all_ids, dtype=np.uint64))
    assert got.tobytes() == want.tobytes()
    keys = np.concatenate(all_keys)
    hexes = [bytes(r).hex() for r in keys]
    created, tagm = np.array(all_created, dtype=np.int64), np.array(all_tags, dtype=np.uint64)
    for q, k, since, tm in ((all_texts[5], 20, None, 0), (all_texts[400], 64, int(np.median(created)), 0), (all_texts[1100], 9, None, 0b1110)):
        qv = orc.embed_array(q)
        oi, osc = orc.compile_arrays(want, created, np.zeros(n, np.uint8), hexes, qv, k, since, tagm, tm)
        gi, gs, gk = ix.search_shard(qv, k, hip.SINCE_NONE if since is None else since, tm)
        assert gi.tolist() == oi.tolist() and gs.tobytes() == osc.tobytes() and gk.tobytes() == keys[oi].tobytes()
    ix.close()
