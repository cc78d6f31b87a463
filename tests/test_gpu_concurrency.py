"""Readers and writers on ONE index / ONE graph from several threads (searches concurrent, add / retract
exclusive; the reference's compile_context takes no lock, lb/node.py:718-734, while publishers hold BatteryNode._lock).
Every answer a reader gets while writers churn unrelated claims must be bit-identical to the quiet answer."""
import hashlib
import threading

import numpy as np
import pytest

from oracle import lb_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from lbm_amd import _hip
    assert "gfx950" in _hip.device_name()
    return _hip


def _run_threads(fns):
    errs = []

    def wrap(fn):
        def run():
            try:
                fn()
            except BaseException as e:  # noqa: BLE001 -- reported by the main thread
                errs.append(repr(e))
        return run

    ts = [threading.Thread(target=wrap(fn)) for fn in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in ts), "a thread did not finish"
    assert not errs, errs


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scan_mode", [0, 1])
def test_index_searches_overlap_with_each_other_and_with_writers(hip, scan_mode):
    """C ABI level: 3 reader threads (vector queries, text queries, batched queries + recent) against 1 writer thread
    that adds and retracts claims whose vector is zero (texts without a token, or zero vectors handed in): score 0, never
    in a top-k whose last score is positive."""
    from lbm_amd import synth
    n = 60_000
    ch = synth.make_chunk(0, n, seed=77)
    hip.set_option("scan_mode", scan_mode)
    try:
        ix = hip.Index()
        ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
        emb = orc.embed_batch_packed(ch.blob, ch.offsets)
        keys_hex = ch.hashes()
        live = np.zeros(n, dtype=np.uint8)
        qtexts = synth.make_queries(12, seed=5)
        qv = np.stack([orc.embed_array(t) for t in qtexts])
        k = 8
        want = []
        for j in range(len(qtexts)):
            idx, sc = orc.compile_arrays(emb, ch.created_ms, live, keys_hex, qv[j], k, None)
            assert sc[-1] > 0.0  # zero-score newcomers cannot enter this top-k
            want.append((ids[idx].tolist(), sc.tobytes()))
        since = int(ch.created_ms[n // 2])
        want_recent = ix.recent(since, 16)[0].tolist()
        stop = threading.Event()
        rounds = {"w": 0, "a": 0, "b": 0, "c": 0}

        def writer():
            # a text without a token embeds to the zero vector (lb/latent.py:44-46); created_ms below `since`
            rng = np.random.default_rng(1)
            while not stop.is_set():
                m = int(rng.integers(1, 400))
                kk = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
                if rounds["w"] % 3 == 2:
                    new = ix.add_vectors(np.zeros((m, 64)), np.full(m, 5, dtype=np.int64), None, kk)
                else:
                    texts = [b"?! " * (1 + (i + rounds["w"]) % 40) for i in range(m)]
                    blob = np.frombuffer(b"".join(texts), dtype=np.uint8)
                    off = np.zeros(m + 1, dtype=np.int64)
                    off[1:] = np.cumsum([len(t) for t in texts])
                    new = ix.add_texts(blob, off, np.full(m, 5, dtype=np.int64), None, kk)
                if rounds["w"] % 2:
                    ix.retract_many(new)
                else:
                    for i in new[: m // 2].tolist():
                        ix.retract(i)
                    ix.retract_many(new[m // 2:])
                rounds["w"] += 1

        def reader_vec():
            for r in range(150):
                j = r % len(qtexts)
                got_ids, got_sc, _ = ix.search_shard(qv[j], k)
                assert (got_ids.tolist(), got_sc.tobytes()) == want[j], ("vec", r)
                rounds["a"] += 1

        def reader_text():
            for r in range(150):
                j = (r * 5) % len(qtexts)
                got_ids, got_sc, _ = ix.search_text(qtexts[j], k)
                assert (got_ids.tolist(), got_sc.tobytes()) == want[j], ("text", r)
                rounds["b"] += 1

        def reader_batched():
            for r in range(40):
                got_ids, got_sc, cnt = ix.search(qv, k)
                for j in range(len(qtexts)):
                    assert int(cnt[j]) == k
                    assert (got_ids[j].tolist(), got_sc[j].tobytes()) == want[j], ("batched", r, j)
                assert ix.recent(since, 16)[0].tolist() == want_recent
                rounds["c"] += 1

        def readers_then_stop():
            try:
                _run_threads([reader_vec, reader_text, reader_batched])
            finally:
                stop.set()

        _run_threads([writer, readers_then_stop])
        assert rounds["a"] == 150 and rounds["b"] == 150 and rounds["c"] == 40 and rounds["w"] > 0
        assert ix.size() == n
        ix.close()
    finally:
        hip.set_option("scan_mode", 0)


@pytest.mark.timeout(300)
def test_graph_compile_from_threads_while_a_publisher_adds_and_retracts(hip, golden_synth):
    """Graph level: the reference's golden answers (hashes, fp64 score bits, context text) from 4 threads while a
    publisher thread adds, overwrites and retracts claims without a token (zero vector, score 0)."""
    from lbm_amd.context_graph import ContextGraph
    d = golden_synth
    g = ContextGraph()
    for i in range(d["n"]):
        tags = ["t%02d" % j for j in range(64) if (d["tagmask"][i] >> j) & 1]
        g.add_claim(d["hashes"][i], d["texts"][i], tags, created_ms=d["created_ms"][i])
    for i in range(d["n"]):
        if d["retracted"][i]:
            g.retract(d["hashes"][i])
    cases = [r for r in d["results"] if r["scores"] and float.fromhex(r["scores"][-1]) > 0.0 and len(r["hashes"]) == r["top_k"]]
    assert len(cases) >= 20
    stop = threading.Event()
    done = []

    def publisher():
        i = 0
        while not stop.is_set():
            h = hashlib.sha256(b"churn %d" % (i % 50)).hexdigest()
            g.add_claim(h, "?! " * (1 + i % 30), [], created_ms=1)
            if i % 3 == 0:
                g.retract(h)
            i += 1
        done.append(i)

    def reader(offset):
        def run():
            for r in cases[offset::4]:
                kw = {"top_k": r["top_k"]}
                if r["since_ms"] is not None:
                    kw["since_ms"] = r["since_ms"]
                text, hs = g.compile(r["query"], **kw)
                assert hs == r["hashes"], r["query"]
                assert hashlib.sha256(text.encode()).hexdigest() == r["context_sha256"]
                _, sc = g.compile_scored(r["query"], **kw)
                assert [float(x).hex() for x in sc] == r["scores"]
        return run

    def readers_then_stop():
        try:
            _run_threads([reader(o) for o in range(4)])
        finally:
            stop.set()

    _run_threads([publisher, readers_then_stop])
    assert done and done[0] > 0


@pytest.mark.timeout(300)
def test_more_searching_threads_than_slots(hip):
    """Eight threads on one index: more than the library has per-search workspaces (four), so some wait for a slot; every
    answer equals the single-threaded one, and a batched search, a recency query and an append slip in between."""
    rng = np.random.default_rng(5)
    n = 200_000
    v = rng.standard_normal((n, 64))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    ix = hip.Index()
    ix.add_vectors(v, np.arange(n, dtype=np.int64), None, rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
    qs = rng.standard_normal((32, 64))
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    want = [tuple(x.tobytes() for x in ix.search_shard(qs[j], 8)[:2]) for j in range(32)]
    want_b = ix.search(qs, 8)
    counts = [0] * 8

    def worker(t):
        def run():
            for r in range(120):
                j = (r * 3 + t) % 32
                got = ix.search_shard(qs[j], 8)
                assert (got[0].tobytes(), got[1].tobytes()) == want[j], (t, r)
                if t == 0 and r % 40 == 7:
                    gi, gs, gc = ix.search(qs, 8)
                    assert gi.tobytes() == want_b[0].tobytes() and gs.tobytes() == want_b[1].tobytes()
                if t == 1 and r % 40 == 9:
                    assert len(ix.recent(n - 10, 64)[0]) == 10
                if t == 2 and r % 40 == 11:  # a zero vector never enters a top-8 whose scores are positive
                    ix.retract_many(ix.add_vectors(np.zeros((3, 64)), np.zeros(3, dtype=np.int64), None,
                                                   rng.integers(0, 256, size=(3, 32), dtype=np.uint8)))
                counts[t] += 1
        return run

    _run_threads([worker(t) for t in range(8)])
    assert counts == [120] * 8 and ix.size() == n
    ix.close()
