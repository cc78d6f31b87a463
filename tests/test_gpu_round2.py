"""Round-2 GPU parity tests: snapshot reload against vectors from the reference, embed(text, dim < 64), the batched
tombstone path (lbh_index_retract_many) and the host mirror's failure handling.  Bit-exact, through the C ABI."""
import hashlib
import json

import numpy as np
import pytest

from oracle import lb_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from lbm_amd import _hip
    assert "gfx950" in _hip.device_name()
    return _hip


def hexes(v):
    return [float(x).hex() for x in v]


def test_embed_dim_golden(hip, golden_embed_dim):
    """lb/latent.py:41-57 with dim != 64, against the reference's own outputs."""
    from lbm_amd.latent import embed
    for c in golden_embed_dim["cases"] + golden_embed_dim["special"]:
        got = embed(c["text"], c["dim"])
        assert isinstance(got, list) and len(got) == c["dim"]
        assert hexes(got) == c["embedding"], (c["text"][:20], c["dim"])
    for c in golden_embed_dim["errors"]:
        with pytest.raises(IndexError):
            embed(c["text"], c["dim"])
    # batched, against the oracle, ragged texts x every dim
    rng = np.random.default_rng(3)
    texts = ["".join(rng.choice(list("abcXYZ019_ -.é"), size=int(rng.integers(0, 200)))) for _ in range(300)]
    for dim in (1, 7, 32, 63):
        got = hip.embed_batch(texts, dim)
        assert np.all(got[:, dim:] == 0.0)
        for i in range(0, len(texts), 7):
            assert hexes(got[i, :dim]) == hexes(orc.embed(texts[i], dim)), (i, dim)


def _replay(ops):
    from lbm_amd.context_graph import Claim, ContextGraph
    g = ContextGraph()
    for op in ops:
        if op[0] == "add":
            g.add_claim(op[1], op[2], op[3], created_ms=op[4], parent_hash=op[5] if len(op) > 5 else None)
        elif op[0] == "retract":
            g.retract(op[1])
        elif op[0] == "raw_insert":
            g.claims[op[1]] = Claim(claim_hash=op[1], text=op[2], tags=list(op[3]), created_ms=op[4], evidence=[])
        elif op[0] == "flag":
            g.claims[op[1]].retracted = op[2]
    return g


def test_snapshot_reload_golden_small(hip, golden_snapshot):
    """snapshot() of a graph with overwritten / retracted / re-added / raw-inserted / hand-flagged claims equals the
    reference's dict; from_snapshot(reference dict) -> compile equals the reference's reloaded graph (:135-148)."""
    from lbm_amd.context_graph import ContextGraph
    case = golden_snapshot[0]
    for interleave in (False, True):
        g = _replay(case["ops"])
        if interleave:
            g.compile("beta")  # the mirror has been on the device before the snapshot is taken
        assert g.snapshot() == case["snapshot"]
    g2 = ContextGraph.from_snapshot(json.loads(json.dumps(case["snapshot"])))
    assert sorted(g2._embedding_cache) == case["cached_after_load"]
    for r in case["results"]:
        kw = {k: r[k] for k in ("top_k", "since_ms") if k in r}
        assert g2.compile(r["query"], **kw) == (r["context"], r["hashes"]), (r["query"], kw)
        assert hexes(g2.compile_scored(r["query"], **kw)[1]) == r["scores"]
    # a reloaded graph keeps working as a live graph
    og = orc.OracleGraph.from_snapshot(case["snapshot"])
    for gg in (g2, og):
        gg.retract("k1")
        gg.add_claim("k7", "beta epsilon gamma", ["n"], created_ms=90)
        gg.claims["k5"].retracted = False
    for q in ("beta", "epsilon", ""):
        assert g2.compile(q, top_k=10) == og.compile(q, top_k=10)


def test_snapshot_reload_golden_synth2k(hip, golden_snapshot, golden_synth):
    from lbm_amd.context_graph import ContextGraph
    case, d = golden_snapshot[1], golden_synth
    n = case["n"]
    g = ContextGraph()
    for i in range(n):
        tags = ["t%02d" % j for j in range(64) if (d["tagmask"][i] >> j) & 1]
        g.add_claim(d["hashes"][i], d["texts"][i], tags, created_ms=d["created_ms"][i])
    g.compile("w00001")  # everything on the device first: the overwrites and retracts below hit live rows
    for i in range(n):
        if d["retracted"][i]:
            g.retract(d["hashes"][i])
    for i in case["overwrite"]:
        g.add_claim(d["hashes"][i], d["texts"][(i * 7 + 1) % n], ["over"], created_ms=d["created_ms"][i] + 5)
    for i in case["retract_extra"]:
        g.retract(d["hashes"][i])
    snap = g.snapshot()
    assert hashlib.sha256(json.dumps(snap, sort_keys=True).encode()).hexdigest() == case["snapshot_sha256"]
    for graph in (g, ContextGraph.from_snapshot(json.loads(json.dumps(snap)))):
        assert len(graph._embedding_cache) == case["n_cached_after_load"]
        for r in case["results"]:
            kw = {k: r[k] for k in ("top_k", "since_ms") if k in r}
            text, hs = graph.compile(r["query"], **kw)
            assert hs == r["hashes"], (r["query"], kw)
            assert hashlib.sha256(text.encode()).hexdigest() == r["context_sha256"]
            assert hexes(graph.compile_scored(r["query"], **kw)[1]) == r["scores"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_retract_many_vs_oracle(hip, seed):
    """C-ABI level: random batches leave the index in one compaction each; searches, recency queries and cached
    vectors must match the oracle on the surviving claims after every batch, in both scan modes."""
    rng = np.random.default_rng(seed)
    n = 20_000
    v = rng.standard_normal((n, 64))
    v[1::50] = v[0::50][: v[1::50].shape[0]]
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    ms = rng.integers(0, 5000, size=n).astype(np.int64)
    tags = rng.integers(0, 2 ** 63, size=n).astype(np.uint64)
    keys = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    packed = orc.pack_texts([bytes(k).hex() for k in keys])
    ix = hip.Index()
    ids = ix.add_vectors(v, ms, tags, keys)
    dead = np.zeros(n, dtype=np.uint8)
    qs = rng.standard_normal((6, 64))
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    hip.set_option("scan_mode", seed % 2)
    try:
        for batch in (1, 7, 500, 3000, 1, 9000, 64):
            alive = np.nonzero(dead == 0)[0]
            if batch == 9000:  # the whole tail and more: few movers, many tail deletions
                pick = alive[-batch:]
            else:
                pick = rng.choice(alive, size=min(batch, alive.shape[0]), replace=False)
            ix.retract_many(ids[pick])
            dead[pick] = 1
            assert ix.size() == int((dead == 0).sum())
            for q in qs[:3]:
                for k, since, tm in ((8, None, 0), (64, 2500, 0), (8, None, 0xFF00)):
                    wi, ws = orc.compile_arrays(v, ms, dead, None, q, k, since, tags, tm, packed_keys=packed)
                    gi, gs, _ = ix.search_shard(q, k, hip.SINCE_NONE if since is None else since, tm)
                    assert gi.tolist() == wi.tolist() and gs.tobytes() == ws.tobytes(), (batch, k, since, tm)
            bi, bs, bc = ix.search(qs, 8)
            for j in range(qs.shape[0]):
                wi, ws = orc.compile_arrays(v, ms, dead, None, qs[j], 8, None, packed_keys=packed)
                assert bi[j, : int(bc[j])].tolist() == wi.tolist() and bs[j, : int(bc[j])].tobytes() == ws.tobytes()
            alive = np.nonzero(dead == 0)[0]
            sample = rng.choice(alive, size=200, replace=False)
            assert ix.get_embeddings(ids[sample]).tobytes() == v[sample].tobytes()
            rid, rcreated, matched = ix.recent(1000, 50)
            want = orc.recent_arrays(np.where(dead == 0, ms, -1), np.arange(n), 1000, 50)
            assert rid.tolist() == want.tolist() and matched == int(((dead == 0) & (ms >= 1000)).sum())
        with pytest.raises(hip.HipBackendError, match="unknown claim id"):
            ix.retract_many(ids[np.nonzero(dead)[0][:1]])
        alive = np.nonzero(dead == 0)[0]
        with pytest.raises(hip.HipBackendError, match="twice"):
            ix.retract_many(ids[[alive[0], alive[1], alive[0]]])
        assert ix.size() == alive.shape[0]  # failed calls change nothing
        ix.retract_many(ids[alive])  # everything
        assert ix.size() == 0 and ix.search_shard(qs[0], 8)[0].shape[0] == 0
    finally:
        hip.set_option("scan_mode", 0)
        ix.close()


def test_failed_device_call_keeps_the_mirror_consistent(hip, monkeypatch):
    """a failing add must not lose claims.  The first add_texts of a sync is made to fail; the next
    compile must retry and see every claim, parked vectors included."""
    from lbm_amd.context_graph import ContextGraph
    g, og = ContextGraph(), orc.OracleGraph()
    for gg in (g, og):
        for i in range(50):
            gg.add_claim("h%02d" % i, "alpha beta w%d" % (i % 7), ["t"], created_ms=i)
    assert g.compile("alpha w3", top_k=5) == og.compile("alpha w3", top_k=5)
    for gg in (g, og):
        gg.claims["h03"].retracted = True      # parked by the next sync
        gg.retract("h04")
        gg.add_claim("n1", "alpha gamma", [], created_ms=100)
        gg.add_claim("n2", "beta w3", [], created_ms=101)
        gg.add_claim("h05", "overwritten w3 w3", [], created_ms=102)
    real = hip.Index.add_texts
    calls = {"n": 0}

    def flaky(self, *a, **kw):
        calls["n"] += 1
        if calls["n"] == 1:
            raise hip.HipBackendError("injected failure")
        return real(self, *a, **kw)

    monkeypatch.setattr(hip.Index, "add_texts", flaky)
    with pytest.raises(hip.HipBackendError, match="injected"):
        g.compile("alpha w3", top_k=5)
    for q in ("alpha w3", "beta", "gamma", ""):
        assert g.compile(q, top_k=60) == og.compile(q, top_k=60)
    for gg in (g, og):
        gg.claims["h03"].retracted = False     # comes back from its parked vector
    assert g.compile("alpha w3", top_k=60) == og.compile("alpha w3", top_k=60)
    with pytest.raises(ValueError, match="finite"):
        g._embedding_cache["h01"] = [float("nan")] * 64
    with pytest.raises(ValueError, match="finite"):
        g._embedding_cache["h01"] = [1e300] * 64
    assert g.compile("alpha", top_k=60) == og.compile("alpha", top_k=60)


def test_in_place_tag_edits_reach_the_device_mask(hip):
    """Claim.tags edited in place (append / slice assignment) must update the fused tag mask."""
    from lbm_amd.context_graph import ContextGraph
    g, og = ContextGraph(), orc.OracleGraph()
    for gg in (g, og):
        gg.add_claim("a1", "alpha beta", ["x"], created_ms=1)
        gg.add_claim("a2", "alpha gamma", [], created_ms=2)
        gg.add_claim("a3", "alpha delta", ["y"], created_ms=3)
    assert g.compile("alpha", tags=["x"]) == og.compile("alpha", tags=["x"])
    for gg in (g, og):
        gg.claims["a2"].tags.append("x")
        gg.claims["a1"].tags[:] = ["z"]
        gg.claims["a3"].tags.remove("y")
    for tags in (["x"], ["y"], ["z"], ["x", "z"]):
        assert g.compile("alpha", tags=tags) == og.compile("alpha", tags=tags), tags
    assert g.compile("alpha") == og.compile("alpha")
    assert g.claims["a2"].to_dict()["tags"] == ["x"] and type(g.claims["a2"].to_dict()["tags"]) is list


def test_sharded_search_behind_the_c_abi_single_rank(hip):
    """lbh_comm_* + lbh_index_search_sharded with a one-rank communicator on the one GPU of this box: the whole call
    sequence of the multi-GPU query (select kernel writes the packed list to device memory, RCCL all-gather on the
    library's stream, publish to mapped host memory, merge) must return exactly what lbh_index_search_shard returns --
    fast path, the gated exact path (all-ties query), filters, top_k beyond one page, an empty shard."""
    from lbm_amd import synth
    ch = synth.make_chunk(0, 60_000, seed=9, real_sha_keys=False, dup_frac=0.01)
    ix = hip.Index()
    ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    ix.retract_many(ids[ch.retracted])
    comm = hip.Comm(hip.comm_unique_id(), 0, 1)
    try:
        qv = hip.embed_batch(synth.make_queries(5, seed=3) + [ch.text(77), ""])
        p50 = int(ch.created_ms[ch.n // 2])
        st = hip.Stats()
        for smode in (0, 1):
            hip.set_option("scan_mode", smode)
            for q in qv:
                for k, since, tm in ((8, hip.SINCE_NONE, 0), (64, p50, 0), (1, hip.SINCE_NONE, 0xF0), (200, hip.SINCE_NONE, 0), (0, p50, 0)):
                    wi, ws, wk = ix.search_shard(q, k, since, tm)
                    gi, gs, gk = ix.search_sharded(comm, q, k, since, tm, st)
                    assert gi.tolist() == wi.tolist(), (smode, k, since, tm)   # world 1, rank 0: global id == local id
                    assert gs.tobytes() == ws.tobytes() and gk.tobytes() == wk.tobytes()
                    gi2, _, _ = ix.search_sharded(comm, q, k, since, tm, None, gid_mul=8, gid_add=3)
                    assert gi2.tolist() == (wi * 8 + 3).tolist()
        hip.set_option("scan_mode", 0)
        empty = hip.Index()
        gi, gs, gk = empty.search_sharded(comm, qv[0], 8)
        assert gi.shape[0] == 0
        empty.close()
    finally:
        hip.set_option("scan_mode", 0)
        comm.close()
        ix.close()


def test_sharded_context_graph_on_the_gpu_single_rank(hip, golden_synth):
    """ShardedContextGraph with the real backend: a one-rank RCCL communicator behind the C ABI, device K1 for the query,
    lbh_index_search_sharded for the ranking -> the reference's golden answers."""
    from lbm_amd.sharded import ShardedContextGraph
    d = golden_synth
    comm = hip.Comm(hip.comm_unique_id(), 0, 1)
    try:
        g = ShardedContextGraph(0, 1, comm=comm)
        for i in range(d["n"]):
            tags = ["t%02d" % j for j in range(64) if (d["tagmask"][i] >> j) & 1]
            g.add_claim(d["hashes"][i], d["texts"][i], tags, created_ms=d["created_ms"][i])
        for i in range(d["n"]):
            if d["retracted"][i]:
                g.retract(d["hashes"][i])
        for r in d["results"][::3]:
            kw = {"top_k": r["top_k"]}
            if r["since_ms"] is not None:
                kw["since_ms"] = r["since_ms"]
            text, hs = g.compile(r["query"], **kw)
            assert hs == r["hashes"], (r["query"], kw)
            assert hashlib.sha256(text.encode()).hexdigest() == r["context_sha256"]
            assert hexes(g.compile_scored(r["query"], **kw)[1]) == r["scores"]
    finally:
        comm.close()


def test_node_level_batched_entry_and_lazy_formatting(hip):
    """RetrievalNode.compile_context_many == [compile_context(q) ...] (K3 behind it), and lazy=True hands
    the hashes back before the context text exists; the text, once read, is what the eager call returned."""
    from lbm_amd import synth
    from lbm_amd.context_graph import LazyContext
    from lbm_amd.node import NodeError, RetrievalNode
    ch = synth.make_chunk(0, 6000, seed=21, dup_frac=0.02)
    node = RetrievalNode()
    gid = node.create_group("g")
    g = node.groups[gid].graph
    texts, hashes = ch.texts(), ch.hashes()
    for i in range(ch.n):
        g.add_claim(hashes[i], texts[i], ch.tags(i), created_ms=int(ch.created_ms[i]))
    qs = synth.make_queries(20, seed=5) + [texts[9], "", "nothing matches this"]
    since = int(ch.created_ms[3000])
    for kw in ({}, {"top_k": 16, "since_ms": since}, {"top_k": 3}, {"top_k": 0}):
        want = [node.compile_context(gid, q, **kw) for q in qs]
        assert node.compile_context_many(gid, qs, **kw) == want
        lazy = node.compile_context_many(gid, qs, lazy=True, **kw)
        assert all(isinstance(x, LazyContext) and x._text is None for x in lazy)
        assert [x.claim_hashes for x in lazy] == [w[1] for w in want]
        g.claims[hashes[0]].text = "edited after the query"  # a lazy text shows the claims as they were at query time
        assert [tuple(x) for x in lazy] == want
        g.claims[hashes[0]].text = texts[0]
    one = node.compile_context(gid, qs[0], lazy=True)
    assert isinstance(one, LazyContext) and (one.context, one.claim_hashes) == node.compile_context(gid, qs[0])
    text, hs = one
    assert (text, hs) == node.compile_context(gid, qs[0])
    with pytest.raises(NodeError, match="unknown group_id nope"):
        node.compile_context_many("nope", ["q"])


def test_raw_key_mode_and_rank_key_fallback(hip):
    """Short free-form hashes (the reference's tests: "old", "new", "h1") take order-preserving raw keys -- no per-sync
    re-ranking; anything else (over 32 bytes, NUL inside, mixed with CAS hashes) falls back to rank keys.  Ties on score
    must break exactly like Python's str order in every mode."""
    from lbm_amd.context_graph import ContextGraph
    dup = "the same exact text appears many times"
    cases = {
        "raw": ["zzz", "Zzz", "a", "aa", "é", "0", "00", "claim_0002", "claim_0001", "~", "a" * 32, "日本", "", "a b"],
        "rank_long": ["zzz", "a" * 33, "a" * 32, "b"],
        "rank_nul": ["a", "a\x00", "a\x00\x00", "b"],
        "rank_mixed": ["f" * 64, "zz", "0" * 64, "a"],
        "hex": ["f" * 64, "0" * 63 + "1", "a" * 64, "b" * 64],
    }
    for name, keys in cases.items():
        g, og = ContextGraph(), orc.OracleGraph()
        for i, h in enumerate(keys):
            for gg in (g, og):
                gg.add_claim(h, dup if i % 3 else "something else", [], created_ms=i)
            if i == 1:
                assert g.compile("same text", top_k=10) == og.compile("same text", top_k=10)  # a query between the adds
        for q in ("same text", "", "else"):
            assert g.compile(q, top_k=20) == og.compile(q, top_k=20), (name, q)
        assert g._key_mode == name.split("_")[0], (name, g._key_mode)
        for gg in (g, og):
            gg.retract(keys[0])
            gg.add_claim("late" if name != "hex" else "c" * 64, dup, [], created_ms=99)
        assert g.compile("same text", top_k=20) == og.compile("same text", top_k=20), name


def test_rank_keys_are_maintained_incrementally(hip, monkeypatch):
    """Rank-key mode (hashes that are neither 64-hex nor <= 32 bytes): a publish costs one key between its neighbours in
    str order, not a re-sort and a rewrite of the key table.  Many exact score ties (identical texts), so the answer is
    decided by the keys; checked against the oracle graph while claims come and go."""
    import random
    from lbm_amd import _hip
    from lbm_amd.context_graph import ContextGraph
    calls = []
    real = _hip.Index.set_keys
    monkeypatch.setattr(_hip.Index, "set_keys", lambda self, ids, keys: (calls.append(len(ids)), real(self, ids, keys))[1])
    rnd = random.Random(11)
    g, og = ContextGraph(), orc.OracleGraph()
    texts = ["alpha beta", "alpha beta", "alpha beta gamma", "gamma", ""]
    known = []

    def new_hash():
        return "".join(rnd.choice("abAB-/~ 0é") for _ in range(rnd.randint(33, 50))) + "#%d" % len(known)

    for step in range(600):
        r = rnd.random()
        if r < 0.6 or not known:
            h = new_hash()
            known.append(h)
            t, ms = rnd.choice(texts), rnd.randint(0, 9)
            g.add_claim(h, t, [], created_ms=ms); og.add_claim(h, t, [], created_ms=ms)
        elif r < 0.75:
            h = rnd.choice(known)
            g.retract(h); og.retract(h)
        elif r < 0.8:
            h = rnd.choice(known)
            if h in og.claims:
                g.claims[h].created_ms = 5; og.claims[h].created_ms = 5
        else:
            kw = {"top_k": rnd.choice([1, 8, 30, 100])}
            if rnd.random() < 0.3:
                kw["since_ms"] = rnd.randint(0, 9)
            q = rnd.choice(["alpha", "gamma beta", ""])
            assert g.compile(q, **kw) == og.compile(q, **kw), (step, q, kw)
    assert g._key_mode == "rank"
    assert calls == [], calls  # an empty graph enters rank mode without a relabel; every publish placed its own key
    # a gap that runs out of midpoints triggers one relabel and nothing else changes
    base = "m" * 40
    for i in range(300):
        h = base + "m" * (i + 1)
        g.add_claim(h, "alpha beta", [], created_ms=1); og.add_claim(h, "alpha beta", [], created_ms=1)
        if i % 50 == 49:
            assert g.compile("alpha", top_k=64) == og.compile("alpha", top_k=64), i
    assert 1 <= len(calls) <= 3, calls


def test_group_graph_save_load_and_rebuild(hip, tmp_path, golden_snapshot):
    """lb/group.py:30-55 (graph half) and lb/node.py:486-515: graph.json round trip and the rebuild from a resolved
    transaction stream, both ending in one bulk K1 launch, against the reference's reloaded-graph results."""
    import json as _json
    from lbm_amd.node import Group, RetrievalNode
    case = golden_snapshot[0]
    g = _replay(case["ops"])
    grp = Group("g1", g)
    path = grp.save_graph(str(tmp_path))
    assert _json.load(open(path)) == case["snapshot"]
    loaded = Group.load_graph("g1", str(tmp_path))
    for r in case["results"]:
        kw = {k: r[k] for k in ("top_k", "since_ms") if k in r}
        assert loaded.graph.compile(r["query"], **kw) == (r["context"], r["hashes"])
    assert Group.load_graph("g2", str(tmp_path / "missing")).graph.compile("x") == ("Context slice\nQuery: x\n", [])
    # rebuild: the same ops as a transaction stream (claims resolved to their objects, retracts by hash)
    node = RetrievalNode()
    node.create_group("g1")
    txs, og = [], orc.OracleGraph()
    for op in case["ops"]:
        if op[0] == "add":
            txs.append({"type": "claim", "artifact_hash": op[1], "object": {"text": op[2], "tags": op[3], "created_ms": op[4],
                                                                            "parent_hash": op[5] if len(op) > 5 else None}})
            og.add_claim(op[1], op[2], op[3], created_ms=op[4])
        elif op[0] == "retract":
            txs.append({"type": "retract", "claim_hash": op[1]})
            og.retract(op[1])
    txs.append({"type": "claim", "artifact_hash": "missing-from-cas"})  # no object at all: cas.has(h) is False, skipped
    txs.append({"type": "vote", "x": 1})
    # lb/node.py:498-509: `cas.get_json(h) or {}` makes a falsy object an EMPTY claim; anything that raises while the
    # claim is rebuilt (a truthy non-dict, created_ms = Infinity, tags that cannot be listed) skips that claim only
    e1, e2, bad = "e" * 64, "f" * 64, "b" * 63
    txs.append({"type": "claim", "artifact_hash": e1, "object": None})
    og.add_claim(e1, "", [], created_ms=0)
    txs.append({"type": "claim", "artifact_hash": e2, "object": {"text": 12, "created_ms": None}})
    og.add_claim(e2, "12", [], created_ms=0)
    txs.append({"type": "claim", "artifact_hash": bad + "1", "object": ["not", "a", "dict"]})
    txs.append({"type": "claim", "artifact_hash": bad + "2", "object": {"text": "gamma", "created_ms": float("inf")}})
    txs.append({"type": "claim", "artifact_hash": bad + "3", "object": {"text": "gamma", "tags": 7}})
    txs.append({"type": "claim", "artifact_hash": 17, "object": {"text": "gamma"}})
    old = node.groups["g1"].graph
    node.rebuild_group_graph("g1", txs)
    assert node.groups["g1"].graph is not old
    assert set(node.groups["g1"].graph.claims) == set(og.claims)
    for q in ("beta", "gamma delta", "", "12"):
        assert node.compile_context("g1", q, top_k=20) == og.compile(q, top_k=20)


def test_ingest_from_pinned_buffers_equals_pageable(hip):
    """lbh_host_alloc + lbh_index_add: the sliced, overlapped upload (pinned arrays, several slices per call, keys decoded
    on the device, one synchronisation) must leave exactly the rows the plain path leaves -- embeddings against the
    oracle, and identical search answers; offsets that do not start at 0 and batches below one slice included."""
    from lbm_amd import synth
    ch = synth.make_chunk(0, 300_000, seed=4, real_sha_keys=False)
    a, b = hip.Index(), hip.Index()
    pins = [hip.pinned_like(x) for x in (ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)]
    ids = a.add_texts(*(p.a for p in pins))
    assert ids.tolist() == list(range(ch.n))
    # the same claims in three calls of different sizes, the later ones with offsets that start inside the blob
    for lo, hi in ((0, 1000), (1000, 200_000), (200_000, ch.n)):
        b.add_texts(ch.blob, ch.offsets[lo: hi + 1], ch.created_ms[lo:hi], ch.tagmask[lo:hi], ch.keys[lo:hi])
    rng = np.random.default_rng(0)
    pick = rng.choice(ch.n, size=3000, replace=False)
    want = orc.embed_batch([ch.text(int(i)) for i in pick])
    assert a.get_embeddings(ids[pick]).tobytes() == want.tobytes()
    assert b.get_embeddings(ids[pick]).tobytes() == want.tobytes()
    qv = hip.embed_batch(synth.make_queries(6, seed=8) + [""])
    for q in qv:
        for k, since, tm in ((8, hip.SINCE_NONE, 0), (64, int(ch.created_ms[150_000]), 0xFF)):
            ra, rb = a.search_shard(q, k, since, tm), b.search_shard(q, k, since, tm)
            assert ra[0].tolist() == rb[0].tolist() and ra[1].tobytes() == rb[1].tobytes() and ra[2].tobytes() == rb[2].tobytes()
    # the ordering keys are the raw key bytes: ties on an all-zero query come back in ascending key order
    ia, _, ka = a.search_shard(np.zeros(64), 32)
    assert [bytes(k) for k in ka] == sorted(bytes(k) for k in ka) and [bytes(ch.keys[i]) for i in ia.tolist()] == [bytes(k) for k in ka]
    for p in pins:
        p.close()
    a.close(); b.close()
