"""N > 1 path on CPU: two gloo ranks each own a shard, the all-gather merge must return exactly the
single-corpus answer (ids, fp64 score bits, order).  The rank-local search is played by the CPU oracle
here (test infrastructure); on the GPU box the same ShardedSearcher wraps Index.search_shard."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_q):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from lbm_amd import synth
    from lbm_amd.sharded import ShardedSearcher
    from oracle import lb_oracle as orc
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ch = synth.make_chunk(0, n, seed=31, dup_frac=0.02)
    emb = orc.embed_batch_packed(ch.blob, ch.offsets)
    mine = np.arange(rank, n, world)            # claims dealt round-robin: global id = local * world + rank
    keys_hex = ch.hashes()
    my_keys = [keys_hex[i] for i in mine]
    retr = ch.retracted.astype(np.uint8)

    def local_search(q, k, since_ms, tag_mask):
        idx, sc = orc.compile_arrays(emb[mine], ch.created_ms[mine], retr[mine], my_keys, q, k,
                                     None if since_ms is None else since_ms, ch.tagmask[mine], tag_mask)
        return idx.astype(np.uint64), sc, ch.keys[mine][idx]

    s = ShardedSearcher(local_search, rank, world)
    results = []
    for qtext in synth.make_queries(5, seed=3) + [ch.text(10), ""]:
        q = orc.embed_array(qtext)
        for k, since in ((8, None), (64, int(ch.created_ms[n // 2])), (3, None)):
            gids, sc, kw = s.search(q, k, since, 0)
            want_idx, want_sc = orc.compile_arrays(emb, ch.created_ms, retr, keys_hex, q, k, since)
            results.append((gids.tolist() == want_idx.tolist(), sc.tobytes() == want_sc.tobytes()))
    dist.barrier()
    dist.destroy_process_group()
    out_q.put((rank, results))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 5])
def test_allgather_merge_equals_single_corpus(world):
    n = 3000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, results in got:
        assert len(results) == 21
        assert all(a and b for a, b in results), (rank, results)


@pytest.mark.parametrize("which", ["numpy", "c_abi"])
def test_merge_orders_ties_by_key_and_handles_short_lists(which):
    from lbm_amd import sharded
    from lbm_amd.sharded import pack_local
    merge_gathered = sharded.merge_gathered if which == "numpy" else sharded.merge_entries
    k = 4
    keys = np.zeros((3, 32), dtype=np.uint8)
    keys[0, 0], keys[1, 0], keys[2, 0] = 9, 1, 5
    a = pack_local(np.array([0.5, 0.5, 0.25]), np.array([10, 11, 12], dtype=np.uint64), keys, k)
    keys_b = np.zeros((1, 32), dtype=np.uint8)
    keys_b[0, 0] = 3
    b = pack_local(np.array([0.5]), np.array([20], dtype=np.uint64), keys_b, k)
    c = pack_local(np.zeros(0), np.zeros(0, dtype=np.uint64), np.zeros((0, 32), dtype=np.uint8), k)
    ids, sc, kw = merge_gathered(np.stack([a, b, c]), k)
    assert ids.tolist() == [11, 20, 10, 12] and sc.tolist() == [0.5, 0.5, 0.5, 0.25]
    ids, sc, _ = merge_gathered(np.stack([c, c]), k)
    assert ids.shape == (0,)
    # -0.0 and +0.0 tie exactly like Python's (-s, h) sort key
    z = pack_local(np.array([-0.0]), np.array([1], dtype=np.uint64), keys[:1], k)
    y = pack_local(np.array([0.0]), np.array([2], dtype=np.uint64), keys[1:2], k)
    ids, _, _ = merge_gathered(np.stack([z, y]), k)
    assert ids.tolist() == [2, 1]


def test_c_merge_equals_numpy_merge_on_random_lists():
    """lbh_merge_topk (host code of the C ABI) against the numpy merge: ties on score, short and empty lists."""
    from lbm_amd.sharded import merge_entries, merge_gathered, pack_local
    rng = np.random.default_rng(17)
    for trial in range(60):
        k = int(rng.choice([1, 3, 8, 64]))
        world = int(rng.integers(1, 9))
        lists = []
        for r in range(world):
            n = int(rng.integers(0, k + 1))
            sc = np.sort(rng.choice([0.0, -0.0, 0.25, 0.5, 0.75, -0.5], size=n))[::-1].copy()
            keys = rng.integers(0, 3, size=(n, 32), dtype=np.uint8)  # few distinct bytes: shared key prefixes
            keys[:, 31] = rng.integers(0, 256, size=n)
            lists.append(pack_local(sc, rng.integers(0, 2 ** 63, size=n).astype(np.uint64), keys, k))
        g = np.stack(lists)
        a, b = merge_gathered(g, k), merge_entries(g, k)
        assert a[1].tobytes() == b[1].tobytes() or np.array_equal(a[1], b[1])  # scores (+-0.0 may differ in bits only on ties)
        assert np.array_equal(a[2], b[2])  # key words: the order of the reference


def _graph_worker(rank, world, port, out_q):
    sys.path.insert(0, REPO)
    import hashlib
    import json
    import torch.distributed as dist
    from lbm_amd.sharded import ShardedContextGraph, owner_of
    from oracle import lb_oracle as orc
    from tests.oracle_index import OracleIndex
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ok = []
    d = json.load(open(os.path.join(REPO, "tests", "golden", "synth2k.json")))["data"]
    g = ShardedContextGraph(rank, world, _index_factory=OracleIndex, _embed=orc.embed_array)
    for i in range(d["n"]):
        tags = ["t%02d" % j for j in range(64) if (d["tagmask"][i] >> j) & 1]
        g.add_claim(d["hashes"][i], d["texts"][i], tags, created_ms=d["created_ms"][i])
    for i in range(d["n"]):
        if d["retracted"][i]:
            g.retract(d["hashes"][i])
    for r in d["results"][::5]:  # the reference's own answers: hashes, fp64 score bits, context text
        kw = {"top_k": r["top_k"]}
        if r["since_ms"] is not None:
            kw["since_ms"] = r["since_ms"]
        text, hs = g.compile(r["query"], **kw)
        _, sc = g.compile_scored(r["query"], **kw)
        ok.append(hs == r["hashes"] and hashlib.sha256(text.encode()).hexdigest() == r["context_sha256"]
                  and [float(x).hex() for x in sc] == r["scores"])
    owned = sum(owner_of(h, world) == rank for h, r in zip(d["hashes"], d["retracted"]) if not r)
    ok.append(g._index.size() == owned and 0 < owned < d["n"])  # only this rank's claims are on its device
    # ... and on its host: per-rank state is ~ N / world (Claim objects, id maps), not N
    owned_all = sum(owner_of(h, world) == rank for h in d["hashes"])
    ok.append(len(g.claims) == owned_all and len(g._id_of) == owned and len(g._meta) == owned and len(g._seq_of) == owned_all)
    ok.append(owned_all < 1.5 * d["n"] / world + 50)
    # the tag extension and a lazily formatted result go through the same exchange
    og = orc.OracleGraph()
    for i in range(0, 300):
        tags = ["t%02d" % j for j in range(64) if (d["tagmask"][i] >> j) & 1]
        og.add_claim(d["hashes"][i], d["texts"][i], tags, created_ms=d["created_ms"][i])
    g2 = ShardedContextGraph(rank, world, _index_factory=OracleIndex, _embed=orc.embed_array)
    for h, c in og.claims.items():
        g2.add_claim(h, c.text, c.tags, created_ms=c.created_ms)
    for tags in (["t03"], ["t10", "t11"], ["nobody"]):
        ok.append(g2.compile("w00001 w00002", top_k=16, tags=tags) == og.compile("w00001 w00002", top_k=16, tags=tags))
    lazy = g2.compile("w00003", top_k=4, lazy=True)
    ok.append(lazy.claim_hashes == og.compile("w00003", top_k=4)[1] and tuple(lazy) == og.compile("w00003", top_k=4))
    ok.append(g2.compile_many(["w00001", ""], top_k=3) == [og.compile("w00001", top_k=3), og.compile("", top_k=3)])
    # a batch goes through ONE exchange of all its lists (ShardedSearcher.search_many)
    qs = ["w00001 w00004", "w00002", "", "w00010 w00003 w00001", d["texts"][7]]
    for kw in ({"top_k": 8}, {"top_k": 70, "since_ms": d["created_ms"][150]}, {"top_k": 5, "tags": ["t03", "t07"]}, {"top_k": 0}):
        ok.append(g2.compile_many(qs, **kw) == [og.compile(q, **kw) for q in qs])
    lz = g2.compile_many(qs[:2], top_k=4, lazy=True)
    ok.append([tuple(x) for x in lz] == [og.compile(q, top_k=4) for q in qs[:2]])
    # the recency query (lb/node.py:736-762): merge of the ranks' newest-first lists; ties keep the dict order, which a
    # re-added hash does not change
    for h in (d["hashes"][5], d["hashes"][17]):
        og.retract(h); g2.retract(h)
    og.add_claim(d["hashes"][3], "re-added text", ["t01"], created_ms=d["created_ms"][40])
    g2.add_claim(d["hashes"][3], "re-added text", ["t01"], created_ms=d["created_ms"][40])
    for i in (290, 291, 292):  # equal timestamps: the stable sort keeps insertion order
        og.add_claim("%064x" % i, "tie %d" % i, [], created_ms=d["created_ms"][100])
        g2.add_claim("%064x" % i, "tie %d" % i, [], created_ms=d["created_ms"][100])
    for since, limit in ((0, 100), (d["created_ms"][100], 7), (d["created_ms"][100], 1000), (d["created_ms"][250], None),
                         (0, -290), (0, 0), (10 ** 18, 5)):
        want = orc.recent_claims(og.claims, since, limit)
        ok.append(g2.recent(since, limit) == [w["claim_hash"] for w in want] and g2.recent_claims(since, limit) == want)
    ok.append(g2.snapshot() == og.snapshot())
    g4 = ShardedContextGraph.from_snapshot_sharded(og.snapshot(), rank, world, _index_factory=OracleIndex, _embed=orc.embed_array)
    ok.append(g4.compile("w00001 w00002", top_k=16) == og.compile("w00001 w00002", top_k=16) and g4.snapshot() == og.snapshot()
              and g4.recent(0, 9) == [w["claim_hash"] for w in orc.recent_claims(og.claims, 0, 9)])
    # failure containment: rank 1's local search raises -> it still takes part in the exchange, EVERY rank raises for
    # that query (nobody waits inside the all-gather), and the next query works
    from lbm_amd.sharded import ShardPeerError
    real = g2._index.search_shard
    if rank == 1:
        def boom(*a, **k):
            raise RuntimeError("injected: local search failed on rank 1")
        g2._index.search_shard = boom
    for call in (lambda: g2.compile("w00001", top_k=4), lambda: g2.compile_many(qs, top_k=4)):
        try:
            call()
            ok.append(False)
        except RuntimeError as e:
            ok.append(isinstance(e, ShardPeerError) != (rank == 1))
    g2._index.search_shard = real
    ok.append(g2.compile("w00001", top_k=4) == og.compile("w00001", top_k=4))
    # short free-form hashes (the reference's tests use "old" / "new"): raw keys, still mergeable
    g3, og3 = ShardedContextGraph(rank, world, _index_factory=OracleIndex, _embed=orc.embed_array), orc.OracleGraph()
    for i, h in enumerate(["old", "new", "h1", "zzz", "Zzz", "a", "aa", "é", "claim_0002", "claim_0001"]):
        for gg in (g3, og3):
            gg.add_claim(h, "the same exact text" if i % 2 else "Old data", ["data"], created_ms=1000 * i)
    for q, kw in (("data", {"top_k": 10, "since_ms": 1500}), ("same text", {"top_k": 4}), ("", {"top_k": 10})):
        ok.append(g3.compile(q, **kw) == og3.compile(q, **kw))
    g3.add_claim("f" * 64, "a CAS hash among free-form ones", [], created_ms=5)
    try:
        g3.compile("data")
        ok.append(False)
    except ValueError:
        ok.append(True)  # raised on every rank alike, before any collective
    dist.barrier()
    dist.destroy_process_group()
    out_q.put((rank, ok))


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world", [2, 5])
def test_sharded_context_graph_equals_reference_golden(world):
    """Graph-level sharding (north_star: compile_context over a claim-sharded corpus returns hash-identical top-k):
    gloo ranks (two; five: uneven ownership, and shards of the ten-claim graph that are empty), each holding the claims it
    owns in an oracle-backed index, answer the reference's golden queries."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=200) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok in got:
        assert len(ok) > 30 and all(ok), (rank, [i for i, x in enumerate(ok) if not x])
