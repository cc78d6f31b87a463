"""get_recent_claims (lb/node.py:736-762) -- time filter + newest-first + limit.

lb/node.py does not import in this image (its `cryptography` dependency is absent), so no golden vectors
could be generated from the reference for this row.  The oracle (oracle.lb_oracle.recent_claims) restates
lines 750-762 and is pinned by replaying the expectations of the reference's own tests:
tests/test_edge_cases.py:560-598 (TestTimeQueryEdgeCases) and tests/test_multi_agent.py:170-193.
The GPU path is then compared with the oracle entry by entry (integer/index work: exact equality).
"""
import random

import numpy as np
import pytest

from oracle import lb_oracle as orc


def _oracle_node():
    """The slice of a node the reference tests touch: publish (now-stamped add) + get_recent_claims."""
    class N:
        def __init__(self):
            self.g = orc.OracleGraph()
            self.t = 1_700_000_000_000
            self.i = 0

        def publish(self, text, tags, dt=1):
            self.t += dt
            self.i += 1
            self.g.add_claim("h%04d" % self.i, text, tags, created_ms=self.t)

        def recent(self, since_ms, limit=100):
            return orc.recent_claims(self.g.claims, since_ms, limit)
    return N()


# ---- oracle pinned by the reference's own test expectations (no GPU) ---------------------------------

def test_oracle_reference_time_query_edge_cases():
    n = _oracle_node()
    n.publish("Claim 1", ["test"]); n.publish("Claim 2", ["test"])
    assert len(n.recent(0)) == 2                    # test_since_ms_zero
    assert len(n.recent(-1000)) == 2                # test_since_ms_negative: all claims
    assert len(n.recent(n.t + 1_000_000)) == 0      # test_since_ms_future
    assert len(n.recent(0, limit=0)) == 0           # test_limit_zero
    one = _oracle_node(); one.publish("Claim 1", ["test"])
    assert len(one.recent(0, limit=-1)) == 0        # test_limit_negative (one claim: [:-1] is empty)


def test_oracle_reference_multi_agent_cases():
    n = _oracle_node()
    n.publish("Claim 1", ["test"], dt=10)
    cutoff = n.t + 10
    n.publish("Claim 2", ["test"], dt=20); n.publish("Claim 3", ["test"])
    recent = n.recent(cutoff)                       # test_get_recent_claims
    assert len(recent) == 2 and all(c["created_ms"] >= cutoff for c in recent)
    assert [c["text"] for c in recent] == ["Claim 3", "Claim 2"]   # docstring: sorted by created_ms descending
    assert set(recent[0]) == {"claim_hash", "text", "tags", "created_ms", "parent_hash"}
    m = _oracle_node()
    for i in range(10):
        m.publish(f"Claim {i}", ["test"])
    assert len(m.recent(0, limit=5)) == 5           # test_get_recent_claims_respects_limit


def test_oracle_ties_keep_dict_order_and_arrays_agree():
    g = orc.OracleGraph()
    for i, ms in enumerate([5, 7, 5, 7, 7, 1, 5]):
        g.add_claim("k%d" % i, "t", [], created_ms=ms)
    g.retract("k3")
    g.add_claim("k0", "again", [], created_ms=7)    # overwrite keeps k0's FIRST position in the dict
    assert [c["claim_hash"] for c in orc.recent_claims(g.claims, 5)] == ["k0", "k1", "k4", "k2", "k6"]
    assert [c["claim_hash"] for c in orc.recent_claims(g.claims, 5, -2)] == ["k0", "k1", "k4"]
    rng = np.random.default_rng(3)
    created = rng.integers(-4, 4, size=500)
    order = rng.permutation(500).astype(np.uint64)
    for since, limit in [(-10, 500), (0, 17), (3, 1000), (4, 5)]:
        exp = sorted([i for i in range(500) if created[i] >= since], key=lambda i: (-created[i], order[i]))[:limit]
        assert orc.recent_arrays(created, order, since, limit).tolist() == exp


# ---- GPU path vs the oracle ---------------------------------------------------------------------------

@pytest.fixture(scope="module")
def hip():
    from lbm_amd import _hip
    _hip.context()
    return _hip


def _pair():
    from lbm_amd.node import RetrievalNode
    node = RetrievalNode()
    gid = node.create_group("g")
    return node, gid, node.groups[gid].graph, orc.OracleGraph()


@pytest.mark.gpu
def test_reference_cases_on_gpu(hip):
    from lbm_amd.node import NodeError
    node, gid, g, og = _pair()
    with pytest.raises(NodeError, match="unknown group_id nope"):
        node.get_recent_claims("nope", 0)
    assert node.get_recent_claims(gid, 0) == []
    t = 1_700_000_000_000
    for i in range(10):
        g.add_claim("c%d" % i, f"Claim {i}", ["test"], created_ms=t + 3 * i, parent_hash=("c0" if i == 4 else None))
        og.add_claim("c%d" % i, f"Claim {i}", ["test"], created_ms=t + 3 * i, parent_hash=("c0" if i == 4 else None))
    for since, kw in [(0, {}), (-1000, {}), (t + 10 ** 9, {}), (0, {"limit": 0}), (0, {"limit": -1}), (0, {"limit": 5}),
                      (t + 9, {}), (t + 10, {"limit": 3}), (0, {"limit": -4}), (0, {"limit": None}), (float(t) + 8.5, {}),
                      (float("nan"), {}), (float("inf"), {}), (float("-inf"), {"limit": 2}), (2 ** 63, {}), (-2 ** 70, {})]:
        assert node.get_recent_claims(gid, since, **kw) == orc.recent_claims(og.claims, since, **kw), (since, kw)
    assert len(node.get_recent_claims(gid, 0, limit=5)) == 5
    with pytest.raises(TypeError):
        node.get_recent_claims(gid, 0, limit=2.5)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_randomised_graph_vs_oracle(hip, seed):
    rnd = random.Random(seed)
    node, gid, g, og = _pair()
    known = []
    checks = 0
    for step in range(900):
        r = rnd.random()
        if r < 0.5 or not known:
            h = "h%d" % step
            ms = rnd.randint(0, 12)  # few distinct timestamps: ties everywhere
            g.add_claim(h, "t%d" % step, ["a"], created_ms=ms); og.add_claim(h, "t%d" % step, ["a"], created_ms=ms)
            known.append(h)
        elif r < 0.6:  # overwrite: keeps the dict position, gets a new device row
            h = rnd.choice(known)
            ms = rnd.randint(0, 12)
            g.add_claim(h, "again", [], created_ms=ms); og.add_claim(h, "again", [], created_ms=ms)
        elif r < 0.7:
            h = rnd.choice(known)
            g.retract(h); og.retract(h)
        elif r < 0.74:
            h = rnd.choice(known)
            if h in og.claims:
                flag = not og.claims[h].retracted
                g.claims[h].retracted = flag; og.claims[h].retracted = flag
        elif r < 0.78:
            h = rnd.choice(known)
            if h in og.claims:
                ms = rnd.randint(0, 12)
                g.claims[h].created_ms = ms; og.claims[h].created_ms = ms
        elif r < 0.8:  # delete and re-insert: moves to the END of the dict order
            h = rnd.choice(known)
            if h in og.claims:
                c, oc = g.claims.pop(h), og.claims.pop(h)
                g.claims[h] = c; og.claims[h] = oc
        else:
            since = rnd.randint(-1, 13)
            limit = rnd.choice([0, 1, 2, 5, 63, 64, 65, 100, 130, 1000, None, -1, -7])
            got = node.get_recent_claims(gid, since, limit=limit)
            assert got == orc.recent_claims(og.claims, since, limit), (step, since, limit)
            checks += 1
            if rnd.random() < 0.3:  # the two query kinds share the mirror
                assert g.compile("t1 again", top_k=5, since_ms=since) == og.compile("t1 again", top_k=5, since_ms=since)
    assert checks > 100


def _index_case(hip, n, seed, distinct):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, 64))
    created = rng.integers(-distinct, distinct, size=n).astype(np.int64) if distinct else np.full(n, 42, dtype=np.int64)
    keys = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ix = hip.Index()
    ids = ix.add_vectors(v, created, np.zeros(n, dtype=np.uint64), keys)
    return ix, ids, created, rng


@pytest.mark.gpu
def test_index_recent_vs_oracle_300k(hip):
    n = 300_000
    ix, ids, created, rng = _index_case(hip, n, 7, 2000)
    order = np.arange(n, dtype=np.uint64)  # default order = claim id
    st = hip.Stats()
    for since, limit in [(hip.SINCE_NONE, 1), (hip.SINCE_NONE, 64), (hip.SINCE_NONE, 200), (0, 5000), (1990, 100_000),
                         (1999, 10), (2000, 10), (-2000, 129)]:
        got, cr, matched = ix.recent(since, limit, st)
        exp = orc.recent_arrays(created, order, since, limit)
        assert got.tolist() == ids[exp].tolist(), (since, limit)
        assert cr.tolist() == created[exp].tolist()
        assert matched == int(np.count_nonzero(created >= since))
        pages = max(1, -(-min(limit, matched) // 64))
        assert st.path == 3 and st.bytes_per_row == 8 and st.scan_launches in (pages, pages + 1)
    # explicit order stamps (a permutation) replace the default tie-break
    perm = rng.permutation(n).astype(np.uint64)
    ix.set_order(ids, perm)
    for since, limit in [(hip.SINCE_NONE, 64), (1500, 777)]:
        got, _, _ = ix.recent(since, limit)
        assert got.tolist() == ids[orc.recent_arrays(created, perm, since, limit)].tolist()
    # swap-deletes move rows: stamps must travel with them
    dead = rng.choice(n, size=5000, replace=False)
    for i in dead.tolist():
        ix.retract(int(ids[i]))
    live = np.ones(n, dtype=bool); live[dead] = False
    lidx = np.nonzero(live)[0]
    for since, limit in [(hip.SINCE_NONE, 300), (1900, 64)]:
        got, _, matched = ix.recent(since, limit)
        exp = lidx[orc.recent_arrays(created[lidx], perm[lidx], since, limit)]
        assert got.tolist() == ids[exp].tolist()
        assert matched == int(np.count_nonzero(created[lidx] >= since))
    ix.close()


@pytest.mark.gpu
def test_index_recent_mass_ties(hip):
    """Every claim shares one timestamp: the answer is purely the insertion order, across pages."""
    n = 250_000
    ix, ids, created, rng = _index_case(hip, n, 9, 0)
    got, _, matched = ix.recent(hip.SINCE_NONE, 100)
    assert got.tolist() == ids[:100].tolist() and matched == n
    got, _, _ = ix.recent(42, 3000)       # 47 pages, cursor inside one huge tie group
    assert got.tolist() == ids[:3000].tolist()
    assert ix.recent(43, 10)[0].tolist() == []
    perm = rng.permutation(n).astype(np.uint64)
    ix.set_order(ids, perm)
    inv = np.argsort(perm)
    got, _, _ = ix.recent(hip.SINCE_NONE, 500)
    assert got.tolist() == ids[inv[:500]].tolist()
    # ascending timestamps in scan order (the adversarial direction for the streaming threshold)
    ix2 = hip.Index()
    cr2 = np.arange(n, 0, -1, dtype=np.int64)  # newest rows FIRST in the index: the backward scan sees old rows first
    ids2 = ix2.add_vectors(rng.standard_normal((n, 64)), cr2, np.zeros(n, dtype=np.uint64),
                           rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
    got, cr, _ = ix2.recent(hip.SINCE_NONE, 200)
    assert got.tolist() == ids2[:200].tolist() and cr.tolist() == cr2[:200].tolist()
    ix.close(); ix2.close()
