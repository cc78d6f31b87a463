"""GPU parity tests: the HIP path (through the C ABI / the Python mirror) against the golden vectors
generated from the reference and against the CPU oracle on seeded inputs.  Bit-exact everywhere:
embeddings, scores, hash lists, order, context text."""
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


def test_embed_golden_bit_exact(hip, golden_embed):
    from lbm_amd.latent import embed, embed_many, _tokenize
    texts = [c["text"] for c in golden_embed["embed"]]
    batch = embed_many(texts)
    for i, c in enumerate(golden_embed["embed"]):
        assert hexes(batch[i]) == c["embedding"], c["text"][:40]
        assert _tokenize(c["text"]) == c["tokens"]
    for c in golden_embed["embed"][:12]:
        e = embed(c["text"])
        assert isinstance(e, list) and len(e) == 64 and hexes(e) == c["embedding"]
    assert embed("hello")[0].hex() == "-0x1.2e0984b25fcebp-3"


def test_embed_random_texts_vs_oracle(hip):
    rng = np.random.default_rng(5)
    alphabet = list("abcXYZ019_ -.,!\n\té日") + ["😀"]
    texts = []
    for i in range(3000):
        n = int(rng.integers(0, 400))
        texts.append("".join(rng.choice(alphabet, size=n)))
    texts += ["a" * n for n in (53, 54, 55, 56, 57, 63, 64, 65, 119, 120, 121, 183, 184, 4000)]
    texts += [" ".join("tok%d" % j for j in range(n)) for n in (31, 32, 33, 63, 64, 65, 127, 128, 129, 1000)]
    got = hip.embed_batch(texts)
    want = orc.embed_batch(texts)
    assert got.tobytes() == want.tobytes()


def test_embed_group_and_head_boundaries_vs_oracle(hip):
    """K1 hashes the tokens of four consecutive claims in shared batches and assembles one-block messages a word at a
    time from each claim's first 128 bytes; anything else takes the byte-wise path.  Directed cases around both
    switches: tokens of every length 1..60 ending at byte 126..130 of their claim, upper case at every byte position,
    lengths that are multiples of four, empty claims and token-free claims between full ones, groups cut short by the
    end of the batch, and batches whose size is not a multiple of the group."""
    letters = "ABCDEFGHIJKLMNOPQRSTUVWXYZ_0123456789abcdefghijklmnopqrstuvwxyz"
    texts = []
    for L in range(1, 61):
        word = "".join(letters[(L + i) % len(letters)] for i in range(L))
        for end in (126, 127, 128, 129, 130):
            pad = end - L - 1
            if pad >= 1:
                texts.append("x" * pad + " " + word)          # the token ends exactly at byte `end`
        texts.append(word)                                      # alone, at the very start
        texts.append(word.swapcase() + " " + word.lower() + "!" + word.upper())
    for pos in range(24):                                       # one upper-case byte at every position of a 24-byte token
        t = list("abcdefghijklmnopqrstuvwx")
        t[pos] = t[pos].upper()
        texts.append("".join(t) + " z")
    texts += ["", " ", "!!!", "é日😀", "a", "", "B c", "", "", "", "", "d" * 55, "e" * 56, "", "F" * 54 + " g"]
    for n in (1, 2, 3, 4, 5, 7, 8, 9, len(texts)):
        got = hip.embed_batch(texts[:n])
        want = orc.embed_batch(texts[:n])
        assert got.tobytes() == want.tobytes(), n
    rev = texts[::-1]
    assert hip.embed_batch(rev).tobytes() == orc.embed_batch(rev).tobytes()


def test_cosine_golden(hip, golden_embed):
    from lbm_amd.latent import cosine, embed
    for c in golden_embed["cosine"]:
        assert cosine(embed(c["a"]), embed(c["b"])).hex() == c["cosine"]
    with pytest.raises(ValueError, match="dimension mismatch"):
        cosine([1.0, 2.0], [1.0])
    assert cosine([], []) == 0.0
    a = np.random.default_rng(1).standard_normal(300)
    b = np.random.default_rng(2).standard_normal(300)
    assert cosine(list(a), list(b)).hex() == orc.cosine(a, b).hex()


def _replay(ops):
    from lbm_amd.context_graph import Claim, ContextGraph
    g = ContextGraph()
    for op in ops:
        if op[0] == "add":
            g.add_claim(op[1], op[2], op[3], created_ms=op[4])
        elif op[0] == "retract":
            g.retract(op[1])
        elif op[0] == "raw_insert":
            g.claims[op[1]] = Claim(claim_hash=op[1], text=op[2], tags=list(op[3]), created_ms=op[4], evidence=[])
    return g


def test_compile_golden_cases(hip, golden_compile):
    for case in golden_compile:
        g = _replay(case["ops"])
        for r in case["results"]:
            kw = {k: r[k] for k in ("top_k", "since_ms") if k in r}
            text, hashes = g.compile(r["query"], **kw)
            assert hashes == r["hashes"], (case["name"], r["query"], kw)
            assert text == r["context"], (case["name"], r["query"], kw)
            hs, sc = g.compile_scored(r["query"], **kw)
            assert hexes(sc) == r["scores"], (case["name"], r["query"], kw)


def test_compile_golden_interleaved_queries(hip, golden_compile):
    """Queries between mutations: the device mirror must follow add / overwrite / retract incrementally."""
    from lbm_amd.context_graph import ContextGraph
    g, og = ContextGraph(), orc.OracleGraph()
    case = [c for c in golden_compile if c["name"] == "overwrite_retract_raw"][0]
    for op in case["ops"]:
        if op[0] == "add":
            g.add_claim(op[1], op[2], op[3], created_ms=op[4]); og.add_claim(op[1], op[2], op[3], created_ms=op[4])
        elif op[0] == "retract":
            g.retract(op[1]); og.retract(op[1])
        else:
            continue
        for q in ("beta", "gamma delta", ""):
            assert g.compile(q, top_k=5) == og.compile(q, top_k=5)


def test_compile_synth2k_golden(hip, golden_synth):
    from lbm_amd.context_graph import ContextGraph
    import hashlib
    d = golden_synth
    g = ContextGraph()
    for i in range(d["n"]):
        tags = ["t%02d" % j for j in range(64) if (d["tagmask"][i] >> j) & 1]
        g.add_claim(d["hashes"][i], d["texts"][i], tags, created_ms=d["created_ms"][i])
    for i in range(d["n"]):
        if d["retracted"][i]:
            g.retract(d["hashes"][i])
    for i_str, hx in d["embedding_samples"].items():
        h = d["hashes"][int(i_str)]
        if not d["retracted"][int(i_str)]:
            assert hexes(g._embedding_cache[h]) == hx
    for r in d["results"]:
        kw = {"top_k": r["top_k"]}
        if r["since_ms"] is not None:
            kw["since_ms"] = r["since_ms"]
        text, hs = g.compile(r["query"], **kw)
        assert hs == r["hashes"], (r["query"], kw)
        assert hashlib.sha256(text.encode()).hexdigest() == r["context_sha256"]
        _, sc = g.compile_scored(r["query"], **kw)
        assert hexes(sc) == r["scores"]


def test_node_compile_context_contract(hip):
    from lbm_amd.node import NodeError, RetrievalNode
    node = RetrievalNode()
    with pytest.raises(NodeError, match="unknown group_id nope"):
        node.compile_context("nope", "q")
    gid = node.create_group("g1")
    g = node.groups[gid].graph
    # tests/test_multi_agent.py:155-168 of the reference
    g.add_claim("old_claim", "Old claim", ["old"], created_ms=5000)
    g.add_claim("new_claim", "New claim about query", ["new"], created_ms=15000)
    context, hashes = node.compile_context(gid, "query", top_k=10, since_ms=5001)
    assert "new_claim" in hashes and "old_claim" not in hashes
    assert (context, hashes) == g.compile("query", top_k=10, since_ms=5001)
    assert context.startswith("Context slice\nQuery: query\n\n1. New claim about query  [tags: new]")


def test_cache_view_and_direct_mutation_semantics(hip):
    from lbm_amd.context_graph import Claim, ContextGraph
    g, og = ContextGraph(), orc.OracleGraph()
    for gg in (g, og):
        gg.add_claim("h1", "alpha beta", ["a"], created_ms=1)
        gg.add_claim("h2", "beta gamma", [], created_ms=2)
    assert "h1" in g._embedding_cache and "zz" not in g._embedding_cache
    assert hexes(g._embedding_cache["h1"]) == hexes(og._embedding_cache["h1"])
    assert len(g._embedding_cache) == 2
    # flipping the flag by hand keeps the cached vector (reference: cache untouched, claim skipped)
    g.claims["h1"].retracted = True; og.claims["h1"].retracted = True
    assert g.compile("beta") == og.compile("beta")
    assert "h1" in g._embedding_cache
    g.claims["h1"].text = "changed text is NOT re-embedded while cached"; og.claims["h1"].text = g.claims["h1"].text
    g.claims["h1"].retracted = False; og.claims["h1"].retracted = False
    assert g.compile("alpha") == og.compile("alpha")
    # created_ms edits are honoured at the next compile
    g.claims["h2"].created_ms = 100; og.claims["h2"].created_ms = 100
    assert g.compile("beta", since_ms=50) == og.compile("beta", since_ms=50)
    # retract() drops the vector; popping the cache forces a re-embed of the current text
    g.retract("h2"); og.retract("h2")
    assert "h2" not in g._embedding_cache
    g._embedding_cache.pop("h1"); og._embedding_cache.pop("h1")
    assert g.compile("changed") == og.compile("changed")
    del g.claims["h1"]; del og.claims["h1"]
    assert g.compile("changed") == og.compile("changed") == ("Context slice\nQuery: changed\n", [])


@pytest.mark.parametrize("seed,hex_hashes,smode", [(11, True, 0), (12, False, 0), (13, True, 0), (14, True, 1), (15, False, 1)])
def test_randomised_incremental_ops_vs_oracle(hip, seed, hex_hashes, smode):
    """publish / overwrite / retract / direct edits interleaved with queries.  The device
    mirror is updated incrementally (appends, swap-deletes, metadata pokes); after every mutation burst the
    ranking, scores and context text must equal the reference semantics (oracle graph, pinned by golden)."""
    import hashlib
    import random
    from lbm_amd.context_graph import ContextGraph
    rnd = random.Random(seed)
    hip.set_option("scan_mode", smode)  # 1: the fp16 copy has to follow appends, swap-deletes and regrowth
    vocab = ["alpha", "beta", "gamma", "delta", "eps", "zeta", "eta", "theta", "iota", "kappa", "lam", "mu",
             "nu", "xi", "omi", "pi", "rho", "sigma", "tau", "ups", "phi", "chi", "psi", "omega", "x1", "y_2"]
    tags_pool = ["a", "b", "c", "d"]
    g, og = ContextGraph(), orc.OracleGraph()
    known = []
    counter = 0

    def new_hash():
        nonlocal counter
        counter += 1
        if hex_hashes:
            return hashlib.sha256(b"claim%d-%d" % (seed, counter)).hexdigest()
        return rnd.choice(["z", "A", "m", "0", "~"]) * rnd.randint(1, 3) + "-%d" % rnd.randint(0, 10 ** 6) + "/%d" % counter

    def text():
        return " ".join(rnd.choice(vocab) for _ in range(rnd.randint(0, 6)))

    n_queries = 0
    for step in range(700):
        r = rnd.random()
        if r < 0.45 or not known:
            h = new_hash()
            args = (h, text(), rnd.sample(tags_pool, rnd.randint(0, 2)))
            ms = rnd.randint(0, 50)
            g.add_claim(*args, created_ms=ms); og.add_claim(*args, created_ms=ms)
            known.append(h)
        elif r < 0.55:
            h = rnd.choice(known)
            args = (h, text(), [])
            ms = rnd.randint(0, 50)
            g.add_claim(*args, created_ms=ms); og.add_claim(*args, created_ms=ms)
        elif r < 0.68:
            h = rnd.choice(known)
            g.retract(h); og.retract(h)
        elif r < 0.70:
            g.retract("never-added"); og.retract("never-added")
        elif r < 0.73:
            h = rnd.choice(known)
            if h in og.claims:
                flag = not og.claims[h].retracted
                g.claims[h].retracted = flag; og.claims[h].retracted = flag
        elif r < 0.76:
            h = rnd.choice(known)
            if h in og.claims:
                ms = rnd.randint(0, 50)
                g.claims[h].created_ms = ms; og.claims[h].created_ms = ms
        elif r < 0.78:
            h = rnd.choice(known)
            if h in og.claims:
                del g.claims[h]; del og.claims[h]
                known.remove(h)
        elif r < 0.81:  # a claim put into the dict by hand: the reference embeds it when a compile first scores it (:113-117)
            from lbm_amd.context_graph import Claim
            h = new_hash() if rnd.random() < 0.7 else rnd.choice(known)
            t, tg, ms = text(), rnd.sample(tags_pool, rnd.randint(0, 2)), rnd.randint(0, 50)
            g.claims[h] = Claim(h, t, list(tg), ms, [])
            og.claims[h] = orc.OracleClaim(h, t, list(tg), ms, [], None)
            if h not in known:
                known.append(h)
        elif r < 0.83:  # text edits only matter for claims the reference has not cached yet
            h = rnd.choice(known)
            if h in og.claims:
                t = text()
                g.claims[h].text = t; og.claims[h].text = t
        else:
            q = " ".join(rnd.choice(vocab) for _ in range(rnd.randint(0, 3)))
            kw = {"top_k": rnd.choice([0, 1, 3, 8, 8, 8, 20, 100])}
            if rnd.random() < 0.4:
                kw["since_ms"] = rnd.randint(0, 55)
            if rnd.random() < 0.25:  # the opt-in tag filter (extension) rides on the same mirror
                kw["tags"] = rnd.sample(tags_pool + ["never"], rnd.randint(1, 2))
            assert g.compile(q, **kw) == og.compile(q, **kw), (step, q, kw)
            hs, sc = g.compile_scored(q, **kw)
            ohs, osc = og.compile_scored(q, **kw)
            assert hs == ohs and hexes(sc) == hexes(osc), (step, q, kw)
            if n_queries % 9 == 0:  # the batched entry must agree with single queries at every state of the graph
                qs = [q, rnd.choice(vocab), rnd.choice(vocab) + " " + rnd.choice(vocab), ""]
                assert g.compile_many(qs, **kw) == [og.compile(x, **kw) for x in qs], (step, qs, kw)
            n_queries += 1
        if step % 23 == 0:  # the cache view holds exactly the hashes the reference's cache holds, with the same vectors
            assert sorted(g._embedding_cache.keys()) == sorted(og._embedding_cache), step
            for h in list(og._embedding_cache)[:3]:
                assert hexes(g._embedding_cache[h]) == hexes(og._embedding_cache[h]), (step, h)
    hip.set_option("scan_mode", 0)
    assert n_queries > 60  # ~119 expected (17 % of 700 steps); a floor, not a distribution test


def test_queries_overlap_publishes_from_another_thread(hip):
    """compile_context takes no lock in the reference while writers publish (lb/node.py:718 vs :465): a reader
    thread must never crash or see a half-updated mirror, and once the writer is done the graph must answer
    exactly like the oracle fed with the same operations."""
    import threading
    from lbm_amd.context_graph import ContextGraph
    g, og = ContextGraph(), orc.OracleGraph()
    ops = []
    for i in range(1500):
        ops.append(("add", "h%05d" % i, "alpha beta gamma delta eps zeta".split()[i % 6] + " w%d x%d" % (i % 17, i % 5), i))
        if i % 7 == 3:
            ops.append(("retract", "h%05d" % (i - 2), None, None))
    errors, done = [], threading.Event()

    def writer():
        try:
            for op, h, text, ms in ops:
                if op == "add":
                    g.add_claim(h, text, ["t"], created_ms=ms)
                else:
                    g.retract(h)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
        finally:
            done.set()

    def reader():
        try:
            n = 0
            while not done.is_set() or n < 50:
                text, hashes = g.compile("alpha w3", top_k=5)
                assert len(hashes) <= 5 and text.startswith("Context slice")
                assert len(g.recent(0, 7)) <= 7
                n += 1
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=writer), threading.Thread(target=reader), threading.Thread(target=reader)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errors, errors
    for op, h, text, ms in ops:
        if op == "add":
            og.add_claim(h, text, ["t"], created_ms=ms)
        else:
            og.retract(h)
    for q, kw in (("alpha w3", {"top_k": 5}), ("zeta x1 w16", {"top_k": 20, "since_ms": 700}), ("", {"top_k": 3})):
        assert g.compile(q, **kw) == og.compile(q, **kw)
    assert g.recent(1000, 50) == [c["claim_hash"] for c in orc.recent_claims(og.claims, 1000, 50)]


def test_c_abi_calls_from_many_threads(hip):
    """The library's own locking (per-index mutex, per-context scratch mutex), without the Python graph lock in
    front of it: four threads search two indexes (single-query, batched, recency, text embedding) while a fifth
    appends to and retracts from a third index on the same context.  Every answer must equal the one computed
    before the threads started."""
    import threading
    rng = np.random.default_rng(5)

    def make(n):
        v = rng.standard_normal((n, 64))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        ix = hip.Index()
        ids = ix.add_vectors(v, rng.integers(0, 1000, size=n).astype(np.int64), None, rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
        return ix, ids

    (a, _), (b, _), (c, _) = make(30_000), make(4_000), make(100)
    qs = rng.standard_normal((8, 64))
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    texts = ["alpha beta %d" % i for i in range(50)]
    want = {}
    for name, ix in (("a", a), ("b", b)):
        want[name] = ([tuple(x.tobytes() for x in ix.search_shard(qs[j], 8, hip.SINCE_NONE, 0)) for j in range(8)],
                      tuple(x.tobytes() for x in ix.search(qs, 8)), tuple(np.asarray(x).tobytes() for x in ix.recent(500, 20)[:2]))
    want_emb = hip.embed_batch(texts).tobytes()
    errors, stop = [], threading.Event()

    def reader(seed):
        try:
            r = np.random.default_rng(seed)
            for it in range(150):
                name, ix = (("a", a), ("b", b))[int(r.integers(0, 2))]
                what = int(r.integers(0, 4))
                if what == 0:
                    j = int(r.integers(0, 8))
                    assert tuple(x.tobytes() for x in ix.search_shard(qs[j], 8, hip.SINCE_NONE, 0)) == want[name][0][j]
                elif what == 1:
                    assert tuple(x.tobytes() for x in ix.search(qs, 8)) == want[name][1]
                elif what == 2:
                    assert tuple(np.asarray(x).tobytes() for x in ix.recent(500, 20)[:2]) == want[name][2]
                else:
                    assert hip.embed_batch(texts).tobytes() == want_emb
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def writer():
        try:
            r = np.random.default_rng(99)
            while not stop.is_set():
                v = r.standard_normal((64, 64))
                ids = c.add_vectors(v, np.zeros(64, dtype=np.int64), None, r.integers(0, 256, size=(64, 32), dtype=np.uint8))
                for i in ids[::2].tolist():
                    c.retract(int(i))
                c.search_shard(qs[0], 4, hip.SINCE_NONE, 0)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=reader, args=(s,)) for s in range(4)]
    w = threading.Thread(target=writer)
    for t in ts + [w]:
        t.start()
    for t in ts:
        t.join(timeout=100)
    stop.set()
    w.join(timeout=30)
    assert not errors, errors
    for ix in (a, b, c):
        ix.close()


def _arrays_case(n, seed, dup_frac=0.002):
    from lbm_amd import synth
    ch = synth.make_chunk(0, n, seed=seed, real_sha_keys=False, dup_frac=dup_frac)
    return ch


@pytest.fixture(params=["scan_fp32", "scan_fp16"])
def scan_mode(request, hip):
    """Single-query scan over the fp32 matrix (default) or over the opt-in fp16 copy; results must not differ."""
    mode = 1 if request.param == "scan_fp16" else 0
    hip.set_option("scan_mode", mode)
    yield mode
    hip.set_option("scan_mode", 0)


def test_index_vs_oracle_200k(hip, scan_mode):
    """C-ABI level: 200k synthetic claims, fast path / paging / filters / tag mask, all bit-exact."""
    from lbm_amd import synth
    ch = _arrays_case(200_000, 77)
    ix = hip.Index()
    ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    assert ids.tolist() == list(range(ch.n)) and ix.size() == ch.n
    for i in np.nonzero(ch.retracted)[0].tolist():
        ix.retract(i)
    assert ix.size() == ch.n - int(ch.retracted.sum())
    emb = orc.embed_batch_packed(ch.blob, ch.offsets)
    for i in (0, 1, 12345, ch.n - 1):
        if not ch.retracted[i]:
            assert ix.get_embedding(i).tobytes() == emb[i].tobytes()
    keys_hex = ch.hashes()
    packed = orc.pack_texts(keys_hex)
    retr = ch.retracted.astype(np.uint8)
    queries = synth.make_queries(6, seed=11) + [ch.text(4242), ""]
    p50, p90 = int(ch.created_ms[ch.n // 2]), int(ch.created_ms[(9 * ch.n) // 10])
    st = hip.Stats()
    for qi, qtext in enumerate(queries):
        qv = orc.embed_array(qtext)
        assert hip.embed_batch([qtext])[0].tobytes() == qv.tobytes()
        for k, since, tmask in ((8, None, 0), (64, p50, 0), (8, p90, 0), (200, None, 0), (8, None, 0b1011), (64, p50, 1 << 63 | 5)):
            if qtext == "" and k == 200:
                continue
            want_idx, want_sc = orc.compile_arrays(emb, ch.created_ms, retr, None, qv, k, since, ch.tagmask, tmask, packed_keys=packed)
            gi, gs, cnt = ix.search(qv[None, :], k, hip.SINCE_NONE if since is None else since, tmask, st)
            n = int(cnt[0])
            assert gi[0, :n].tolist() == want_idx.tolist(), (qi, k, since, tmask)
            assert gs[0, :n].tobytes() == want_sc.tobytes(), (qi, k, since, tmask)
            if qtext == "":
                assert st.fallbacks >= 1 and st.path == 1  # all scores 0.0: pure hash order via the exact path
            elif k <= 64:
                assert st.path == 0 and st.fallbacks == 0 and st.candidates >= n
                assert st.bytes_per_row == (128 if scan_mode else 256) + (8 if since is not None else 0) + (8 if tmask else 0)
                assert st.max_score_err < (1.1e-3 if scan_mode else 5e-6)
    # text entry point == vector entry point
    gi, gs, _ = ix.search_text(queries[0], 8)
    gi2, gs2, _ = ix.search(orc.embed_array(queries[0])[None, :], 8)
    assert gi.tolist() == gi2[0].tolist() and gs.tobytes() == gs2[0].tobytes()
    ix.close()


def test_mass_duplicates_and_ties(hip):
    """5000 copies of one text + noise: exact ties far beyond any candidate buffer, resolved by hash."""
    from lbm_amd.context_graph import ContextGraph
    rng = np.random.default_rng(3)
    g, og = ContextGraph(), orc.OracleGraph()
    import hashlib
    for i in range(6000):
        h = hashlib.sha256(b"k%d" % i).hexdigest()
        text = "identical duplicate text body" if i % 6 else "noise %d words here" % int(rng.integers(0, 50))
        for gg in (g, og):
            gg.add_claim(h, text, ["d"] if i % 2 else [], created_ms=1000 + i)
    for q, kw in (("duplicate text", {"top_k": 8}), ("duplicate text", {"top_k": 300}), ("", {"top_k": 64}),
                  ("noise", {"top_k": 64, "since_ms": 4000}), ("identical duplicate text body", {"top_k": 64, "since_ms": 6990})):
        assert g.compile(q, **kw) == og.compile(q, **kw), (q, kw)
        assert g.compile_scored(q, **kw)[1] == og.compile_scored(q, **kw)[1]


def test_tags_extension_matches_oracle(hip):
    from lbm_amd.context_graph import ContextGraph
    from lbm_amd import synth
    ch = synth.make_chunk(0, 4000, seed=5)
    g, og = ContextGraph(), orc.OracleGraph()
    texts, hashes = ch.texts(), ch.hashes()
    for i in range(ch.n):
        for gg in (g, og):
            gg.add_claim(hashes[i], texts[i], ch.tags(i), created_ms=int(ch.created_ms[i]))
    for q in synth.make_queries(3, seed=2):
        for tags in (["t03"], ["t10", "t11", "t63"], ["unknown"], []):
            assert g.compile(q, top_k=16, tags=tags) == og.compile(q, top_k=16, tags=tags), (q, tags)
        assert g.compile(q, top_k=16, tags=["t00"], since_ms=int(ch.created_ms[2000])) == \
            og.compile(q, top_k=16, tags=["t00"], since_ms=int(ch.created_ms[2000]))


# K3 scan arithmetic: (option value, candidate margin).  Whatever the scan's precision, results are exact:
# the margin is twice the scan's error bound and every candidate is re-scored in fp64.
BQ_MODES = {"fp32_mfma": (0, 1.0e-5), "bf16_split3": (1, 1.0e-4), "fp16": (2, 2.2e-3), "int8": (3, None)}  # int8: margin per batch (stats.margin)


@pytest.fixture(params=list(BQ_MODES))
def bq_mode(request, hip):
    mode, margin = BQ_MODES[request.param]
    hip.set_option("bq_mode", mode)
    yield margin
    hip.set_option("bq_mode", 2)


def test_batched_queries_mfma_path_vs_oracle(hip, bq_mode):
    """K3: many queries in one pass on the matrix cores; every query must equal the oracle bit-for-bit,
    and the scan's measured error must stay inside the bound its candidate margin assumes."""
    from lbm_amd import synth
    ch = _arrays_case(150_000, 123)
    ix = hip.Index()
    ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    for i in np.nonzero(ch.retracted)[0].tolist():
        ix.retract(i)
    emb = orc.embed_batch_packed(ch.blob, ch.offsets)
    packed = orc.pack_texts(ch.hashes())
    retr = ch.retracted.astype(np.uint8)
    texts = synth.make_queries(37, seed=21) + [ch.text(777), ch.text(778) + " " + ch.text(779), ""]
    qv = hip.embed_batch(texts)
    p50 = int(ch.created_ms[ch.n // 2])
    st = hip.Stats()
    tiled = bq_mode in (None, BQ_MODES["fp16"][1])
    p99 = int(ch.created_ms[ch.n * 999 // 1000])
    # k > 16: the fp16 / int8 kernel only; k > 64: served in result pages of 64 by the same pass (the last case leaves
    # fewer live rows than k: pages past the end are empty)
    big_k = ((40, None, 0), (64, p50, 0b1), (100, None, 0), (200, p50, 0), (256, None, 0b110), (130, p99, 0)) if tiled else ()
    for k, since, tmask in ((8, None, 0), (16, p50, 0), (1, None, 0b110), (8, p50, 1 << 7 | 1)) + big_k:
        gi, gs, cnt = ix.search(qv, k, hip.SINCE_NONE if since is None else since, tmask, st)
        assert st.path == 2 and st.scan_launches >= 1
        margin = bq_mode if bq_mode is not None else st.margin
        if bq_mode is None:
            assert st.bytes_per_row in (64, 72, 80) and 0.0 < st.margin < 0.09, (st.bytes_per_row, st.margin)  # the int8 kernel served it
        assert 0.0 <= st.max_score_err < margin / 2, (st.max_score_err, margin)
        for j in range(len(texts)):
            want_idx, want_sc = orc.compile_arrays(emb, ch.created_ms, retr, None, qv[j], k, since, ch.tagmask, tmask,
                                                   packed_keys=packed)
            n = int(cnt[j])
            assert gi[j, :n].tolist() == want_idx.tolist(), (k, since, tmask, j)
            assert gs[j, :n].tobytes() == want_sc.tobytes(), (k, since, tmask, j)
        if (k, since, tmask) not in ((64, p50, 0b1), (130, p99, 0)):  # (those filters leave few enough rows for the tied list to fit)
            assert st.fallbacks >= 1  # the empty query ties every claim at 0.0 -> served by the exact path
    # 300 queries: two launches of the 256-query kernel
    many = synth.make_queries(300, seed=5)
    qm = hip.embed_batch(many)
    gi, gs, cnt = ix.search(qm, 8, hip.SINCE_NONE, 0, st)
    assert st.path == 2 and st.scan_launches == 2 and st.fallbacks == 0
    assert 0.0 < st.max_score_err < (bq_mode if bq_mode is not None else st.margin) / 2
    for j in (0, 1, 128, 255, 256, 257, 299):
        want_idx, want_sc = orc.compile_arrays(emb, ch.created_ms, retr, None, qm[j], 8, None, packed_keys=packed)
        assert gi[j, : int(cnt[j])].tolist() == want_idx.tolist() and gs[j, : int(cnt[j])].tobytes() == want_sc.tobytes()
    ix.close()


def test_compile_many_equals_compile(hip):
    from lbm_amd.context_graph import ContextGraph
    from lbm_amd import synth
    ch = synth.make_chunk(0, 5000, seed=8, dup_frac=0.05)
    g = ContextGraph()
    texts, hashes = ch.texts(), ch.hashes()
    for i in range(ch.n):
        g.add_claim(hashes[i], texts[i], ch.tags(i), created_ms=int(ch.created_ms[i]))
    qs = synth.make_queries(9, seed=1) + [texts[3], "", "zzz unknown"]
    since = int(ch.created_ms[2500])
    for kw in ({"top_k": 8}, {"top_k": 16, "since_ms": since}, {"top_k": 5, "tags": ["t01", "t02"]}, {"top_k": 0},
               {"top_k": 40}):
        many = g.compile_many(qs, **kw)
        assert many == [g.compile(q, **kw) for q in qs], kw
    assert g.compile_many([], top_k=3) == []


def test_sharded_searcher_over_rccl_single_rank(hip):
    """The N > 1 code path of bench.py on the one GPU we have: RCCL process group of size 1, device tensors
    through all_gather_into_tensor, merge -- must equal the plain search."""
    import os
    import torch
    import torch.distributed as dist
    from lbm_amd import synth
    from lbm_amd.sharded import ShardedSearcher
    ch = _arrays_case(50_000, 9)
    ix = hip.Index()
    ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        s = ShardedSearcher(ix.search_shard, 0, 1, device=torch.device("cuda", 0), always_gather=True)
        qv = hip.embed_batch(synth.make_queries(4, seed=3))
        for q in qv:
            for k in (8, 64):
                gi, gs, _ = s.search(q, k, hip.SINCE_NONE, 0)
                wi, ws, _ = ix.search_shard(q, k, hip.SINCE_NONE, 0)
                assert gi.tolist() == wi.tolist() and gs.tobytes() == ws.tobytes()
    finally:
        dist.destroy_process_group()
        ix.close()
