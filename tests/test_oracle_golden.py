"""Pins the CPU oracle (oracle/lb_oracle.c) bit-for-bit to vectors produced by the reference itself.

Golden files come from tests/golden/make_golden.py (imports lb.latent / lb.context_graph from the
reference under CPython 3.10.12).  No GPU needed.
"""
import hashlib

import numpy as np
import pytest

from oracle import lb_oracle as orc


def test_lut_matches_reference_expression(golden_embed):
    # lb/latent.py:52  (b / 127.5) - 1.0
    for b, hx in enumerate(golden_embed["lut"]):
        assert orc.lib().lbo_lut(b).hex() == hx
    assert orc.lib().lbo_lut(0).hex() == "-0x1.0000000000000p+0"
    assert orc.lib().lbo_lut(1).hex() == "-0x1.fbfbfbfbfbfc0p-1"
    assert orc.lib().lbo_lut(127).hex() == "-0x1.0101010101000p-8"
    assert orc.lib().lbo_lut(128).hex() == "0x1.0101010101000p-8"
    assert orc.lib().lbo_lut(255).hex() == "0x1.0000000000000p+0"


@pytest.mark.parametrize("n", [0, 1, 3, 54, 55, 56, 57, 63, 64, 65, 119, 120, 121, 1000])
def test_sha256_against_hashlib(n):
    data = bytes((i * 7 + 3) & 0xFF for i in range(n))
    assert orc.sha256(data) == hashlib.sha256(data).digest()


def test_sha256_kat():
    assert orc.sha256(b"hello").hex().startswith("2cf24dba5fb0a30e")
    assert orc.sha256(b"hello\x00").hex().startswith("f3aefe62965a9190")


def test_tokenize_golden(golden_embed):
    for case in golden_embed["embed"]:
        assert orc.tokenize(case["text"]) == case["tokens"], case["text"][:40]


def test_embed_golden_bit_exact(golden_embed):
    for case in golden_embed["embed"]:
        got = [float(x).hex() for x in orc.embed(case["text"])]
        assert got == case["embedding"], case["text"][:40]


def test_embed_kat_from_survey():
    e = orc.embed("hello")
    assert e[0].hex() == "-0x1.2e0984b25fcebp-3"
    assert e[32].hex() == "0x1.a1c9b78cf5f60p-3"
    assert e[63].hex() == "-0x1.e852d68a29761p-4"
    assert orc.embed("Hello, hello! HELLO") != e


def test_embed_batch_equals_single(golden_embed):
    texts = [c["text"] for c in golden_embed["embed"]]
    got = orc.embed_batch(texts)
    for i, c in enumerate(golden_embed["embed"]):
        assert [float(x).hex() for x in got[i]] == c["embedding"]


def test_cosine_golden(golden_embed):
    for c in golden_embed["cosine"]:
        assert orc.cosine(orc.embed(c["a"]), orc.embed(c["b"])).hex() == c["cosine"]
    assert orc.cosine(orc.embed("hello"), orc.embed("hello world")).hex() == "0x1.8e586ca656a84p-1"
    with pytest.raises(ValueError, match="dimension mismatch"):
        orc.cosine([1.0, 2.0], [1.0])


def _replay(ops):
    g = orc.OracleGraph()
    for op in ops:
        if op[0] == "add":
            g.add_claim(op[1], op[2], op[3], created_ms=op[4])
        elif op[0] == "retract":
            g.retract(op[1])
        elif op[0] == "raw_insert":
            g.claims[op[1]] = orc.OracleClaim(op[1], op[2], op[3], op[4], [])
    return g


def test_compile_golden(golden_compile):
    for case in golden_compile:
        g = _replay(case["ops"])
        for r in case["results"]:
            kw = {k: r[k] for k in ("top_k", "since_ms") if k in r}
            text, hashes = g.compile(r["query"], **kw)
            assert hashes == r["hashes"], (case["name"], r["query"], kw)
            assert text == r["context"], (case["name"], r["query"], kw)
            hs, sc = g.compile_scored(r["query"], **kw)
            assert [s.hex() for s in sc] == r["scores"], (case["name"], r["query"], kw)


def test_compile_synth2k_golden(golden_synth):
    d = golden_synth
    g = orc.OracleGraph()
    emb = orc.embed_batch(d["texts"])
    for i_str, hx in d["embedding_samples"].items():
        assert [float(x).hex() for x in emb[int(i_str)]] == hx
    keys = d["hashes"]
    created = np.array(d["created_ms"], dtype=np.int64)
    retr = np.array(d["retracted"], dtype=np.uint8)
    for r in d["results"]:
        qv = orc.embed_array(r["query"])
        idx, sc = orc.compile_arrays(emb, created, retr, keys, qv, r["top_k"], r["since_ms"])
        assert [keys[i] for i in idx] == r["hashes"], (r["query"], r["top_k"], r["since_ms"])
        assert [float(s).hex() for s in sc] == r["scores"]


def test_embed_dim_golden(golden_embed_dim):
    """embed(text, dim) for dim != 64 (lb/latent.py:41-57): truncated accumulation and norm, [] for 0, IndexError past 64."""
    for c in golden_embed_dim["cases"] + golden_embed_dim["special"]:
        assert [float(x).hex() for x in orc.embed(c["text"], c["dim"])] == c["embedding"], (c["text"][:20], c["dim"])
    for c in golden_embed_dim["errors"]:
        assert c["raises"] == "IndexError"
        with pytest.raises(IndexError):
            orc.embed(c["text"], c["dim"])


def _replay_r2(ops):
    g = orc.OracleGraph()
    for op in ops:
        if op[0] == "add":
            g.add_claim(op[1], op[2], op[3], created_ms=op[4], parent_hash=op[5] if len(op) > 5 else None)
        elif op[0] == "retract":
            g.retract(op[1])
        elif op[0] == "raw_insert":
            g.claims[op[1]] = orc.OracleClaim(op[1], op[2], op[3], op[4], [])
        elif op[0] == "flag":
            g.claims[op[1]].retracted = op[2]
    return g


def test_snapshot_reload_golden(golden_snapshot):
    """lb/context_graph.py:135-148: the oracle graph rebuilt from the reference's snapshot dict answers like the
    reference graph rebuilt from it (retracted claims are not embedded, everything else is re-embedded from text)."""
    case = golden_snapshot[0]
    g = orc.OracleGraph.from_snapshot(case["snapshot"])
    assert sorted(g._embedding_cache) == case["cached_after_load"]
    for r in case["results"]:
        kw = {k: r[k] for k in ("top_k", "since_ms") if k in r}
        assert g.compile(r["query"], **kw) == (r["context"], r["hashes"])
        assert [s.hex() for s in g.compile_scored(r["query"], **kw)[1]] == r["scores"]
    assert _replay_r2(case["ops"]).snapshot() == case["snapshot"]
