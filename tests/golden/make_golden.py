#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REFERENCE (lb.latent, lb.context_graph).

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference python3 tests/golden/make_golden.py

Writes small JSON fixtures next to this script.  Fixtures are data only: inputs
and the outputs the reference produced for them (floats as float.hex()), plus
the interpreter version that produced them (CPython < 3.12 sums floats
left-to-right; 3.12+ uses compensated summation, which would change low bits).
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from lb.latent import embed, cosine, _tokenize  # noqa: E402  (the reference)
from lb.context_graph import ContextGraph  # noqa: E402  (the reference)

from lbm_amd import synth  # noqa: E402  (our input generator; produces inputs only)

META = {"python": sys.version.split()[0], "reference": "amazedsaint/lbm v0.6.2 (lb/latent.py, lb/context_graph.py)"}


def hexv(v):
    return [float(x).hex() for x in v]


def embed_cases():
    texts = [
        "hello", "hello world", "Hello, hello! HELLO", "", "   ", "!!! ??? ...", "Héllo wörld_1 FOO-bar 42",
        "Hello_World 42!x", "a", "A", "_", "__init__ __main__", "0", "42 42 42 42", "the quick brown fox jumps over the lazy dog",
        "The Quick Brown Fox Jumps Over The Lazy Dog", "dog lazy the over jumps fox brown quick the",
        "database optimization", "how to debug build", "data", "query", "test", "Old data", "New data",
        "日本語 テキスト mixed with ascii", "emoji 😀 split😀token", "tab\tseparated\nnewline\r\nwords", "nul\x00byte inside",
        "x" * 54, "x" * 55, "x" * 56, "x" * 57, "x" * 63, "x" * 64, "x" * 65, "y" * 118, "y" * 119, "y" * 120, "y" * 121,
        "Z" * 200, "q" * 1000, "ab " * 300, " ".join("w%05d" % i for i in range(100)),
        "MiXeD CaSe ToKeNs AND_under_SCORES 123abc ABC123", "trailing space ", " leading space", "a-b-c-d-e-f-g",
        "é", "ééé", "naïve café résumé", "ßıK K k",  # non-ASCII letters never lower() into ASCII tokens
        "w00001 w00002 w00003 w00001", "x" * 5000 + " tail",
    ]
    out = []
    for t in texts:
        out.append({"text": t, "tokens": _tokenize(t), "embedding": hexv(embed(t))})
    return out


def cosine_cases():
    pairs = [("hello", "hello world"), ("hello", "hello"), ("", "hello"), ("a b c", "c b a"),
             ("database optimization", "optimization of the database index"), ("x" * 64, "y" * 64),
             ("the quick brown fox", "lazy dog"), ("Hello_World 42!x", "hello_world 42 x")]
    out = []
    for a, b in pairs:
        out.append({"a": a, "b": b, "cosine": float(cosine(embed(a), embed(b))).hex()})
    return out


def run_graph(ops, queries):
    """ops: list of ["add", hash, text, tags, created_ms] | ["retract", hash]; queries: list of dicts."""
    g = ContextGraph()
    for op in ops:
        if op[0] == "add":
            g.add_claim(op[1], op[2], op[3], created_ms=op[4])
        elif op[0] == "retract":
            g.retract(op[1])
        elif op[0] == "raw_insert":  # tests/test_multi_agent.py style: straight into the dict, no cache entry
            from lb.context_graph import Claim
            g.claims[op[1]] = Claim(claim_hash=op[1], text=op[2], tags=list(op[3]), created_ms=op[4], evidence=[])
    res = []
    for q in queries:
        kw = {}
        if "top_k" in q:
            kw["top_k"] = q["top_k"]
        if "since_ms" in q:
            kw["since_ms"] = q["since_ms"]
        text, hashes = g.compile(q["query"], **kw)
        qv = embed(q["query"])
        scores = [float(cosine(qv, g._embedding_cache[h])).hex() for h in hashes]
        res.append({**q, "context": text, "hashes": hashes, "scores": scores})
    return res


def compile_cases():
    cases = []
    # 1. reference tests' own scenarios (tests/test_multi_agent.py:199-219, tests/test_edge_cases.py:608-623)
    ops = [["add", "old", "Old data", ["data"], 1000], ["add", "new", "New data", ["data"], 2000]]
    qs = [{"query": "data", "top_k": 10, "since_ms": 1500}, {"query": "data", "top_k": 10},
          {"query": "data", "top_k": 10, "since_ms": 2000}, {"query": "data", "top_k": 10, "since_ms": 2001},
          {"query": "data", "top_k": 10, "since_ms": 0}, {"query": "data", "top_k": 1}, {"query": "data", "top_k": 0},
          {"query": "data", "top_k": -3}, {"query": "", "top_k": 10}, {"query": "!!!"}, {"query": "data"}]
    cases.append({"name": "time_filter", "ops": ops, "results": run_graph(ops, qs)})
    ops = []
    cases.append({"name": "empty_graph", "ops": ops, "results": run_graph(ops, [{"query": "query"}, {"query": ""}, {"query": "trailing  "}])})
    ops = [["add", "h1", "Test claim", ["test"], 1000], ["retract", "h1"]]
    cases.append({"name": "all_retracted", "ops": ops, "results": run_graph(ops, [{"query": "test"}])})
    ops = [["add", "old_claim", "Old claim", ["old"], 5000], ["add", "new_claim", "New claim about query", ["new"], 15000]]
    cases.append({"name": "node_since", "ops": ops, "results": run_graph(ops, [{"query": "query", "top_k": 10, "since_ms": 5001}])})
    # 2. ties: duplicate texts under different hashes, mixed-case hex-looking and non-hex keys
    dup = "the same exact text appears many times"
    keys = ["b" * 64, "a" * 64, "0" * 63 + "1", "f" * 64, "zzz", "Zzz", "a", "aa", "a" * 65, "é", "0", "00", "claim_0002", "claim_0001",
            "ABCDEF" + "0" * 58, "abcdef" + "0" * 58]
    ops = [["add", k, dup, ["dup"] if i % 2 else [], 1000 + i] for i, k in enumerate(keys)]
    ops += [["add", "other1", "something else entirely", ["x", "y"], 50], ["add", "other2", "the same text", [], 60]]
    qs = [{"query": "same text", "top_k": 5}, {"query": "same text", "top_k": 100}, {"query": "", "top_k": 100},
          {"query": "unrelated words only", "top_k": 7}, {"query": "same text", "top_k": 6, "since_ms": 1004},
          {"query": dup, "top_k": 3}]
    cases.append({"name": "ties_mixed_keys", "ops": ops, "results": run_graph(ops, qs)})
    # 3. overwrite same hash (resets retracted, re-embeds), retract then re-add, retract unknown, raw dict insert
    ops = [["add", "k1", "alpha beta gamma", ["a"], 10], ["add", "k2", "beta gamma delta", ["b"], 20],
           ["add", "k1", "delta epsilon", ["c", "d"], 30], ["retract", "k2"], ["retract", "nope"],
           ["add", "k3", "gamma", [], 40], ["add", "k2", "beta again", ["b2"], 50], ["retract", "k3"],
           ["raw_insert", "k4", "inserted without add_claim beta", ["raw"], 60]]
    qs = [{"query": "beta", "top_k": 10}, {"query": "delta epsilon", "top_k": 2}, {"query": "gamma", "top_k": 10, "since_ms": 35},
          {"query": "beta", "top_k": 10}]
    cases.append({"name": "overwrite_retract_raw", "ops": ops, "results": run_graph(ops, qs)})
    # 4. formatting: tags, whitespace-only query, text with newlines
    ops = [["add", "f1", "line one\nline two  ", ["t1", "t2", "t3"], 1], ["add", "f2", "  padded  ", [], 2], ["add", "f3", "", ["empty"], 3]]
    qs = [{"query": "line", "top_k": 3}, {"query": "  ", "top_k": 3}, {"query": "padded\n", "top_k": 1}]
    cases.append({"name": "formatting", "ops": ops, "results": run_graph(ops, qs)})
    return cases


def synth_case(n=2000, nq=12):
    ch = synth.make_chunk(0, n, seed=1234, chunk_index=0, real_sha_keys=True, dup_frac=0.01)
    texts, hashes = ch.texts(), ch.hashes()
    g = ContextGraph()
    for i in range(n):
        g.add_claim(hashes[i], texts[i], ch.tags(i), created_ms=int(ch.created_ms[i]))
    for i in range(n):
        if ch.retracted[i]:
            g.retract(hashes[i])
    queries = synth.make_queries(nq, seed=4321) + [texts[7], texts[100] + " " + texts[101], "", "w00000"]
    p50 = int(ch.created_ms[n // 2])
    p90 = int(ch.created_ms[(9 * n) // 10])
    results = []
    for q in queries:
        qv = embed(q)
        for top_k in (8, 64, 200):
            for since in (None, p50, p90):
                kw = {"top_k": top_k}
                if since is not None:
                    kw["since_ms"] = since
                text, hs = g.compile(q, **kw)
                scores = [float(cosine(qv, g._embedding_cache[h])).hex() for h in hs]
                results.append({"query": q, "top_k": top_k, "since_ms": since, "hashes": hs, "scores": scores,
                                "context_sha256": __import__("hashlib").sha256(text.encode()).hexdigest()})
    sample = [0, 1, 2, 7, 100, 101, n - 1]
    return {"n": n, "seed": 1234, "dup_frac": 0.01, "texts": texts, "hashes": hashes,
            "created_ms": [int(x) for x in ch.created_ms], "tagmask": [int(x) for x in ch.tagmask],
            "retracted": [bool(x) for x in ch.retracted],
            "embedding_samples": {str(i): hexv(embed(texts[i])) for i in sample},
            "results": results}


def main():
    def dump(name, obj):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump({"meta": META, "data": obj}, f, ensure_ascii=True, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")

    lut = [((b / 127.5) - 1.0).hex() for b in range(256)]  # lb/latent.py:52
    dump("embed.json", {"lut": lut, "embed": embed_cases(), "cosine": cosine_cases()})
    dump("compile.json", compile_cases())
    dump("synth2k.json", synth_case())


if __name__ == "__main__":
    main()
