#!/usr/bin/env python3
"""Round-2 golden vectors, again produced by IMPORTING THE REFERENCE (lb.latent, lb.context_graph):

    PYTHONPATH=/root/reference python3 tests/golden/make_golden_r2.py

  snapshot.json   ContextGraph.snapshot() / from_snapshot() (lb/context_graph.py:135-148): graphs with retracted,
                  overwritten, re-added and raw-inserted claims -> the snapshot dict the reference writes, and the
                  compile() results of the reference graph rebuilt from that snapshot.
  embed_dim.json  embed(text, dim) for dim < 64 (lb/latent.py:41-57: first `dim` digest bytes, norm over `dim`
                  components) and the reference's behaviour for dim = 0 and dim > 64.
Fixtures are data only (inputs + outputs, floats as float.hex(), interpreter version recorded)."""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from lb.latent import embed, cosine  # noqa: E402  (the reference)
from lb.context_graph import Claim, ContextGraph  # noqa: E402  (the reference)

from lbm_amd import synth  # noqa: E402  (our input generator; produces inputs only)

META = {"python": sys.version.split()[0], "reference": "amazedsaint/lbm v0.6.2 (lb/latent.py, lb/context_graph.py)"}


def hexv(v):
    return [float(x).hex() for x in v]


def build(ops):
    g = ContextGraph()
    for op in ops:
        if op[0] == "add":
            g.add_claim(op[1], op[2], op[3], created_ms=op[4], parent_hash=op[5] if len(op) > 5 else None)
        elif op[0] == "retract":
            g.retract(op[1])
        elif op[0] == "raw_insert":
            g.claims[op[1]] = Claim(claim_hash=op[1], text=op[2], tags=list(op[3]), created_ms=op[4], evidence=[])
        elif op[0] == "flag":  # flip Claim.retracted by hand (no cache change in the reference)
            g.claims[op[1]].retracted = op[2]
    return g


def results(g, queries):
    out = []
    for q in queries:
        kw = {k: q[k] for k in ("top_k", "since_ms") if k in q}
        text, hashes = g.compile(q["query"], **kw)
        qv = embed(q["query"])
        out.append({**q, "context": text, "hashes": hashes,
                    "scores": [float(cosine(qv, g._embedding_cache[h])).hex() for h in hashes]})
    return out


def snapshot_cases():
    cases = []
    ops = [["add", "k1", "alpha beta gamma", ["a"], 10], ["add", "k2", "beta gamma delta", ["b"], 20],
           ["add", "k1", "delta epsilon", ["c", "d"], 30], ["retract", "k2"], ["retract", "nope"],
           ["add", "k3", "gamma", [], 40, "k1"], ["add", "k2", "beta again", ["b2"], 50], ["retract", "k3"],
           ["raw_insert", "k4", "inserted without add_claim beta", ["raw"], 60], ["add", "k5", "beta beta", [], 70],
           ["flag", "k5", True], ["add", "k6", "epsilon beta", ["e"], 80, "k4"]]
    qs = [{"query": "beta", "top_k": 10}, {"query": "delta epsilon", "top_k": 2}, {"query": "gamma", "top_k": 10, "since_ms": 35},
          {"query": "", "top_k": 10}, {"query": "beta", "top_k": 10, "since_ms": 60}]
    g = build(ops)
    snap = g.snapshot()
    g2 = ContextGraph.from_snapshot(json.loads(json.dumps(snap)))  # through JSON, like Group.save / Group.load
    cases.append({"name": "small_mixed", "ops": ops, "snapshot": snap, "cached_after_load": sorted(g2._embedding_cache),
                  "results": results(g2, qs)})
    # the 2000-claim synthetic corpus of synth2k.json, with overwrites and retracts, through snapshot -> from_snapshot
    n = 2000
    ch = synth.make_chunk(0, n, seed=1234, chunk_index=0, real_sha_keys=True, dup_frac=0.01)
    texts, hashes = ch.texts(), ch.hashes()
    ops = [["add", hashes[i], texts[i], ch.tags(i), int(ch.created_ms[i])] for i in range(n)]
    ops += [["retract", hashes[i]] for i in range(n) if ch.retracted[i]]
    ops += [["add", hashes[i], texts[(i * 7 + 1) % n], ["over"], int(ch.created_ms[i]) + 5] for i in range(3, n, 97)]  # overwrite
    ops += [["retract", hashes[i]] for i in range(5, n, 211)]
    g = build(ops)
    snap = g.snapshot()
    g2 = ContextGraph.from_snapshot(json.loads(json.dumps(snap)))
    p50 = int(ch.created_ms[n // 2])
    qs = []
    for q in synth.make_queries(6, seed=4321) + [texts[7], ""]:
        qs += [{"query": q, "top_k": 8}, {"query": q, "top_k": 64, "since_ms": p50}, {"query": q, "top_k": 200}]
    res = results(g2, qs)
    for r in res:  # keep the fixture small: the context text as a digest
        r["context_sha256"] = __import__("hashlib").sha256(r.pop("context").encode()).hexdigest()
    cases.append({"name": "synth2k_overwrites", "n": n, "seed": 1234, "dup_frac": 0.01,
                  "overwrite": list(range(3, n, 97)), "retract_extra": list(range(5, n, 211)),
                  "snapshot_sha256": __import__("hashlib").sha256(json.dumps(snap, sort_keys=True).encode()).hexdigest(),
                  "n_cached_after_load": len(g2._embedding_cache), "results": res})
    return cases


def embed_dim_cases():
    texts = ["hello", "hello world", "", "Hello, hello! HELLO", "the quick brown fox jumps over the lazy dog",
             "x" * 55 + " " + "y" * 56, " ".join("w%05d" % i for i in range(40)), "é日😀", "a"]
    out = []
    for t in texts:
        for dim in (1, 2, 16, 31, 32, 33, 63, 64):
            out.append({"text": t, "dim": dim, "embedding": hexv(embed(t, dim))})
    special = []
    for t, dim in (("hello", 0), ("", 0), ("", 65), ("", 100), ("!!!", 70)):
        special.append({"text": t, "dim": dim, "embedding": hexv(embed(t, dim))})
    errors = []
    for t, dim in (("hello", 65), ("a b", 1000)):
        try:
            embed(t, dim)
            errors.append({"text": t, "dim": dim, "raises": None})
        except Exception as e:  # noqa: BLE001
            errors.append({"text": t, "dim": dim, "raises": type(e).__name__})
    return {"cases": out, "special": special, "errors": errors}


def main():
    def dump(name, obj):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump({"meta": META, "data": obj}, f, ensure_ascii=True, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")

    dump("snapshot.json", snapshot_cases())
    dump("embed_dim.json", embed_dim_cases())


if __name__ == "__main__":
    main()
