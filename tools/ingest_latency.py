#!/usr/bin/env python3
"""cost of the incremental ingest path on a populated graph, through the drop-in API.

Builds a ContextGraph of --claims claims, then measures (wall clock, p50 / p99 over --reps):
  compile                      query on an unchanged graph (mirror already in sync)
  add_claim + compile          one publish (K1 embeds one text, one row appended) then a query
  retract + compile            one tombstone (swap-delete of one device row) then a query
  overwrite + compile          add_claim on an existing hash (re-embed, row rewritten in place)
  add_claim x100 + compile     a burst of 100 publishes folded into one K1 launch at the next query
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth  # noqa: E402
from lbm_amd.context_graph import ContextGraph  # noqa: E402


def pct(v):
    v = np.array(v) * 1e3
    return {"p50_ms": float(np.percentile(v, 50)), "p99_ms": float(np.percentile(v, 99))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--claims", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    g = ContextGraph()
    t0 = time.perf_counter()
    done = 0
    hashes_all = []
    while done < args.claims:
        m = min(250_000, args.claims - done)
        ch = synth.make_chunk(done, m, seed=1234, chunk_index=done // 250_000)
        texts, hashes = ch.texts(), ch.hashes()
        for i in range(m):
            g.add_claim(hashes[i], texts[i], ch.tags(i), created_ms=int(ch.created_ms[i]))
        hashes_all.extend(hashes)
        done += m
    t_build = time.perf_counter() - t0
    queries = synth.make_queries(64, seed=4321)
    t0 = time.perf_counter()
    g.compile(queries[0])
    t_first = time.perf_counter() - t0
    extra = synth.make_chunk(10 ** 9, args.reps * 101 + 8, seed=99)
    etexts, ehashes = extra.texts(), extra.hashes()
    out = {"device": _hip.device_name(), "claims": args.claims, "host_add_claim_loop_s": t_build,
           "first_compile_incl_bulk_vectorise_s": t_first}
    lat = {k: [] for k in ("compile", "add+compile", "retract+compile", "overwrite+compile", "add100+compile")}
    e = 0
    for r in range(args.reps):
        q = queries[r % 64]
        t0 = time.perf_counter(); g.compile(q); lat["compile"].append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        g.add_claim(ehashes[e], etexts[e], ["x"], created_ms=5); e += 1
        g.compile(q); lat["add+compile"].append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        g.retract(hashes_all[(r * 7919) % args.claims])
        g.compile(q); lat["retract+compile"].append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        g.add_claim(hashes_all[(r * 104729 + 1) % args.claims], etexts[e], [], created_ms=6)
        g.compile(q); lat["overwrite+compile"].append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(100):
            g.add_claim(ehashes[e], etexts[e], [], created_ms=7); e += 1
        g.compile(q); lat["add100+compile"].append(time.perf_counter() - t0)
    for k, v in lat.items():
        out[k] = pct(v)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
