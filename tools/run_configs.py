#!/usr/bin/env python3
"""Measure every BASELINE.json config that fits one MI355X and print one JSON document.

  config 1  ~1k claims through the Python drop-in API (plumbing / latency floor)
  config 2  1 M claims, single query (K2 scan + K4 re-rank)
  config 3  10 M claims, 256 batched queries (K3 MFMA path)
  config 4  one 12.5 M-claim shard of the 100 M / 8-GPU corpus, single query (per-GPU leg; the
            all-gather merge is covered by bench.py --gpus N on a multi-GPU node)
  config 5  bulk vectorisation (texts -> HBM index, K1) + since_ms / tag filtered queries (fused masks)

Usage: python3 tools/run_configs.py [--big 10000000] > profiles/rNN_configs.json
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


def build_index(n, seed=1234):
    ix = _hip.Index()
    ix.reserve(n)
    t_add = 0.0
    done = ci = 0
    while done < n:
        m = min(1_000_000, n - done)
        ch = synth.make_chunk(done, m, seed=seed, chunk_index=ci, real_sha_keys=False)
        t0 = time.perf_counter()
        ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
        t_add += time.perf_counter() - t0
        done += m
        ci += 1
    return ix, t_add


def ingest_prepacked(n_chunks=4, chunk=1_000_000):
    """lbh_index_add on PRE-PACKED buffers in pinned host memory (lbh_host_alloc): what the ingest call itself costs --
    PCIe + K1 + the host-side id maps -- with text generation and packing outside the timed region."""
    ix = _hip.Index()
    ix.reserve(n_chunks * chunk)
    t_pin = t_page = 0.0
    pin_calls, page_calls = [], []
    for ci in range(n_chunks):
        ch = synth.make_chunk(ci * chunk, chunk, seed=1234, chunk_index=ci, real_sha_keys=False)
        pins = [_hip.pinned_like(a) for a in (ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)]
        t0 = time.perf_counter()
        ids = ix.add_texts(*(p.a for p in pins))
        pin_calls.append(time.perf_counter() - t0)
        t_pin += pin_calls[-1]
        ix.retract_many(ids)  # same rows again from ordinary (pageable) numpy arrays
        t0 = time.perf_counter()
        ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
        page_calls.append(time.perf_counter() - t0)
        t_page += page_calls[-1]
        for p in pins:
            p.close()
    ix.close()
    return {"claims": n_chunks * chunk, "pinned_buffers_claims_per_s": n_chunks * chunk / t_pin,
            "pageable_buffers_claims_per_s": n_chunks * chunk / t_page,
            "pinned_call_ms": [round(1e3 * t, 3) for t in pin_calls], "pageable_call_ms": [round(1e3 * t, 3) for t in page_calls],
            "pinned_buffers_claims_per_s_median_call": chunk / float(np.median(pin_calls)),
            "pageable_buffers_claims_per_s_median_call": chunk / float(np.median(page_calls)),
            "note": "1 M-claim calls, ~112 B of text + 56 B of metadata per claim; K1 alone runs at ~430 M claims/s (token memo warm); every chunk is added, retracted whole and added again, so the id map doubles now and then (the slow call)"}


def time_single(ix, qv, k, since, tmask, reps=60):
    st = _hip.Stats()
    st.want_timing = 1
    for i in range(5):
        ix.search_shard(qv[i % len(qv)], k, since, tmask, st)
    wall, scan, cand = [], [], 0
    plain = _hip.Stats()  # latency is taken on calls WITHOUT device events (two event records cost ~6 us of queue time)
    for i in range(reps):
        t0 = time.perf_counter()
        ix.search_shard(qv[i % len(qv)], k, since, tmask, plain)
        wall.append(time.perf_counter() - t0)
        ix.search_shard(qv[i % len(qv)], k, since, tmask, st)
        scan.append(st.scan_ms * 1e-3)
        cand += st.candidates
    n = ix.size()
    bpr = st.bytes_per_row  # 256 (fp32 matrix) or 128 (fp16 copy), + 8 per fused mask
    s = float(np.median(scan))
    return {"claims": n, "top_k": k, "p50_ms": float(np.percentile(wall, 50) * 1e3), "p99_ms": float(np.percentile(wall, 99) * 1e3),
            "scan_us": s * 1e6, "bytes_per_claim": bpr, "scan_GBps": n * bpr / s / 1e9, "frac_of_8TBps": n * bpr / s / 8e12,
            "claims_per_s": n / float(np.median(wall)), "avg_candidates": cand / reps, "path": st.path}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", type=int, default=10_000_000)
    args = ap.parse_args()
    out = {"device": _hip.device_name()}
    queries = synth.make_queries(256, seed=4321)
    qv = _hip.embed_batch(queries)

    # config 1: the Python API on ~1k claims
    ch = synth.make_chunk(0, 1000, seed=1234)
    g = ContextGraph()
    texts, hashes = ch.texts(), ch.hashes()
    t0 = time.perf_counter()
    for i in range(ch.n):
        g.add_claim(hashes[i], texts[i], ch.tags(i), created_ms=int(ch.created_ms[i]))
    g.compile(queries[0])
    t_first = time.perf_counter() - t0
    lat = []
    for i in range(200):
        t0 = time.perf_counter()
        g.compile(queries[i % 64], top_k=8)
        lat.append(time.perf_counter() - t0)
    out["config1_1k_python_api"] = {"claims": 1000, "add_1000_claims_plus_first_compile_ms": t_first * 1e3,
                                    "compile_p50_ms": float(np.percentile(lat, 50) * 1e3),
                                    "compile_p99_ms": float(np.percentile(lat, 99) * 1e3)}

    # config 2: 1 M claims, single query
    ix1, _ = build_index(1_000_000)
    out["config2_1M_single_query"] = {"k8": time_single(ix1, qv, 8, _hip.SINCE_NONE, 0),
                                      "k64": time_single(ix1, qv, 64, _hip.SINCE_NONE, 0)}
    # at this size the scan's last block runs the select step (scan_fuse, grid <= one block per CU), so scan_us above is
    # the whole fused launch (scan + select); the two-kernel flavour gives the scan alone and the previous p50
    _hip.set_option("scan_fuse", 0)
    out["config2_1M_single_query"]["two_kernels_scan_fuse_0"] = {"k8": time_single(ix1, qv, 8, _hip.SINCE_NONE, 0),
                                                                 "k64": time_single(ix1, qv, 64, _hip.SINCE_NONE, 0)}
    _hip.set_option("scan_fuse", 1)
    out["config2_1M_single_query"]["note"] = ("k8/k64: default (fused select; scan_us, scan_GBps and frac_of_8TBps are of the "
                                              "fused launch, select included); two_kernels_scan_fuse_0: scan_us is the scan kernel alone")
    ix1.close()

    # big index: configs 3, 4 (per-GPU leg), 5
    n = args.big
    t0 = time.perf_counter()
    ix, t_add = build_index(n)
    t_build = time.perf_counter() - t0
    out["config5_bulk_vectorise"] = {"claims": n, "tokens_avg": 16, "index_add_s_incl_pack_h2d": t_add,
                                     "claims_per_s_incl_h2d": n / t_add, "host_generation_plus_add_s": t_build,
                                     "note": "kernel-only rate: see embed_kernel in the rocprof kernel stats"}
    out["config5_ingest_prepacked"] = ingest_prepacked()
    p50 = synth.BASE_MS + 1000 * (n // 2)
    p90 = synth.BASE_MS + 1000 * (9 * n // 10)
    out["config4_shard_single_query"] = {"k8": time_single(ix, qv, 8, _hip.SINCE_NONE, 0),
                                         "k64": time_single(ix, qv, 64, _hip.SINCE_NONE, 0)}
    _hip.set_option("scan_mode", 1)  # opt-in: scans read the 128 B/claim fp16 copy (results identical)
    out["config4_shard_single_query_fp16_copy"] = {"k8": time_single(ix, qv, 8, _hip.SINCE_NONE, 0)}
    _hip.set_option("scan_mode", 0)
    out["config5_filtered_queries"] = {
        "since_p50_k8": time_single(ix, qv, 8, p50, 0), "since_p90_k8": time_single(ix, qv, 8, p90, 0),
        "tags3_k8": time_single(ix, qv, 8, _hip.SINCE_NONE, 0b111), "since_p50_tags3_k8": time_single(ix, qv, 8, p50, 0b111)}
    # config 3: batched, in each scan precision (bq_mode 2 = fp16 is the default)
    st = _hip.Stats()
    st.want_timing = 1
    single = out["config4_shard_single_query"]["k8"]["p50_ms"]
    res = {}
    for mode, name, mfma_per_pair_dim in ((2, "fp16", 1), (1, "bf16_split3", 3), (0, "fp32_mfma", 1)):
        _hip.set_option("bq_mode", mode)
        for nq in ((256, 64, 16) if mode == 2 else (256,)):
            ix.search(qv[:nq], 8, _hip.SINCE_NONE, 0, st)
            wall, scan, err = [], [], 0.0
            plain = _hip.Stats()  # wall time on calls without device events (they add a gap in front of the first scan kernel)
            for r in range(8):
                t0 = time.perf_counter()
                ix.search(qv[:nq], 8, _hip.SINCE_NONE, 0, plain)
                wall.append(time.perf_counter() - t0)
                ix.search(qv[:nq], 8, _hip.SINCE_NONE, 0, st)
                scan.append(st.scan_ms * 1e-3)
                err = max(err, st.max_score_err)
            w, s = float(np.median(wall)), float(np.median(scan))
            useful = 2.0 * 64 * nq * n            # flops of the scores themselves
            issued = useful * mfma_per_pair_dim   # flops the matrix pipe executes (3 products in the split-bf16 mode)
            peak = 157.3e12 if mode == 0 else 2500e12
            res[f"{name}_q{nq}"] = {"claims": n, "queries": nq, "top_k": 8, "wall_ms": w * 1e3, "scan_ms": s * 1e3,
                                    "pairs_per_s": n * nq / w, "useful_TFLOPs": useful / s / 1e12,
                                    "mfma_TFLOPs_issued": issued / s / 1e12, "frac_of_dense_mfma_peak": issued / s / peak,
                                    "mfma_peak_TFLOPs": peak / 1e12, "scan_GBps_of_claim_bytes": n * st.bytes_per_row / s / 1e9,
                                    "max_score_err": err, "path": st.path, "fallbacks": st.fallbacks,
                                    "vs_looping_single_query_path": single * nq / (w * 1e3)}
    _hip.set_option("bq_mode", 2)
    _hip.set_option("scan_mode", 0)
    out["config3_batched_mfma"] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
