#!/usr/bin/env python3
"""bench.py -- claims scored/sec of compile_context's ranking on MI355X (BASELINE.json metric).

Workload (``config.workload``): a synthetic claim corpus (SURVEY.md §8d distribution) of
``--claims`` claims PER GPU (default 10 M, the size the metric is quoted on), resident in HBM;
one step = one single-query top-k search over the whole corpus: K2 fp32 scan with the LDS
top-k, K4 exact fp64 re-rank, and for N > 1 the all-gather merge of the per-GPU lists.
value = claims scored per second over all GPUs = (claims per GPU * N * steps) / wall time,
wall time = max over ranks between two barrier + synchronize brackets.

Extra objects on the JSON line:
  roofline      dominant kernel (scan_f32_kernel): algorithmic bytes = live rows * 256 B per launch,
                divided by the kernel's mean HIP-event duration on the library's stream.
  cpu_baseline  the CPU oracle (oracle/lb_oracle.c, a port of the reference's algorithm) timed on
                one host core on a bounded sample of the same workload (rank 0, N = 1 only).
  fp16_prefilter  side leg, N = 1 only, never `value`: the same steps with the opt-in scan_mode 1 (the scan reads a
                128 B/claim fp16 copy of the matrix; the exact fp64 re-rank and so the results are unchanged).

Launching: ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(one child process per GPU, before anything touches a GPU) and waits for them; under torchrun
(``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``) each process is one rank.  Either way
the run fails if the number of ranks is not ``--gpus``.

LBM_BENCH_REHEARSE_GLOO=1 rehearses the N > 1 flow on a box with fewer GPUs than ranks (see main()).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--claims", type=int, default=10_000_000, help="claims per GPU")
    p.add_argument("--top-k", type=int, default=8)
    p.add_argument("--since", choices=["none", "p50", "p90"], default="none")
    p.add_argument("--queries", type=int, default=64)
    p.add_argument("--time-every", type=int, default=4, help="record HIP events around the scan on every N-th timed step")
    p.add_argument("--cpu-sample", type=int, default=2_000_000, help="claims in the CPU-baseline sample (0 = skip)")
    return p.parse_args()


def host_cores() -> int:
    """Worker threads for the all-cores leg: the box's CPU share (OMP_NUM_THREADS is set to it there), not the
    whole machine that os.cpu_count() reports."""
    for var in ("LBM_BENCH_CORES", "OMP_NUM_THREADS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            return min(int(v), os.cpu_count() or 1)
    return min(16, os.cpu_count() or 1)


def cpu_baseline(sample_claims: int, top_k: int):
    """Oracle (port of lb/latent.py + ContextGraph.compile) on a bounded sample: one core (the reference is a
    single-threaded program) and all cores (claim-sharded threads + a top-k merge)."""
    from concurrent.futures import ThreadPoolExecutor
    from lbm_amd import synth
    from lbm_amd.sharded import merge_gathered, pack_local
    from oracle import lb_oracle as orc
    cores = host_cores()
    ch = synth.make_chunk(0, sample_claims, seed=1234, chunk_index=0, real_sha_keys=False)
    n = ch.n
    cuts = [n * i // cores for i in range(cores + 1)]
    # embedding: the whole sample on all cores (ctypes drops the GIL), a tenth of it on one core for the 1-core rate
    emb = np.zeros((n, 64), dtype=np.float64)

    def embed_part(i):
        a, b = cuts[i], cuts[i + 1]
        off = ch.offsets[a: b + 1]
        emb[a:b] = orc.embed_batch_packed(ch.blob[off[0]: off[-1] + 1], off - off[0])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(embed_part, range(cores)))
    t_embed_all = time.perf_counter() - t0
    n1 = max(1, n // 10)
    t0 = time.perf_counter()
    e1 = orc.embed_batch_packed(ch.blob[: ch.offsets[n1] + 1], ch.offsets[: n1 + 1])
    t_embed_1 = time.perf_counter() - t0
    assert e1.tobytes() == emb[:n1].tobytes()
    keys_hex = ch.hashes()
    packed = orc.pack_texts(keys_hex)
    retr = ch.retracted.astype(np.uint8)
    queries = synth.make_queries(8, seed=4321)
    qv = [orc.embed_array(q) for q in queries]
    t0 = time.perf_counter()
    one = [orc.compile_arrays(emb, ch.created_ms, retr, None, v, top_k, None, packed_keys=packed) for v in qv]
    t_rank = time.perf_counter() - t0
    # all cores: each thread ranks its contiguous shard, the per-shard top-k lists are merged like the GPUs' lists
    parts = []
    for i in range(cores):
        a, b = cuts[i], cuts[i + 1]
        parts.append((a, np.ascontiguousarray(emb[a:b]), np.ascontiguousarray(ch.created_ms[a:b]), np.ascontiguousarray(retr[a:b]),
                      orc.pack_texts(keys_hex[a:b]), ch.keys[a:b]))

    def rank_part(arg):
        (a, e, cm, rt, pk, kb), v = arg
        idx, sc = orc.compile_arrays(e, cm, rt, None, v, top_k, None, packed_keys=pk)
        return pack_local(sc, (idx + a).astype(np.uint64), kb[idx], top_k)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        many = []
        for v in qv:
            lists = list(ex.map(rank_part, [(p, v) for p in parts]))
            many.append(merge_gathered(np.stack(lists), top_k))
    t_rank_all = time.perf_counter() - t0
    for (i1, s1), (ia, sa, _) in zip(one, many):
        assert i1.tolist() == ia.tolist() and s1.tobytes() == sa.tobytes()  # sharded == unsharded, bit for bit
    live = int((~ch.retracted).sum())
    return {
        "value": live * len(qv) / t_rank, "unit": "claims scored/s", "cores": 1, "kind": "port",
        "sample": f"{sample_claims} claims ({live} live) x {len(qv)} queries, top_k={top_k}: "
                  f"score every claim in fp64 + full sort like the reference ({t_rank:.2f} s on 1 core, "
                  f"{t_rank_all:.2f} s on {cores} cores, claim-sharded threads + top-k merge, identical results)",
        "all_cores": {"value": live * len(qv) / t_rank_all, "unit": "claims scored/s", "cores": cores},
        "embed_claims_per_s": n1 / t_embed_1,
        "embed_claims_per_s_all_cores": {"value": n / t_embed_all, "cores": cores},
    }


def launch_ranks(args) -> int:
    """``--gpus N`` without a launcher: start the N ranks as child processes (nothing in this process has touched a
    GPU), wait for all of them, fail if any fails.  Rank 0 prints the JSON line on the inherited stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:  # one rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but this process is one of WORLD_SIZE={world} ranks")
    # LBM_BENCH_REHEARSE_GLOO=1: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (every rank on GPU 0,
    # collectives over gloo on host tensors).  Not a measurement: the line it prints says so in `data`.
    rehearse = os.environ.get("LBM_BENCH_REHEARSE_GLOO") == "1"
    if rehearse:
        local_rank = 0
    device = local_rank  # LB_HIP_DEVICES ("0,1,2,3": rank LOCAL_RANK takes that entry) wins, as in lbm_amd._hip.default_device
    listed = [x.strip() for x in os.environ.get("LB_HIP_DEVICES", "").split(",") if x.strip()]
    if local_rank < len(listed):
        device = int(listed[local_rank])
    os.environ.setdefault("LBM_HIP_DEVICE", str(device))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    import torch
    import torch.distributed as dist
    have_gpu = torch.cuda.is_available()  # False only in the CPU rehearsal of the launcher (tests/bench_cpu_harness.py)
    if have_gpu or not rehearse:
        torch.cuda.set_device(device)
    coll_dev = "cpu"
    if world > 1:
        # torch.distributed is plumbing only (rendezvous, barriers, the max / sum of the timing scalars), over gloo on the
        # host; the data path's one collective -- the all-gather of the per-GPU top-k lists -- runs inside liblbhip on RCCL
        # (lbh_index_search_sharded).  The rehearsal has no second GPU, so there the lists travel over gloo as well.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")

    from lbm_amd import _hip, synth
    from lbm_amd.sharded import ShardedSearcher

    # ---- build this rank's shard in HBM (texts -> K1 on the GPU) -------------------------------
    ix = _hip.Index()
    ix.reserve(args.claims)
    chunk = 1_000_000
    n_chunks = (args.claims + chunk - 1) // chunk
    t_gen = t_add = t_add_first = 0.0
    created_p50 = created_p90 = None
    done = 0
    for ci in range(n_chunks):
        m = min(chunk, args.claims - done)
        t0 = time.perf_counter()
        ch = synth.make_chunk(rank * args.claims + done, m, seed=1234, chunk_index=rank * n_chunks + ci,
                              real_sha_keys=False)
        t1 = time.perf_counter()
        ids = ix.add_texts(ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
        ix.retract_many(ids[ch.retracted])  # 1 % of the claims: one compaction kernel per chunk
        t2 = time.perf_counter()
        t_gen += t1 - t0
        t_add += t2 - t1
        if ci == 0:
            t_add_first = t2 - t1  # carries the one-off costs: code object load, scratch and staging allocation
        done += m
    live = ix.size()
    base = synth.BASE_MS + 1000 * rank * args.claims
    since = {"none": _hip.SINCE_NONE, "p50": base + 1000 * (args.claims // 2),
             "p90": base + 1000 * (9 * args.claims // 10)}[args.since]

    queries = synth.make_queries(args.queries, seed=4321)
    qvecs = _hip.embed_batch(queries)
    searcher = comm = None
    exchange_note = ""
    if world > 1 and rehearse:
        searcher = ShardedSearcher(ix.search_shard, rank, world, device=None)
    elif world > 1:
        comm_err = ""
        try:
            uid = [_hip.comm_unique_id() if rank == 0 else None]
        except _hip.HipBackendError as e:  # librccl missing / unusable: every rank learns it through the broadcast
            uid, comm_err = [b""], str(e)
        dist.broadcast_object_list(uid, src=0)
        if uid[0]:
            try:
                comm = _hip.Comm(uid[0], rank, world)  # collective: RCCL communicator over the node's xGMI links
            except _hip.HipBackendError as e:
                comm_err = str(e)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            # RCCL could not be brought up on every rank: the run still measures the sharded search, with the lists
            # travelling over gloo instead (said so on the printed line); never silently
            if comm is not None:
                comm.close()
                comm = None
            exchange_note = "RCCL unavailable (%s): lists gathered over gloo" % (comm_err or "another rank failed")
            print(f"bench.py rank {rank}: {exchange_note}", file=sys.stderr)
            searcher = ShardedSearcher(ix.search_shard, rank, world, device=None)
    stats = _hip.Stats()
    every = max(1, args.time_every)

    def step(i):
        q = qvecs[i % len(queries)]
        if world == 1:
            # HIP events around the scan kernel on every `every`-th launch of the timed region (an event
            # record costs ~3 us of queue time, so the other launches run without them)
            stats.want_timing = 1 if i % every == 0 else 0
            return ix.search_shard(q, args.top_k, since, 0, stats)
        if comm is not None:
            return ix.search_sharded(comm, q, args.top_k, since, 0)
        return searcher.search(q, args.top_k, since, 0)

    def fence():
        if world > 1:
            dist.barrier()
        if have_gpu:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    scan_ms, lat = 0.0, []
    fallbacks = cands = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        s0 = time.perf_counter()
        step(i)
        lat.append(time.perf_counter() - s0)
        if world == 1:
            if i % every == 0:
                scan_ms += stats.scan_ms
            fallbacks += stats.fallbacks
            cands += stats.candidates
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        lv = torch.tensor([live], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(lv, op=dist.ReduceOp.SUM)
        total_live = int(lv.item())
    else:
        total_live = live

    if world > 1:  # roofline leg needs per-kernel timing: measure locally, outside the timed region
        stats.want_timing = 1
        for i in range(min(args.steps, 50)):
            ix.search_shard(qvecs[i % len(queries)], args.top_k, since, 0, stats)
            scan_ms += stats.scan_ms
        n_timed = min(args.steps, 50)
    else:
        n_timed = (args.steps + every - 1) // every

    if rank == 0:
        bytes_per_row = 256 + (8 if args.since != "none" else 0)
        traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/*pmc_scan.json), scaled to this row count
        try:
            import glob
            pmc = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_scan.json")))
            if pmc and args.since == "none":
                with open(pmc[-1]) as f:
                    traffic = json.load(f)["scan_f32_kernel"]["hbm_bytes_per_row"] * live
        except Exception:
            traffic = None
        avg_scan_s = scan_ms / 1e3 / max(n_timed, 1)
        achieved = live * bytes_per_row / avg_scan_s / 1e9 if avg_scan_s > 0 else 0.0
        out = {
            "metric": "claims scored/sec (compile_context ranking, single query, top-k)",
            "value": total_live * args.steps / elapsed,
            "unit": "claims/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 scan + f64 exact re-rank",
            "data": "synthetic" if not rehearse else "synthetic; REHEARSAL (all ranks on one GPU, gloo): not a measurement",
            "config": {"workload": f"{args.claims}-claim shard per GPU (synthetic Zipf corpus, 1% retracted), "
                                   f"single-query compile_context ranking, top_k={args.top_k}, since_ms={args.since}",
                       "claims_per_gpu": args.claims, "live_claims_total": total_live, "top_k": args.top_k,
                       "parallelism": (f"claim-sharded x{world}, one RCCL all-gather of the per-GPU top-k lists per query (inside liblbhip), "
                                       "same-comparator merge" if comm is not None else
                                       f"claim-sharded x{world}, lists gathered over gloo ({exchange_note or 'rehearsal'})")
                       if world > 1 else "single GPU"},
            "latency_ms": {"p50": float(np.percentile(lat, 50) * 1e3), "p99": float(np.percentile(lat, 99) * 1e3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": live * bytes_per_row, "kernel": "scan_f32_kernel",
                         "bytes_per_claim": bytes_per_row, "avg_kernel_us": avg_scan_s * 1e6,
                         "timed_launches": n_timed},
            "vectorise": {"claims_per_s_incl_h2d": args.claims / t_add if t_add > 0 else None,
                          "claims_per_s_incl_h2d_after_first_chunk":
                              (args.claims - min(chunk, args.claims)) / (t_add - t_add_first) if n_chunks > 1 else None,
                          "first_chunk_s": t_add_first,
                          "host_text_generation_s": t_gen},
            "fp32_fallbacks": fallbacks, "avg_candidates": cands / max(args.steps, 1),
        }
        if world == 1:
            # Side leg, never `value`: the same queries with the opt-in fp16 prefilter (scan_mode 1: the scan reads a
            # 128 B/claim fp16 copy with a wider margin; the fp64 re-rank and therefore the results are the same).
            _hip.set_option("scan_mode", 1)
            for i in range(args.warmup):
                ix.search_shard(qvecs[i % len(queries)], args.top_k, since, 0, stats)
            fence()
            h_ms, h_timed = 0.0, 0
            h0 = time.perf_counter()
            for i in range(args.steps):
                stats.want_timing = 1 if i % every == 0 else 0
                ix.search_shard(qvecs[i % len(queries)], args.top_k, since, 0, stats)
                if i % every == 0:
                    h_ms += stats.scan_ms
                    h_timed += 1
            fence()
            h_el = time.perf_counter() - h0
            _hip.set_option("scan_mode", 0)
            h_bpr = bytes_per_row - 128
            h_s = h_ms / 1e3 / max(h_timed, 1)
            out["fp16_prefilter"] = {"value": live * args.steps / h_el, "unit": "claims/s", "ms_per_step": h_el / args.steps * 1e3,
                                     "bytes_per_claim": h_bpr, "avg_kernel_us": h_s * 1e6,
                                     "achieved_GBs": live * h_bpr / h_s / 1e9 if h_s > 0 else 0.0,
                                     "frac": live * h_bpr / h_s / 1e9 / HBM_PEAK_GBS if h_s > 0 else 0.0,
                                     "note": "opt-in scan_mode=1; results identical (fp64 re-rank); not the headline"}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.top_k)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # no rank tears its communicator down while another may still be inside a collective
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
