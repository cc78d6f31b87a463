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

``--scaling strong`` keeps the TOTAL corpus fixed (``--claims`` claims dealt out in N contiguous parts); the default,
``weak``, gives every GPU ``--claims`` claims.  The printed line says which (``scaling``, ``config.workload``).

``latency_ms`` carries two pairs: ``p50`` / ``p99`` time the ranking call on a query VECTOR (lbh_index_search_shard, the
step that ``value`` counts), ``compile_context_p50`` / ``_p99`` time what the metric names -- query TEXT in, context
TEXT out: embed(query) on the GPU (K1) + scan (K2) + select / exact re-rank (K4) in one stream-ordered call
(lbh_index_search_text), then claim id -> hash and the formatting of lb/context_graph.py:122-133 for the k winners.

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
READ_ONLY_KERNEL_GBS = 6950.0  # what a read-only streaming kernel reaches on this part (tools/hbm_read_rate.hip: 6.8-7.04 TB/s, profiles/r03_hbm_read_rate.txt)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--claims", type=int, default=10_000_000,
                   help="claims per GPU (--scaling weak, the default) or in total, split over the GPUs (--scaling strong)")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="weak: every GPU holds --claims claims; strong: --claims is the whole corpus, dealt out in N contiguous parts")
    p.add_argument("--cc-steps", type=int, default=200, help="timed compile_context calls (text in, context text out) of the latency leg, N = 1")
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


VALU_ISSUE_PER_S = 256 * 4 * 2.4e9 / 2  # wave-instructions/s: 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md)
PCIE_GBS = 57.0                         # pinned H2D rate measured on the pool's boxes (tools/h2d_rate.py)


def vectorise_roofline(k1, claims, t_add, t_add_first, chunk, n_chunks):
    """K1 (embed_kernel) is bound by VALU issue, not by HBM: its achieved fraction is wave-instructions per claim (from
    the committed counter pass, profiles/r*_pmc_k1.json: SQ_INSTS_VALU per claim of this very corpus) x claims/s of the
    kernel alone (text resident in HBM, measured here) / the chip's VALU issue rate.  The ingest call (text + metadata
    from pinned host buffers) is priced against PCIe; HBM traffic (text in + 768 B out per claim) is given for scale."""
    if not k1:
        return None
    out = {"bound": "valu-issue", "kernel": "embed_kernel", "kernel_claims_per_s": k1["claims"] / k1["seconds"],
           "kernel_ms_per_million_claims": k1["seconds"] / k1["claims"] * 1e9,
           "peak": VALU_ISSUE_PER_S, "unit": "wave-instructions/s"}
    try:
        import glob
        pmc = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_k1.json")))
        with open(pmc[-1]) as f:
            ek = json.load(f)["embed_kernel"]
        per_claim = float(ek["valu_wave_instructions_per_claim"])
        out["valu_wave_instructions_per_claim"] = per_claim
        # same pass: share of SIMD-cycles with a vector instruction executing (these integer instructions occupy the pipe
        # for ~4 cycles each, so the pipe is busier than the 2-cycle issue model says)
        out["valu_busy_fraction_pmc"] = ek.get("valu_busy_fraction")
        cold = json.load(open(pmc[-1])).get("first_launch_cold_memo")
        if cold:  # the token memo removes work (hash compressions), it does not raise the issue rate: both counts are given
            out["cold_memo_first_launch"] = cold
            out["note"] = ("per-claim instruction count and busy fraction are of a launch with the token memo warm (steady "
                           "state of an ingest stream); cold_memo_first_launch = every token hashed")
        out["instructions_source"] = f"profiles/{os.path.basename(pmc[-1])} (SQ_INSTS_VALU / claims, same corpus); not measured in this run"
        out["achieved"] = per_claim * out["kernel_claims_per_s"]
        out["frac"] = out["achieved"] / VALU_ISSUE_PER_S
    except Exception:
        out["achieved"] = out["frac"] = None
    bytes_per_claim_hbm = k1["text_bytes"] / k1["claims"] + 8 + 768
    out["hbm_floor_claims_per_s"] = HBM_PEAK_GBS * 1e9 / bytes_per_claim_hbm
    bytes_per_claim_pcie = k1["text_bytes"] / k1["claims"] + 8 + 48  # text + offset + created / tags / key
    out["pcie_floor_claims_per_s"] = PCIE_GBS * 1e9 / bytes_per_claim_pcie
    steady = (claims - min(chunk, claims)) / (t_add - t_add_first) if n_chunks > 1 and t_add > t_add_first else None
    out["ingest_claims_per_s_steady"] = steady
    out["ingest_frac_of_pcie_floor"] = steady / out["pcie_floor_claims_per_s"] if steady else None
    return out


def bring_up_rccl(_hip, dist, torch, rank: int, world: int):
    """-> (Comm or None, reason, stuck).  Every step that can fail on ONE rank alone is agreed on over gloo BEFORE the
    collective ncclCommInitRank, so that no rank waits inside it for a peer that has already given up:
      1. every rank opens librccl and resolves its symbols locally (lbh_comm_unique_id does exactly that);
      2. all-reduce(MIN) of the outcome -- if any rank cannot, nobody calls RCCL;
      3. rank 0's id is broadcast, every rank calls Comm() (collective) -- in a helper thread with a deadline, because a
         bring-up that never returns must end in the gloo fallback, not in a hung run;
      4. all-reduce(MIN) of that outcome.  On a timeout the helper thread is abandoned (``stuck``): the process then
         leaves through os._exit at the end instead of joining it."""
    import threading
    err = ""
    try:
        my_id = _hip.comm_unique_id()  # dlopen + dlsym + ncclGetUniqueId: local, no peer involved
    except _hip.HipBackendError as e:
        my_id, err = b"", str(e)
    ok = torch.tensor([1 if my_id else 0], dtype=torch.int64)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return None, err or "another rank cannot open librccl", False
    uid = [my_id if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    box = {}

    def init():
        try:
            box["comm"] = _hip.Comm(uid[0], rank, world)  # collective: RCCL communicator over the node's xGMI links
        except Exception as e:  # noqa: BLE001
            box["err"] = str(e)

    t = threading.Thread(target=init, daemon=True)
    t.start()
    t.join(float(os.environ.get("LBM_COMM_INIT_TIMEOUT_S", "180")))
    stuck = t.is_alive()
    comm = None if stuck else box.get("comm")
    ok = torch.tensor([1 if comm is not None else 0, 1 if stuck else 0], dtype=torch.int64)
    flags = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(flags, ok)
    all_ok = all(int(f[0]) == 1 for f in flags)
    any_stuck = any(int(f[1]) == 1 for f in flags)
    if all_ok:
        return comm, "", False
    # some rank failed or hangs: a communicator that did come up on this rank cannot be torn down collectively any more
    why = box.get("err") or ("ncclCommInitRank did not return in time" if stuck else "another rank failed to join")
    return None, why, True if (comm is not None or stuck or any_stuck) else False


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
    if args.scaling == "strong":  # the whole corpus is --claims claims; rank r takes the r-th contiguous part
        first = args.claims * rank // world
        my_claims = args.claims * (rank + 1) // world - first
    else:
        first, my_claims = rank * args.claims, args.claims
    ix = _hip.Index()
    ix.reserve(my_claims)
    chunk = 1_000_000
    n_chunks = (my_claims + chunk - 1) // chunk
    t_gen = t_add = t_add_first = 0.0
    done = 0
    # The chunk's arrays are generated straight into page-locked buffers (lbh_host_alloc), so the ingest call moves them
    # by asynchronous DMA under K1 (DESIGN.md "Ingest path"); filling the buffers is part of the generation time.
    pins = None
    if have_gpu:
        cap = min(chunk, max(my_claims, 1))
        pins = {"blob": _hip.PinnedArray(cap * 175 + 64, np.uint8), "offsets": _hip.PinnedArray(cap + 1, np.int64),
                "created": _hip.PinnedArray(cap, np.int64), "tags": _hip.PinnedArray(cap, np.uint64),
                "keys": _hip.PinnedArray((cap, 32), np.uint8)}
    host_chunks = []  # (blob, offsets, keys, tagmask) per chunk, kept on the host for the compile_context leg (N = 1 only)
    for ci in range(n_chunks):
        m = min(chunk, my_claims - done)
        t0 = time.perf_counter()
        ch = synth.make_chunk(first + done, m, seed=1234, chunk_index=rank * 4096 + ci, real_sha_keys=False)  # chunk seeds never repeat across ranks
        arrs = (ch.blob, ch.offsets, ch.created_ms, ch.tagmask, ch.keys)
        if pins is not None and ch.blob.shape[0] <= pins["blob"].a.shape[0]:
            nb = ch.blob.shape[0]
            pins["blob"].a[:nb] = ch.blob
            pins["offsets"].a[: m + 1] = ch.offsets
            pins["created"].a[:m] = ch.created_ms
            pins["tags"].a[:m] = ch.tagmask
            pins["keys"].a[:m] = ch.keys
            arrs = (pins["blob"].a[:nb], pins["offsets"].a[: m + 1], pins["created"].a[:m], pins["tags"].a[:m], pins["keys"].a[:m])
        t1 = time.perf_counter()
        ids = ix.add_texts(*arrs)
        ix.retract_many(ids[ch.retracted])  # 1 % of the claims: one compaction kernel per chunk
        t2 = time.perf_counter()
        t_gen += t1 - t0
        t_add += t2 - t1
        if ci == 0:
            t_add_first = t2 - t1  # carries the one-off costs: code object load, scratch and staging allocation
        if world == 1:
            host_chunks.append((ch.blob, ch.offsets, ch.keys, ch.tagmask))
        done += m
    live = ix.size()
    base = synth.BASE_MS + 1000 * first
    since = {"none": _hip.SINCE_NONE, "p50": base + 1000 * (my_claims // 2),
             "p90": base + 1000 * (9 * my_claims // 10)}[args.since]

    queries = synth.make_queries(args.queries, seed=4321)
    qvecs = _hip.embed_batch(queries)
    searcher = comm = None
    exchange_note = ""
    comm_stuck = False
    if world > 1 and rehearse:
        searcher = ShardedSearcher(ix.search_shard, rank, world, device=None)
    elif world > 1:
        comm, comm_err, comm_stuck = bring_up_rccl(_hip, dist, torch, rank, world)
        if comm is None:
            # RCCL could not be brought up on every rank: the run still measures the sharded search, with the lists
            # travelling over gloo instead (said so on the printed line); never silently
            exchange_note = "RCCL unavailable (%s): lists gathered over gloo" % comm_err
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

    if comm is not None:
        # canary: the first all-gather over xGMI this communicator ever carries.  Its outcome is agreed on over gloo, so
        # a data path that does not work (error, or the library's bounded wait running out) ends in the gloo exchange on
        # EVERY rank, said so on the printed line -- not in a hang and not in a half-failed run.
        canary_err = ""
        try:
            step(0)
        except _hip.HipBackendError as e:
            canary_err = str(e)
        ok = torch.tensor([0 if canary_err else 1], dtype=torch.int64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm, comm_stuck = None, True  # never torn down: a collective may still be queued on it
            exchange_note = "RCCL all-gather failed (%s): lists gathered over gloo" % (canary_err or "on another rank")
            print(f"bench.py rank {rank}: {exchange_note}", file=sys.stderr)
            searcher = ShardedSearcher(ix.search_shard, rank, world, device=None)
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

    # ---- what the metric names: compile_context latency, query TEXT in -> context TEXT out (N = 1) ----------------
    cc = None
    if world == 1 and args.cc_steps > 0 and host_chunks:
        from lbm_amd.context_graph import format_context

        # The claim store of this leg: the generated chunks as they are (a dict of 10 M Claim objects would not be the
        # thing measured).  Buffer views, so that fetching a chosen claim costs what a dict lookup of a stored Claim
        # would (indexing the numpy arrays per element and testing all 64 tag bits took longer than the formatting).
        tag_names = ["t%02d" % b for b in range(64)]
        views = [(memoryview(b), memoryview(o), memoryview(np.ascontiguousarray(k).reshape(-1)), memoryview(t))
                 for b, o, k, t in host_chunks]

        def claim(cid):  # claim id = position in the generated corpus (ids are handed out in insertion order)
            blob, off, keys, tagmask = views[cid // chunk]
            j = cid % chunk
            m, tags = tagmask[j], []
            while m:
                low = m & -m
                tags.append(tag_names[low.bit_length() - 1])
                m ^= low
            return keys[32 * j: 32 * j + 32].hex(), str(blob[off[j]: off[j + 1]], "ascii"), tags

        def compile_context(query):  # lb/node.py:718-734 -> lb/context_graph.py:104-133 on the HBM index
            ids, _, _ = ix.search_text(query, args.top_k, since, 0)  # embed(query) + scan + exact top-k: one call, one wait
            chosen = [claim(int(i)) for i in ids]
            return format_context(query, [(t, tg) for _, t, tg in chosen]), [h for h, _, _ in chosen]

        for i in range(min(args.warmup, 20)):
            compile_context(queries[i % len(queries)])
        cc_lat = []
        for i in range(args.cc_steps):
            s0 = time.perf_counter()
            text, hashes = compile_context(queries[i % len(queries)])
            cc_lat.append(time.perf_counter() - s0)
        # the same answer as the vector entry the timed steps use (ids -> hashes), so both legs rank identically
        ids0, _, keys0 = ix.search_shard(qvecs[(args.cc_steps - 1) % len(queries)], args.top_k, since, 0)
        assert hashes == [bytes(k).hex() for k in keys0], "search_text and search_shard disagree"
        cc = {"p50": float(np.percentile(cc_lat, 50) * 1e3), "p99": float(np.percentile(cc_lat, 99) * 1e3), "calls": args.cc_steps,
              "context_bytes": len(text)}

    # ---- K1 alone, text resident in HBM (vectorise.roofline) -------------------------------------------------------
    k1 = None
    if world == 1 and have_gpu and host_chunks:
        blob, off = host_chunks[0][0], host_chunks[0][1]
        m1 = off.shape[0] - 1
        d_blob = torch.from_numpy(np.concatenate([blob, np.zeros(64, np.uint8)])).cuda()
        d_off = torch.from_numpy(off).cuda()
        d_e64 = torch.empty((m1, 64), dtype=torch.float64, device="cuda")
        d_e32 = torch.empty((m1, 64), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            s0 = time.perf_counter()
            _hip.embed_batch_dev(d_blob.data_ptr(), d_off.data_ptr(), m1, int(off[-1]), d_e64.data_ptr(), d_e32.data_ptr())
            ts.append(time.perf_counter() - s0)
        t_k1 = float(np.median(ts[2:]))
        k1 = {"claims": m1, "text_bytes": int(off[-1]), "seconds": t_k1}
        del d_blob, d_off, d_e64, d_e32

    if rank == 0:
        bytes_per_row = 256 + (8 if args.since != "none" else 0)
        # HBM bytes per launch: NOT collected in this run (PMC passes need rocprofv3 around the process) -- the newest
        # committed counter pass (profiles/*pmc_scan.json: FETCH_SIZE x 2 + WRITE_SIZE per row) scaled to this row count
        traffic, traffic_source = None, None
        try:
            import glob
            pmc = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_scan.json")))
            if pmc and args.since == "none":
                with open(pmc[-1]) as f:
                    traffic = json.load(f)["scan_f32_kernel"]["hbm_bytes_per_row"] * live
                traffic_source = f"profiles/{os.path.basename(pmc[-1])} (committed rocprofv3 --pmc passes) x {live} live rows; not measured in this run"
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
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32 scan + f64 exact re-rank",
            "data": "synthetic" if not rehearse else "synthetic; REHEARSAL (all ranks on one GPU, gloo): not a measurement",
            "config": {"workload": (f"{args.claims}-claim shard per GPU" if args.scaling == "weak" else
                                    f"{args.claims}-claim corpus in total, split into {world} contiguous shards (strong scaling)")
                                   + " (synthetic Zipf corpus, 1% retracted), "
                                   f"single-query compile_context ranking, top_k={args.top_k}, since_ms={args.since}",
                       "claims_per_gpu": my_claims, "claims_total": args.claims * world if args.scaling == "weak" else args.claims,
                       "live_claims_total": total_live, "top_k": args.top_k,
                       "parallelism": (f"claim-sharded x{world}, one RCCL all-gather of the per-GPU top-k lists per query (inside liblbhip), "
                                       "same-comparator merge" if comm is not None else
                                       f"claim-sharded x{world}, lists gathered over gloo ({exchange_note or 'rehearsal'})")
                       if world > 1 else "single GPU"},
            "latency_ms": {"p50": float(np.percentile(lat, 50) * 1e3), "p99": float(np.percentile(lat, 99) * 1e3),
                           "what": "p50/p99: the ranking call on a query vector (the timed step); compile_context_*: query text -> "
                                   "embed on the GPU + scan + exact top-k (lbh_index_search_text) -> hashes + formatted context text",
                           "compile_context_p50": cc["p50"] if cc else None, "compile_context_p99": cc["p99"] if cc else None,
                           "compile_context_calls": cc["calls"] if cc else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": live * bytes_per_row, "kernel": "scan_f32_kernel",
                         "bytes_per_claim": bytes_per_row, "avg_kernel_us": avg_scan_s * 1e6,
                         "timed_launches": n_timed,
                         "read_only_kernel_GBs": READ_ONLY_KERNEL_GBS, "frac_of_read_only_kernel": achieved / READ_ONLY_KERNEL_GBS,
                         "read_only_kernel_source": "profiles/r03_hbm_read_rate.txt (tools/hbm_read_rate.hip, same access pattern without the arithmetic; not measured in this run)"},
            "vectorise": {"claims_per_s_incl_h2d": my_claims / t_add if t_add > 0 else None,
                          "claims_per_s_incl_h2d_after_first_chunk":
                              (my_claims - min(chunk, my_claims)) / (t_add - t_add_first) if n_chunks > 1 else None,
                          "first_chunk_s": t_add_first,
                          "source_buffers": "pinned (lbh_host_alloc)" if pins is not None else "pageable",
                          "host_text_generation_s": t_gen,
                          "roofline": vectorise_roofline(k1, my_claims, t_add, t_add_first, chunk, n_chunks)},
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
    if comm_stuck:
        # a communicator that never came up everywhere (or a collective still queued on one): nothing to tear down in
        # order -- leave without running destructors that would wait for it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
