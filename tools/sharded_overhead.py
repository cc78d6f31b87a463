#!/usr/bin/env python3
"""Per-step cost of the multi-GPU exchange (pack + H2D + RCCL all-gather + D2H + merge) measured on ONE GPU with a
1-rank RCCL group (always_gather=True): everything except the inter-GPU ring latency is the same as at N > 1."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from lbm_amd import _hip, synth  # noqa: E402
from lbm_amd.sharded import ShardedSearcher  # noqa: E402
from tools.run_configs import build_index  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ix, _ = build_index(n)
    qv = _hip.embed_batch(synth.make_queries(64, seed=4321))
    s = ShardedSearcher(ix.search_shard, 0, 1, device=torch.device("cuda", 0), always_gather=True)
    out = {"claims": n}
    for k in (8, 64):
        for i in range(20):
            s.search(qv[i % 64], k, _hip.SINCE_NONE, 0)
        a, b = [], []
        for i in range(300):
            t0 = time.perf_counter(); ix.search_shard(qv[i % 64], k, _hip.SINCE_NONE, 0); a.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); s.search(qv[i % 64], k, _hip.SINCE_NONE, 0); b.append(time.perf_counter() - t0)
        out[f"k{k}"] = {"local_p50_us": float(np.percentile(a, 50) * 1e6), "sharded_p50_us": float(np.percentile(b, 50) * 1e6),
                        "exchange_overhead_us": float((np.percentile(b, 50) - np.percentile(a, 50)) * 1e6)}
    # host-side cost of each phase of ShardedSearcher.search (k = 8), same calls in the same order
    from lbm_amd.sharded import pack_local
    b = s._bufs[8]
    ph = {x: [] for x in ("pack", "h2d_launch", "allgather_launch", "d2h_launch", "sync", "merge")}
    for i in range(300):
        ids, sc, keys = ix.search_shard(qv[i % 64], 8, _hip.SINCE_NONE, 0)
        t = [time.perf_counter()]
        local = pack_local(sc, s.gid_of_local(np.asarray(ids)), keys, 8); b.send_np[...] = local; t.append(time.perf_counter())
        b.send_dev.copy_(b.send_host, non_blocking=True); t.append(time.perf_counter())
        dist.all_gather_into_tensor(b.recv_dev, b.send_dev); t.append(time.perf_counter())
        b.recv_host.copy_(b.recv_dev, non_blocking=True); t.append(time.perf_counter())
        torch.cuda.current_stream().synchronize(); t.append(time.perf_counter())
        s._merge(b.recv_np, 8); t.append(time.perf_counter())
        for j, x in enumerate(ph):
            ph[x].append(t[j + 1] - t[j])
    out["phases_p50_us"] = {x: float(np.percentile(v, 50) * 1e6) for x, v in ph.items()}
    print(json.dumps(out, indent=1))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
