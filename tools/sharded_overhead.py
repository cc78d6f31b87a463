#!/usr/bin/env python3
"""Per-query cost of the multi-GPU exchange measured on ONE GPU with a one-rank RCCL communicator behind the C ABI
(lbh_index_search_sharded): everything except the inter-GPU hop itself -- the packed list written by the select kernel,
the gated exact-path launches, ncclAllGather on the library's stream, the publish kernel, the host merge -- is the same
as at N > 1.  Compared with lbh_index_search_shard on the same index.
Usage: python3 tools/sharded_overhead.py [claims]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lbm_amd import _hip, synth  # noqa: E402
from tools.run_configs import build_index  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ix, _ = build_index(n)
    qv = _hip.embed_batch(synth.make_queries(64, seed=4321))
    comm = _hip.Comm(_hip.comm_unique_id(), 0, 1)
    out = {"claims": n, "exchange": "lbh_index_search_sharded, 1-rank RCCL communicator"}
    for k in (8, 64):
        for i in range(30):
            ix.search_sharded(comm, qv[i % 64], k, _hip.SINCE_NONE, 0)
            ix.search_shard(qv[i % 64], k, _hip.SINCE_NONE, 0)
        a, b = [], []
        for i in range(400):
            t0 = time.perf_counter(); ix.search_shard(qv[i % 64], k, _hip.SINCE_NONE, 0); a.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); ix.search_sharded(comm, qv[i % 64], k, _hip.SINCE_NONE, 0); b.append(time.perf_counter() - t0)
        out[f"k{k}"] = {"local_p50_us": float(np.percentile(a, 50) * 1e6), "sharded_p50_us": float(np.percentile(b, 50) * 1e6),
                        "exchange_overhead_us": float((np.percentile(b, 50) - np.percentile(a, 50)) * 1e6)}
    print(json.dumps(out, indent=1))
    comm.close()


if __name__ == "__main__":
    main()
