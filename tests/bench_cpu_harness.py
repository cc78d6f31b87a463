"""CPU rehearsal of bench.py's launcher and N > 1 flow (test infrastructure, run by tests/test_bench_launcher.py).

bench.py itself has no CPU path: without a GPU ``lbm_amd._hip.Index()`` raises.  To exercise what does NOT need a GPU
-- ``--gpus N`` spawning N ranks, the gloo rendezvous, barriers, max-over-ranks timing, summed claim counts, one JSON
line from rank 0, the failure when WORLD_SIZE and --gpus disagree -- this script swaps the rank-local index for the
CPU oracle (tests may use it) and then runs bench.main() unchanged.  The numbers it prints mean nothing."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from lbm_amd import _hip  # noqa: E402
from oracle import lb_oracle as orc  # noqa: E402


class OracleIndex:
    def __init__(self):
        self.emb, self.created, self.keys, self.live = [], [], [], []

    def reserve(self, n):
        pass

    def add_texts(self, blob, offsets, created_ms, tagmask, keys):
        first = sum(len(c) for c in self.created)
        self.emb.append(orc.embed_batch_packed(blob, offsets))
        self.created.append(np.asarray(created_ms))
        self.keys.append(np.asarray(keys).reshape(-1, 32))
        self.live.append(np.ones(len(created_ms), dtype=bool))
        return np.arange(first, first + len(created_ms), dtype=np.uint64)

    def retract_many(self, ids):
        live = np.concatenate(self.live)
        live[np.asarray(ids, dtype=np.int64)] = False
        self.live = [live]
        self.emb, self.created, self.keys = [np.concatenate(self.emb)], [np.concatenate(self.created)], [np.concatenate(self.keys)]

    def size(self):
        return int(np.concatenate(self.live).sum())

    def search_shard(self, q, k, since_ms=_hip.SINCE_NONE, tag_mask=0, stats=None):
        emb, created, keys, live = (np.concatenate(x) for x in (self.emb, self.created, self.keys, self.live))
        hexes = [bytes(r).hex() for r in keys]
        idx, sc = orc.compile_arrays(emb, created, (~live).astype(np.uint8), hexes, q, k,
                                     None if since_ms == _hip.SINCE_NONE else since_ms)
        if stats is not None:
            stats.scan_ms = 1e-3
        return idx.astype(np.uint64), sc, keys[idx]


_hip.Index = OracleIndex
_hip.embed_batch = orc.embed_batch
_hip.set_option = lambda *a: None

if __name__ == "__main__":
    bench.main()
