"""Claim-sharded retrieval over N GPUs: one process per GPU, exact global top-k.

The reference ranks by a total order on (-score, claim_hash) with unique hashes
(lb/context_graph.py:120) and a claim's score depends only on that claim and the
query, so the global top-k is contained in the union of the per-shard top-k
lists.  Each rank runs the normal single-GPU search on its shard
(``lbh_index_search_shard``: exact fp64 scores + 32-byte hash keys), the ranks
exchange their k entries with ONE all-gather (48 B per entry; RCCL over xGMI when
the backend is "nccl", gloo in the CPU tests) and every rank merges the k*N
entries with the same comparator.  No other collective is on the data path.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

ENTRY_WORDS = 6  # score bits, global id, 4 key words  (48 bytes)


def pack_local(scores: np.ndarray, gids: np.ndarray, keys: np.ndarray, k: int) -> np.ndarray:
    """-> int64 [k + 1, 6]; row 0 carries the count, unused rows are zero."""
    n = int(scores.shape[0])
    buf = np.zeros((k + 1, ENTRY_WORDS), dtype=np.int64)
    buf[0, 0] = n
    if n:
        buf[1: n + 1, 0] = np.ascontiguousarray(scores, dtype=np.float64).view(np.int64)
        buf[1: n + 1, 1] = np.ascontiguousarray(gids, dtype=np.uint64).view(np.int64)
        kw = np.ascontiguousarray(keys, dtype=np.uint8).reshape(n, 32).view(">u8").astype(np.uint64)
        buf[1: n + 1, 2:6] = kw.view(np.int64)
    return buf


def merge_gathered(gathered: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """gathered: int64 [world, k + 1, 6] -> (global ids, scores, key words) of the global top-k,
    ordered like lb/context_graph.py:120 (score descending, then hash ascending)."""
    rows = []
    for r in range(gathered.shape[0]):
        n = int(gathered[r, 0, 0])
        if n:
            rows.append(gathered[r, 1: n + 1])
    if not rows:
        z = np.zeros(0)
        return z.astype(np.uint64), z.astype(np.float64), np.zeros((0, 4), dtype=np.uint64)
    e = np.concatenate(rows, axis=0)
    scores = e[:, 0].copy().view(np.float64)
    kw = e[:, 2:6].copy().view(np.uint64)
    # np.lexsort: last key is the primary one.  -score ascending == score descending; -0.0 == 0.0 tie like Python.
    order = np.lexsort((kw[:, 3], kw[:, 2], kw[:, 1], kw[:, 0], -scores))[:k]
    return e[order, 1].copy().view(np.uint64), scores[order], kw[order]


class ShardedSearcher:
    """Wraps a rank-local search function with the all-gather merge.

    ``local_search(q, k, since_ms, tag_mask) -> (local_ids, scores, keys[n, 32])`` is normally
    ``lbm_amd._hip.Index.search_shard``.  ``gid_of_local`` maps local claim ids to global ones
    (default: ``local_id * world + rank``, i.e. claims dealt round-robin to ranks).
    """

    def __init__(self, local_search: Callable, rank: int, world: int, *, device=None, group=None,
                 gid_of_local: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                 always_gather: bool = False) -> None:
        self.local_search, self.rank, self.world = local_search, rank, world
        self.device, self.group = device, group
        self.always_gather = always_gather  # run the collective even for world == 1 (tests)
        self.gid_of_local = gid_of_local or (lambda ids: ids.astype(np.uint64) * np.uint64(world) + np.uint64(rank))

    def search(self, q: np.ndarray, k: int, since_ms: int, tag_mask: int = 0):
        k = max(0, int(k))
        ids, scores, keys = self.local_search(q, k, since_ms, tag_mask)
        local = pack_local(scores, self.gid_of_local(np.asarray(ids)), keys, k)
        if self.world == 1 and not self.always_gather:
            return merge_gathered(local[None], k)
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(local)
        if self.device is not None:
            t = t.to(self.device, non_blocking=True)
        out = torch.empty((self.world * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)  # rank-major concatenation
        return merge_gathered(out.cpu().numpy().reshape(self.world, t.shape[0], t.shape[1]), k)
