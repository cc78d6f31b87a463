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


def merge_entries(gathered: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Same result as ``merge_gathered`` through the C ABI's host merge (``lbh_merge_topk``, a partial sort
    of the k*N entries in C instead of five numpy passes)."""
    from . import _hip
    e = _hip.merge_topk(gathered, k)
    return (e[:, 1].copy().view(np.uint64), e[:, 0].copy().view(np.float64),
            np.ascontiguousarray(e[:, 2:6]).view(np.uint64))


class _Buffers:
    """Exchange buffers for one k: pinned staging on both sides of the collective, allocated once."""

    def __init__(self, k: int, world: int, device) -> None:
        import torch
        rows = k + 1
        on_gpu = device is not None and torch.device(device).type == "cuda"
        self.send_host = torch.zeros((rows, ENTRY_WORDS), dtype=torch.int64, pin_memory=on_gpu)
        self.recv_host = torch.zeros((world * rows, ENTRY_WORDS), dtype=torch.int64, pin_memory=on_gpu)
        self.send_np = self.send_host.numpy()
        self.recv_np = self.recv_host.numpy().reshape(world, rows, ENTRY_WORDS)
        if on_gpu:
            self.send_dev = torch.zeros((rows, ENTRY_WORDS), dtype=torch.int64, device=device)
            self.recv_dev = torch.zeros((world * rows, ENTRY_WORDS), dtype=torch.int64, device=device)
        else:  # gloo on CPU tensors (tests)
            self.send_dev, self.recv_dev = self.send_host, self.recv_host
        self.on_gpu = on_gpu


class ShardedSearcher:
    """Wraps a rank-local search function with the all-gather merge.

    ``local_search(q, k, since_ms, tag_mask) -> (local_ids, scores, keys[n, 32])`` is normally
    ``lbm_amd._hip.Index.search_shard``.  ``gid_of_local`` maps local claim ids to global ones
    (default: ``local_id * world + rank``, i.e. claims dealt round-robin to ranks).

    Per query: pack k entries into pinned memory, one async H2D copy, ONE ``all_gather_into_tensor``
    (RCCL over xGMI under backend "nccl"), one async D2H copy into pinned memory, one stream
    synchronisation, then the host merge of the k*N entries.
    """

    def __init__(self, local_search: Callable, rank: int, world: int, *, device=None, group=None,
                 gid_of_local: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                 always_gather: bool = False, c_merge: bool = True) -> None:
        self.local_search, self.rank, self.world = local_search, rank, world
        self.device, self.group = device, group
        self.always_gather = always_gather  # run the collective even for world == 1 (tests)
        self.gid_of_local = gid_of_local or (lambda ids: ids.astype(np.uint64) * np.uint64(world) + np.uint64(rank))
        self._merge = merge_entries if c_merge else merge_gathered
        self._bufs = {}

    def search(self, q: np.ndarray, k: int, since_ms: int, tag_mask: int = 0):
        k = max(0, int(k))
        ids, scores, keys = self.local_search(q, k, since_ms, tag_mask)
        local = pack_local(scores, self.gid_of_local(np.asarray(ids)), keys, k)
        if self.world == 1 and not self.always_gather:
            return self._merge(local[None], k)
        import torch
        import torch.distributed as dist
        b = self._bufs.get(k)
        if b is None:
            b = self._bufs[k] = _Buffers(k, self.world, self.device)
        b.send_np[...] = local
        if b.on_gpu:
            b.send_dev.copy_(b.send_host, non_blocking=True)
        dist.all_gather_into_tensor(b.recv_dev, b.send_dev, group=self.group)  # rank-major concatenation
        if b.on_gpu:
            b.recv_host.copy_(b.recv_dev, non_blocking=True)
            torch.cuda.current_stream(b.recv_dev.device).synchronize()
        return self._merge(b.recv_np, k)
