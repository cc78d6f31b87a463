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

import zlib
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .context_graph import _HEX64, ContextGraph

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


def owner_of(claim_hash: str, world: int) -> int:
    """Rank that scans a claim: a stable function of the hash alone (CRC-32 of its UTF-8 bytes), so every rank routes
    every claim the same way without talking to anyone."""
    return zlib.crc32(claim_hash.encode("utf-8", "surrogatepass")) % world


class ShardedContextGraph(ContextGraph):
    """``ContextGraph`` over the GPUs of a node, one process per GPU (SPMD): every rank applies the SAME sequence of
    ``add_claim`` / ``retract`` / dict edits -- the way every node of the reference replays the same chain -- and keeps the
    host-side ``claims`` dict in full, but only the claims it owns (``owner_of(hash)``) get a row in its GPU's index.
    ``compile`` is collective: each rank ranks its shard (same kernels as the single-GPU graph), the ranks exchange their
    top-k lists once -- ``lbh_index_search_sharded`` (RCCL all-gather over xGMI inside liblbhip) when a ``comm`` is given,
    else ``torch.distributed`` on ``group`` (gloo in the CPU tests) -- and merge them with the reference's comparator, so
    every rank returns exactly what one ``ContextGraph`` holding all claims returns (lb/context_graph.py:91-133).

    Claim hashes must order like their 32-byte keys on every rank: either all 64-hex CAS hashes or all short strings
    (<= 32 UTF-8 bytes, no NUL).  Anything else raises ``ValueError`` at the next query (rank keys are private to one
    index and cannot be merged across shards)."""

    _overlap_searches = False  # a search here is a collective: every rank must issue them in the same order, one at a time

    def __init__(self, rank: int, world: int, *, comm=None, group=None, _index_factory=None, _embed=None) -> None:
        super().__init__()
        if not 0 <= int(rank) < int(world):
            raise ValueError("rank must be in [0, world)")
        self.rank, self.world = int(rank), int(world)
        self._comm, self._group = comm, group
        self._index_factory, self._embed_fn = _index_factory, _embed
        self._seen_hex = self._seen_raw = False
        self._searcher: Optional[ShardedSearcher] = None

    # -- hooks of the base class -------------------------------------------------------------------------------------
    @property
    def _index(self):
        if self._index_obj is None and self._index_factory is not None:
            self._index_obj = self._index_factory()  # CPU tests: an oracle-backed stand-in
        return ContextGraph._index.fget(self)

    def _owns(self, h: str) -> bool:
        # called for every live claim of a sync, on every rank, before ownership decides anything: the key mode has to
        # follow ALL hashes of the graph, not just this rank's, or the ranks' keys would not be comparable
        if isinstance(h, str) and _HEX64.match(h):
            self._seen_hex = True
        elif self._raw_key(h) is not None:
            self._seen_raw = True
        else:
            raise ValueError(f"claim-sharded graph: hash {h!r} is neither 64 hex chars nor <= 32 UTF-8 bytes without NUL")
        if self._seen_hex and self._seen_raw:
            raise ValueError("claim-sharded graph: 64-hex hashes and short free-form hashes cannot be mixed "
                             "(their 32-byte keys would not order like the strings)")
        self._key_mode = "hex" if self._seen_hex else "raw"
        return owner_of(h, self.world) == self.rank

    def _hash_of_key(self, key: np.ndarray) -> str:
        b = bytes(key)
        return b.hex() if self._key_mode == "hex" else b.rstrip(b"\x00").decode("utf-8", "surrogatepass")

    def _rank(self, query, top_k, since_ms, tags, stats=None) -> Tuple[List[str], List[float]]:
        """Collective: every rank must call it with the same arguments."""
        from . import _hip
        with self._lock:
            self._sync()
            self._mark_scored(since_ms, tags)
            k = max(0, int(top_k))
            since, tag_mask, empty = self._filters(since_ms, tags)
            index = self._index
            if k == 0:
                return [], []  # on every rank alike: no exchange for an empty request
            if empty:  # nothing on THIS rank can pass (no claim of this shard carries the tag): still take part in the exchange
                since, tag_mask = _hip.INT64_MAX, 0
            qv = (self._embed_fn or (lambda t: _hip.embed_batch([t])[0]))(query)
            if self._comm is not None:
                _, scores, keys = index.search_sharded(self._comm, qv, k, since, tag_mask, stats, gid_mul=1, gid_add=0)
            else:
                if self._searcher is None:
                    self._searcher = ShardedSearcher(self._local_search, self.rank, self.world, group=self._group, always_gather=True)
                _, scores, kw = self._searcher.search(qv, k, since, tag_mask)
                keys = np.ascontiguousarray(kw).astype(">u8").view(np.uint8).reshape(-1, 32)
            return [self._hash_of_key(keys[i]) for i in range(len(scores))], [float(x) for x in scores]

    def _local_search(self, q, k, since_ms, tag_mask):
        if not self._id_of:
            return np.zeros(0, np.uint64), np.zeros(0), np.zeros((0, 32), np.uint8)
        return self._index.search_shard(q, k, since_ms, tag_mask)

    def compile_many(self, queries: Sequence[str], *, top_k: int = 8, since_ms=None, tags=None, lazy: bool = False):
        """Collective; one exchange per query (the batched MFMA path is per GPU and has no sharded form yet)."""
        return [self.compile(q, top_k=top_k, since_ms=since_ms, tags=tags, lazy=lazy) for q in queries]

    def recent(self, since_ms, limit=100):
        raise NotImplementedError("get_recent_claims is not sharded: ask a single-GPU ContextGraph")
