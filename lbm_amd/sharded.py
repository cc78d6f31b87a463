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

from .context_graph import _HEX64, ContextGraph, LazyContext, format_context

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


class ShardPeerError(RuntimeError):
    """Another rank's local search failed; every rank fails that query together instead of waiting for it."""


class ShardedSearcher:
    """Wraps a rank-local search function with the all-gather merge.

    ``local_search(q, k, since_ms, tag_mask) -> (local_ids, scores, keys[n, 32])`` is normally
    ``lbm_amd._hip.Index.search_shard``.  ``gid_of_local`` maps local claim ids to global ones
    (default: ``local_id * world + rank``, i.e. claims dealt round-robin to ranks).

    Per query: pack k entries into pinned memory, one async H2D copy, ONE ``all_gather_into_tensor``
    (RCCL over xGMI under backend "nccl"), one async D2H copy into pinned memory, one stream
    synchronisation, then the host merge of the k*N entries.

    A rank whose local search raises still takes part in the all-gather -- its list's count word is -1 -- and then
    every rank raises: the failing one its own exception, the others ``ShardPeerError``.  Nobody waits for a list
    that will never come.
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

    def _local_lists(self, qs: np.ndarray, k: int, since_ms: int, tag_mask: int, failed: Optional[BaseException]):
        """-> (int64 [nq, k + 1, 6], exception or None).  ``failed``: the caller's own local phase already failed."""
        out = np.zeros((len(qs), k + 1, ENTRY_WORDS), dtype=np.int64)
        try:
            if failed is not None:
                raise failed
            for j, q in enumerate(qs):
                ids, scores, keys = self.local_search(q, k, since_ms, tag_mask)
                out[j] = pack_local(scores, self.gid_of_local(np.asarray(ids)), keys, k)
        except Exception as e:  # noqa: BLE001 -- whatever it is, the peers must hear of it through the exchange
            out[...] = 0
            out[0, 0, 0] = -1
            return out, e
        return out, None

    def _gather(self, local: np.ndarray) -> np.ndarray:
        """local: int64 [rows, 6] -> [world, rows, 6] (rank-major), one all-gather."""
        rows = local.shape[0]
        if self.world == 1 and not self.always_gather:
            return local[None]
        import torch
        import torch.distributed as dist
        b = self._bufs.get(rows)
        if b is None:
            if len(self._bufs) > 32:
                self._bufs.clear()
            b = self._bufs[rows] = _Buffers(rows - 1, self.world, self.device)
        b.send_np[...] = local
        if b.on_gpu:
            b.send_dev.copy_(b.send_host, non_blocking=True)
        dist.all_gather_into_tensor(b.recv_dev, b.send_dev, group=self.group)  # rank-major concatenation
        if b.on_gpu:
            b.recv_host.copy_(b.recv_dev, non_blocking=True)
            torch.cuda.current_stream(b.recv_dev.device).synchronize()
        return b.recv_np

    def search_many(self, qs, k: int, since_ms: int, tag_mask: int = 0, failed: Optional[BaseException] = None):
        """Batch of queries, ONE all-gather of all their lists -> [(global ids, scores, key words)] per query."""
        k = max(0, int(k))
        qs = list(qs)
        if not qs:
            return []
        local, err = self._local_lists(qs, k, since_ms, tag_mask, failed)
        g = self._gather(local.reshape(len(qs) * (k + 1), ENTRY_WORDS)).reshape(self.world, len(qs), k + 1, ENTRY_WORDS)
        bad = [r for r in range(self.world) if g[r, 0, 0, 0] < 0]
        if err is not None:
            raise err
        if bad:
            raise ShardPeerError(f"claim-sharded search: the local search of rank(s) {bad} failed; this query fails on every rank")
        return [self._merge(np.ascontiguousarray(g[:, j]), k) for j in range(len(qs))]

    def search(self, q: np.ndarray, k: int, since_ms: int, tag_mask: int = 0, failed: Optional[BaseException] = None):
        return self.search_many([q], k, since_ms, tag_mask, failed)[0]


def owner_of(claim_hash: str, world: int) -> int:
    """Rank that scans a claim: a stable function of the hash alone (CRC-32 of its UTF-8 bytes), so every rank routes
    every claim the same way without talking to anyone."""
    return zlib.crc32(claim_hash.encode("utf-8", "surrogatepass")) % world


class ShardedContextGraph(ContextGraph):
    """``ContextGraph`` over the GPUs of a node, one process per GPU (SPMD): every rank is handed the SAME sequence of
    ``add_claim`` / ``retract`` calls -- the way every node of the reference replays the same chain -- but KEEPS only the
    claims it owns (``owner_of(hash)``): their ``Claim`` objects in ``claims`` (this rank's shard of the reference's dict,
    lb/context_graph.py:57) and their rows in its GPU's index.  Host and device state per rank are ~ N / world, so a
    100 M-claim corpus on 8 ranks holds 12.5 M ``Claim`` objects per process, not 100 M.  What every rank does track for
    every claim it is shown is O(1): the tag -> bit table (so that the ranks' tag masks mean the same), the kind of hash
    (hex / short free-form) and a global operation counter (the ``claims``-dict position stamp of the recency query).

    ``compile`` is collective: each rank ranks its shard (same kernels as the single-GPU graph), the ranks exchange their
    top-k lists once -- ``lbh_index_search_sharded`` (RCCL all-gather over xGMI inside liblbhip) when a ``comm`` is given,
    else ``torch.distributed`` on ``group`` (gloo in the CPU tests) -- and merge them with the reference's comparator; the
    owners of the k winners then hand their (text, tags) to every rank in a second small exchange, and every rank formats
    and returns exactly what one ``ContextGraph`` holding all claims returns (lb/context_graph.py:91-133).
    ``compile_many`` does the same for a batch (local K3 pass, one exchange of all lists); ``recent`` / ``recent_claims``
    merge the ranks' newest-first lists (lb/node.py:736-762); ``snapshot`` gathers the shards into the reference's form.

    Claim hashes must order like their 32-byte keys on every rank: either all 64-hex CAS hashes or all short strings
    (<= 32 UTF-8 bytes, no NUL).  Anything else raises ``ValueError`` at the next query, on every rank alike (rank keys
    are private to one index and cannot be merged across shards).  Edits made by hand to ``claims`` or to a ``Claim``
    reach only the rank they are made on: SPMD callers go through ``add_claim`` / ``retract``."""

    _overlap_searches = False  # a search here is a collective: every rank must issue them in the same order, one at a time

    def __init__(self, rank: int, world: int, *, comm=None, group=None, _index_factory=None, _embed=None) -> None:
        super().__init__()
        if not 0 <= int(rank) < int(world):
            raise ValueError("rank must be in [0, world)")
        self.rank, self.world = int(rank), int(world)
        self._comm, self._group = comm, group
        self._index_factory, self._embed_fn = _index_factory, _embed
        self._seen_hex = self._seen_raw = False
        self._key_error: Optional[str] = None
        self._ops = 0  # add_claim calls seen so far (owned or not): the same number on every rank
        self._searcher: Optional[ShardedSearcher] = None

    # -- hooks of the base class -------------------------------------------------------------------------------------
    @property
    def _index(self):
        if self._index_obj is None and self._index_factory is not None:
            self._index_obj = self._index_factory()  # CPU tests: an oracle-backed stand-in
        return ContextGraph._index.fget(self)

    def _observe_hash(self, h) -> None:
        """The key mode has to follow ALL hashes of the graph, not just this rank's, or the ranks' keys would not be
        comparable.  Problems are kept for the next query, which raises them on every rank before any collective."""
        if isinstance(h, str) and _HEX64.match(h):
            self._seen_hex = True
        elif self._raw_key(h) is not None:
            self._seen_raw = True
        else:
            self._key_error = f"claim-sharded graph: hash {h!r} is neither 64 hex chars nor <= 32 UTF-8 bytes without NUL"
        if self._seen_hex and self._seen_raw and self._key_error is None:
            self._key_error = ("claim-sharded graph: 64-hex hashes and short free-form hashes cannot be mixed "
                               "(their 32-byte keys would not order like the strings)")
        self._key_mode = "hex" if self._seen_hex else "raw"

    def _owns(self, h: str) -> bool:
        self._observe_hash(h)  # also reached by claims put into ``claims`` by hand
        return isinstance(h, str) and owner_of(h, self.world) == self.rank

    def add_claim(self, claim_hash, text, tags, evidence=None, *, created_ms=None, parent_hash=None) -> None:
        """lb/context_graph.py:60-82, SPMD: every rank is called, the owner stores.  Pass ``created_ms`` (with None the
        owner's clock stamps the claim, as the reference's single process would)."""
        with self._lock:
            self._ops += 1
            self._observe_hash(claim_hash)
            self._tagmask(list(tags))  # tag bits are handed out in first-seen order: the same table on every rank
            if not (isinstance(claim_hash, str) and owner_of(claim_hash, self.world) == self.rank):
                return
            self._next_seq = max(self._next_seq, self._ops)  # a NEW key's dict position, comparable across ranks
            super().add_claim(claim_hash, text, tags, evidence, created_ms=created_ms, parent_hash=parent_hash)

    def retract(self, claim_hash) -> None:
        """lb/context_graph.py:84-89, SPMD: a no-op on the ranks that do not own the claim."""
        with self._lock:
            if isinstance(claim_hash, str) and owner_of(claim_hash, self.world) == self.rank:
                super().retract(claim_hash)

    def _hash_of_key(self, key: np.ndarray) -> str:
        b = bytes(key)
        return b.hex() if self._key_mode == "hex" else b.rstrip(b"\x00").decode("utf-8", "surrogatepass")

    # -- exchanges -----------------------------------------------------------------------------------------------------
    def _allgather_obj(self, obj) -> list:
        """Every rank's (JSON-able) ``obj``, rank-major.  Collective."""
        import json
        if self._comm is not None:
            return [json.loads(b.decode("utf-8", "surrogatepass")) for b in
                    self._comm.allgather(json.dumps(obj).encode("utf-8", "surrogatepass"))]
        if self.world == 1 and self._group is None:
            return [json.loads(json.dumps(obj))]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self._group)
        return out

    def _prepare(self, since_ms, tags):
        """The rank-local phase ahead of a search: reconcile, filters.  -> (since, tag_mask)."""
        if self._key_error is not None:
            raise ValueError(self._key_error)
        self._sync()
        if self._key_error is not None:  # a hash put into ``claims`` by hand
            raise ValueError(self._key_error)
        self._mark_scored(since_ms, tags)
        since, tag_mask, empty = self._filters(since_ms, tags)
        if empty:  # nothing on ANY rank can pass (same tag table everywhere): still take part in the exchange
            since, tag_mask = _hip_int64_max(), 0
        return since, tag_mask

    def _rank_many(self, queries: Sequence[str], top_k, since_ms, tags, stats=None) -> List[Tuple[List[str], List[float]]]:
        """Collective: every rank must call it with the same arguments.  -> [(hashes, fp64 scores)] per query."""
        from . import _hip
        with self._lock:
            k = max(0, int(top_k))
            if k == 0 or not queries:
                return [([], []) for _ in queries]  # on every rank alike: no exchange for an empty request
            embed = self._embed_fn or (lambda t: _hip.embed_batch([t])[0])
            if self._comm is not None:
                # failures of this phase are the same on every rank (bad hashes, tag overflow) or fatal for the device;
                # a rank that dies here is noticed by its peers' bounded wait inside liblbhip (coll_timeout_ms)
                since, tag_mask = self._prepare(since_ms, tags)
                index = self._index
                if len(queries) == 1:
                    _, scores, keys = index.search_sharded(self._comm, embed(queries[0]), k, since, tag_mask, stats, gid_mul=1, gid_add=0)
                    per_query = [(scores, keys)]
                else:
                    qv = np.stack([embed(q) for q in queries]) if self._embed_fn else _hip.embed_batch(list(queries))
                    _, sc, ky, cnt = index.search_sharded_many(self._comm, qv, k, since, tag_mask, stats, gid_mul=1, gid_add=0)
                    per_query = [(sc[j, : int(cnt[j])], ky[j, : int(cnt[j])]) for j in range(len(queries))]
            else:
                # torch.distributed (gloo) has no bounded wait of ours: a rank whose local phase raises must still send
                failed, since, tag_mask, qv = None, 0, 0, [None] * len(queries)
                try:
                    since, tag_mask = self._prepare(since_ms, tags)
                    qv = [embed(q) for q in queries]
                except Exception as e:  # noqa: BLE001
                    failed = e
                if self._searcher is None:
                    self._searcher = ShardedSearcher(self._local_search, self.rank, self.world, group=self._group,
                                                     always_gather=self.world > 1 or self._group is not None)
                res = self._searcher.search_many(qv, k, since, tag_mask, failed=failed)
                per_query = [(scores, np.ascontiguousarray(kw).astype(">u8").view(np.uint8).reshape(-1, 32)) for _, scores, kw in res]
            return [([self._hash_of_key(keys[i]) for i in range(len(scores))], [float(x) for x in scores])
                    for scores, keys in per_query]

    def _rank(self, query, top_k, since_ms, tags, stats=None) -> Tuple[List[str], List[float]]:
        return self._rank_many([query], top_k, since_ms, tags, stats)[0]

    def _local_search(self, q, k, since_ms, tag_mask):
        if not self._id_of:
            return np.zeros(0, np.uint64), np.zeros(0), np.zeros((0, 32), np.uint8)
        return self._index.search_shard(q, k, since_ms, tag_mask)

    def _fetch(self, hashes: Sequence[str]) -> dict:
        """hash -> stored claim dict (``Claim.to_dict``) for claims owned by ANY rank: the owners hand them to everybody in
        one small exchange (k entries).  Collective."""
        with self._lock:
            mine = {h: self.claims[h].to_dict() for h in set(hashes) if h in self.claims}
        out: dict = {}
        for part in self._allgather_obj(mine):
            out.update(part)
        return out

    def _format(self, query: str, chosen: List[str]) -> str:
        got = self._fetch(chosen)
        return format_context(query, [(got[h]["text"], got[h]["tags"]) for h in chosen])

    def _lazy(self, query: str, chosen: List[str]) -> LazyContext:
        got = self._fetch(chosen)  # the exchange cannot be deferred: it is collective; the formatting is
        return LazyContext(query, chosen, [(got[h]["text"], tuple(got[h]["tags"])) for h in chosen])

    def compile_many(self, queries: Sequence[str], *, top_k: int = 8, since_ms=None, tags=None, lazy: bool = False):
        """Collective.  The batch is ranked in ONE local pass per rank (K3 on the matrix cores) and ONE exchange of all
        the lists (``lbh_index_search_sharded_many``); one more small exchange hands out the winners' text and tags."""
        queries = list(queries)
        ranked = [hs for hs, _ in self._rank_many(queries, top_k, since_ms, tags)]
        got = self._fetch([h for hs in ranked for h in hs]) if queries else {}
        pairs = [[(got[h]["text"], tuple(got[h]["tags"])) for h in hs] for hs in ranked]
        if lazy:
            return [LazyContext(q, hs, pr) for q, hs, pr in zip(queries, ranked, pairs)]
        return [(format_context(q, pr), hs) for q, hs, pr in zip(queries, ranked, pairs)]

    # -- recency query (lb/node.py:736-762) ---------------------------------------------------------------------------
    def _recent_entries(self, since_ms, limit) -> List[list]:
        """Global newest-first list of [created_ms, dict-position stamp, hash]: every rank's local list (the device's
        masked top-``limit`` on its shard), gathered and merged by (created desc, stamp asc).  Collective."""
        import operator
        if limit is not None:
            limit = operator.index(limit)
        with self._lock:
            # a non-negative limit bounds what each rank has to send; None / negative limits need every match
            local_limit = limit if (limit is not None and limit >= 0) else None
            hs = ContextGraph.recent(self, since_ms, local_limit)
            mine = [[int(self.claims[h].created_ms), int(self._seq_of[h]), h] for h in hs]
        merged = [e for part in self._allgather_obj(mine) for e in part]
        merged.sort(key=lambda e: (-e[0], e[1]))
        return merged[:limit]

    def recent(self, since_ms, limit: Optional[int] = 100) -> List[str]:
        return [e[2] for e in self._recent_entries(since_ms, limit)]

    def recent_claims(self, since_ms, limit: Optional[int] = 100) -> List[dict]:
        hs = self.recent(since_ms, limit)
        got = self._fetch(hs)
        return [_recent_dict(got[h]) for h in hs]

    # -- snapshot (lb/context_graph.py:135-148) ------------------------------------------------------------------------
    def snapshot(self) -> dict:
        """Collective: the reference's snapshot of the WHOLE graph (every rank's claims, in the order a single graph's
        dict would hold them), identical on every rank."""
        with self._lock:
            mine = [[int(self._seq_of.get(h, 0)), h, c.to_dict()] for h, c in self.claims.items()]
        rows = sorted((e for part in self._allgather_obj(mine) for e in part), key=lambda e: e[0])
        return {"claims": {h: d for _, h, d in rows}}

    @staticmethod
    def from_snapshot_sharded(s: dict, rank: int, world: int, **kw) -> "ShardedContextGraph":
        """``from_snapshot`` for one rank: every rank reads the same snapshot and keeps the claims it owns (re-vectorised
        in one K1 launch at its first query, like the single-GPU loader)."""
        from .context_graph import Claim
        g = ShardedContextGraph(rank, world, **kw)
        for h, cd in (s.get("claims", {}) or {}).items():
            g._ops += 1
            g._observe_hash(h)
            c = Claim.from_dict(cd)
            g._tagmask(list(c.tags))
            if isinstance(h, str) and owner_of(h, world) == rank:
                g._next_seq = max(g._next_seq, g._ops)
                g.claims[h] = c
                if not c.retracted:
                    g._pending_text[h] = c.text
        return g


def _hip_int64_max() -> int:
    from . import _hip
    return _hip.INT64_MAX


def _recent_dict(d: dict) -> dict:
    """The dict literal of lb/node.py:753-760 from a stored claim dict."""
    return {"claim_hash": d["claim_hash"], "text": d["text"], "tags": list(d["tags"]), "created_ms": d["created_ms"],
            "parent_hash": d.get("parent_hash")}
