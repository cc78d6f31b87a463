"""Drop-in for the reference's ``lb/context_graph.py`` with the ranking on the GPU.

Public surface is the reference's (lb/context_graph.py:10-148): the ``Claim``
dataclass (+ ``to_dict`` / ``from_dict``), ``ContextGraph`` with a live
``claims`` dict, ``_embedding_cache``, ``add_claim``, ``retract``, ``compile``,
``snapshot`` / ``from_snapshot``.  ``compile`` gains one opt-in keyword,
``tags=None`` (any-of tag filter, fused into the scan; absent from the
reference, whose behaviour is the default).

How the mirror works
  * ``claims`` stays a real ``dict`` (subclass) because callers read and write
    it directly (lb/admin.py:396-397, tests/test_multi_agent.py:160-162).  Every
    mutation of the dict or of a ``Claim`` attribute marks that hash dirty.
  * The device index (liblbhip.so, include/lbhip.h) holds one row per claim the
    reference would score: present in ``claims``, not retracted, embedding
    cached.  ``compile`` first reconciles dirty hashes -- all new texts go to the
    GPU in ONE batched embed call (K1) -- then runs the search kernels and
    formats the text on the host (lb/context_graph.py:122-133).
  * Cache semantics are the reference's: the vector is computed when the claim
    is added (``add_claim``) or first scored; later edits of ``Claim.text`` do
    not re-embed; ``retract`` drops the vector (:89); flipping ``retracted`` or
    deleting a dict entry by hand keeps it ("parked" on the host as fp64 bits).
  * Tie-break keys (32 bytes per claim, compared as big-endian integers; they must
    order like Python ``str``, which lb/context_graph.py:120 sorts by).  Three modes:
    "hex"  every hash is 64 lowercase hex chars (what the CAS produces): the 32 raw
           bytes (byte order == str order);
    "raw"  every hash is at most 32 UTF-8 bytes without NUL (the reference's tests
           use "old", "new", "h1"): the UTF-8 bytes, zero-padded (UTF-8 byte order ==
           code-point order, and without NUL the padding cannot tie two hashes);
    "rank" anything else, or a mix: hashes are ranked with Python's own ``str``
           order and the ranks become the keys: 256-bit integers, a new hash takes the
           midpoint of its neighbours' keys (O(log n) per publish); every row is relabelled
           only when a gap runs out of midpoints or the mode changes (INTEGRATION.md Limits).
"""
from __future__ import annotations

import bisect
import functools
import operator
import re
import threading
import time
from dataclasses import dataclass
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _hip

def _locked(fn):
    """Run a ContextGraph method under the graph's re-entrant lock.  The reference's compile_context takes no
    lock while writers hold BatteryNode._lock (lb/node.py:718 vs :465), so queries and publishes do overlap;
    here both touch the device mirror's bookkeeping, which must not be seen half-updated."""
    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        with self._lock:
            return fn(self, *a, **kw)
    return wrapper


def _locked_g(fn):
    """Same, for the helper objects that hold their graph in ``_g`` (claims dict, cache view)."""
    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        with self._g._lock:
            return fn(self, *a, **kw)
    return wrapper


_RK_SPACE = 1 << 256  # rank keys are 32-byte integers in (0, 2^256)


class _BisectList(list):
    """The three SortedList operations the rank keys need, on a plain list (inserts move O(N) pointers)."""

    def add(self, v) -> None:
        bisect.insort_left(self, v)

    def bisect_left(self, v) -> int:
        return bisect.bisect_left(self, v)

    def remove(self, v) -> None:
        i = bisect.bisect_left(self, v)
        if i < len(self) and self[i] == v:
            del self[i]


def _sorted_list(items=()):
    try:
        from sortedcontainers import SortedList  # O(log N) inserts; optional
        return SortedList(items)
    except ImportError:
        return _BisectList(sorted(items))


_HEX64 = re.compile(r"[0-9a-f]{64}\Z")
_HEXRUN = re.compile(r"[0-9a-f]*\Z")
MAX_TAG_BITS = 64


class _TagList(list):
    """``Claim.tags`` of a claim that lives in a graph: a list that reports in-place edits (``tags.append(..)``,
    ``tags[:] = ..``) to the graph, so the device-side tag mask of the opt-in ``tags=`` filter cannot go stale."""

    __slots__ = ("_lbm_claim",)

    def _ping(self) -> None:
        c = getattr(self, "_lbm_claim", None)
        if c is not None:
            owner = c.__dict__.get("_lbm_owner")
            if owner is not None and c.__dict__.get("tags") is self:
                owner[0]._touch(owner[1])

    def _wrap(name):  # noqa: N805
        base = getattr(list, name)

        def method(self, *a, **kw):
            r = base(self, *a, **kw)
            self._ping()
            return r
        method.__name__ = name
        return method

    for _n in ("append", "extend", "insert", "remove", "pop", "clear", "sort", "reverse", "__setitem__", "__delitem__",
               "__iadd__", "__imul__"):
        locals()[_n] = _wrap(_n)
    del _n, _wrap


@dataclass
class Claim:  # lb/context_graph.py:10-43
    claim_hash: str
    text: str
    tags: List[str]
    created_ms: int
    evidence: List[str]
    parent_hash: Optional[str] = None  # the claim this one answers, if any (carried, never ranked on)
    retracted: bool = False

    def __setattr__(self, name: str, value: Any) -> None:
        if name == "tags" and type(value) is list:
            value = _TagList(value)
            value._lbm_claim = self
        object.__setattr__(self, name, value)
        owner = self.__dict__.get("_lbm_owner")
        if owner is not None and name in ("retracted", "created_ms", "tags", "text"):
            owner[0]._touch(owner[1])

    def to_dict(self) -> Dict[str, Any]:
        """The stored form (lb/context_graph.py:20-31): the fields of ``_STORED`` in that order, ``parent_hash`` only when set."""
        out: Dict[str, Any] = {"claim_hash": self.claim_hash, "text": self.text}
        for name, conv, _ in _STORED[2:]:
            out[name] = conv(getattr(self, name))
        if self.parent_hash is not None:
            out["parent_hash"] = self.parent_hash
        return out

    @staticmethod
    def from_dict(d: Dict[str, Any]) -> "Claim":
        """lb/context_graph.py:33-43: ``claim_hash`` is required, every other field falls back to its default."""
        rest = {name: conv(d.get(name, default)) for name, conv, default in _STORED[1:]}
        return Claim(claim_hash=str(d["claim_hash"]), parent_hash=d.get("parent_hash"), **rest)


# stored fields of a claim: (name, coercion applied when reading / writing the stored form, default when a stored dict lacks it)
_STORED = (("claim_hash", str, None), ("text", str, ""), ("tags", list, ()), ("created_ms", int, 0), ("evidence", list, ()),
           ("retracted", bool, False))


def format_context(query: str, chosen: Sequence[Tuple[str, Sequence[str]]]) -> str:
    """The context text of lb/context_graph.py:122-133 from (text, tags) pairs: a two-line header, a blank line, one
    numbered line per claim (tags appended when their joined form is non-empty), surrounding whitespace stripped, one
    final newline."""
    def entry(i: int, text: str, tags: Sequence[str]) -> str:
        label = ", ".join(tags) if tags else ""
        return f"{i}. {text}  [tags: {label}]" if label else f"{i}. {text}"

    body = "\n".join(entry(i, text, tags) for i, (text, tags) in enumerate(chosen, 1))
    return f"Context slice\nQuery: {query}\n\n{body}".strip() + "\n"


class LazyContext:
    """Result of a ``lazy=True`` compile: the ranked claim hashes are ready, the context text (lb/context_graph.py:122-133)
    is put together on first use -- off the query's critical path, in whatever thread asks for it.  The chosen claims'
    text and tags are captured when the query ran, so the text is what an eager call would have returned then.
    Unpacks like the reference's result: ``text, hashes = result``."""

    __slots__ = ("query", "claim_hashes", "_chosen", "_text")

    def __init__(self, query: str, hashes: List[str], chosen: List[Tuple[str, Tuple[str, ...]]]) -> None:
        self.query, self.claim_hashes, self._chosen, self._text = query, hashes, chosen, None

    @property
    def context(self) -> str:
        if self._text is None:
            self._text = format_context(self.query, self._chosen)
        return self._text

    def __iter__(self):
        yield self.context
        yield self.claim_hashes

    def __len__(self) -> int:
        return 2

    def __getitem__(self, i):
        return (self.context, self.claim_hashes)[i]

    def __eq__(self, other) -> bool:
        return tuple(self) == (tuple(other) if isinstance(other, (tuple, list, LazyContext)) else other)

    def __repr__(self) -> str:
        return f"LazyContext(query={self.query!r}, claim_hashes={self.claim_hashes!r})"


class _ClaimDict(dict):
    """``ContextGraph.claims``: a dict that reports which hashes changed."""

    def __init__(self, graph: "ContextGraph") -> None:
        super().__init__()
        self._g = graph

    def _adopt(self, k, v) -> None:
        if isinstance(v, Claim):
            v.__dict__["_lbm_owner"] = (self._g, k)
        self._g._touch(k)

    def _forget(self, k) -> None:
        self._g._seq_of.pop(k, None)  # a re-inserted key goes to the END of the dict order
        self._g._touch(k)

    @_locked_g
    def __setitem__(self, k, v) -> None:
        if k not in self:  # assigning to an existing key keeps its position in the dict order
            g = self._g
            g._seq_of[k] = g._next_seq
            g._next_seq += 1
            g._order_pending[k] = None  # a popped and re-inserted claim may still own its device row
        super().__setitem__(k, v)
        self._adopt(k, v)

    @_locked_g
    def __delitem__(self, k) -> None:
        super().__delitem__(k)
        self._forget(k)

    @_locked_g
    def pop(self, k, *default):
        had = k in self
        r = super().pop(k, *default)
        if had:
            self._forget(k)
        return r

    @_locked_g
    def popitem(self):
        k, v = super().popitem()
        self._forget(k)
        return k, v

    @_locked_g
    def clear(self) -> None:
        for k in list(self.keys()):
            self._forget(k)
        super().clear()

    @_locked_g
    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    @_locked_g
    def update(self, *a, **kw) -> None:
        for k, v in dict(*a, **kw).items():
            self[k] = v

    @_locked_g
    def __ior__(self, other):
        self.update(other)
        return self


class _EmbeddingCacheView:
    """``ContextGraph._embedding_cache``: claim_hash -> embedding (list of 64 floats), read from HBM."""

    def __init__(self, graph: "ContextGraph") -> None:
        self._g = graph

    @_locked_g
    def __contains__(self, h) -> bool:
        g = self._g
        g._sync()
        return (h in g._id_of and h not in g._unscored) or h in g._parked or h in g._pending_text

    @_locked_g
    def __getitem__(self, h) -> List[float]:
        g = self._g
        g._sync()
        if h in g._id_of and h not in g._unscored:
            return g._index.get_embedding(g._id_of[h]).tolist()
        if h in g._pending_text:  # add_claim'ed, then taken out of the ranking by hand before any query: embed it now
            g._parked[h] = _hip.embed_batch([g._pending_text.pop(h)])[0].copy()
        if h in g._parked:
            return g._parked[h].tolist()
        raise KeyError(h)

    @_locked_g
    def get(self, h, default=None):
        try:
            return self[h]
        except KeyError:
            return default

    @_locked_g
    def __setitem__(self, h, vec) -> None:
        g = self._g
        g._sync()
        v = np.ascontiguousarray(vec, dtype=np.float64)
        if v.shape != (_hip.DIM,):
            raise ValueError("embedding must have 64 components")
        # what lbh_index_add would refuse is refused here, at the assignment, not at some later sync
        with np.errstate(over="ignore"):
            too_big = not float(np.sqrt(np.sum(v * v))) <= 1e15
        if not np.all(np.isfinite(v)) or too_big:
            raise ValueError("embedding must be finite with a norm of at most 1e15")
        g._drop_row(h, park=False)
        g._pending_text.pop(h, None)
        g._parked[h] = v
        g._touch(h)

    @_locked_g
    def pop(self, h, *default):
        g = self._g
        g._sync()
        if (h in g._id_of and h not in g._unscored) or h in g._parked or h in g._pending_text:
            val = self[h]
            g._drop_row(h, park=False)
            g._parked.pop(h, None)
            g._touch(h)
            return val
        if default:
            return default[0]
        raise KeyError(h)

    @_locked_g
    def __delitem__(self, h) -> None:
        self.pop(h)

    @_locked_g
    def keys(self) -> List[str]:
        g = self._g
        g._sync()
        return ([h for h in g._id_of if h not in g._unscored] + [h for h in g._parked if h not in g._id_of]
                + [h for h in g._pending_text if h not in g._id_of and h not in g._parked])

    @_locked_g
    def __iter__(self) -> Iterator[str]:
        return iter(self.keys())

    @_locked_g
    def __len__(self) -> int:
        return len(self.keys())

    @_locked_g
    def items(self):
        return [(h, self[h]) for h in self.keys()]

    @_locked_g
    def clear(self) -> None:
        for h in self.keys():
            self.pop(h, None)


class ContextGraph:
    """Drop-in for the reference's claim graph (lb/context_graph.py:46-148): the ``claims`` dict stays on the host, the
    vectors and everything ``compile`` ranks on live in HBM, mirrored incrementally."""

    # compile() releases the graph lock while its search runs on the device (the index's own reader/writer lock and
    # per-search slots take over), so queries from several threads overlap like the reference's lock-free
    # compile_context (lb/node.py:718-734).  SPMD subclasses, whose searches are collectives, turn this off.
    _overlap_searches = True

    def __init__(self) -> None:
        self._lock = threading.RLock()
        self._cond = threading.Condition(self._lock)
        self._readers = 0  # searches in flight outside the lock; the id maps and device rows stay put until they are back
        self._dirty: Dict[str, None] = {}      # insertion-ordered set of hashes to reconcile
        self._pending_text: Dict[str, str] = {}  # text captured by add_claim (embedded at next sync)
        self._id_of: Dict[str, int] = {}       # hash -> device claim id (rows the GPU scans)
        self._hash_of: Dict[int, str] = {}
        self._meta: Dict[str, Tuple[int, int]] = {}  # hash -> (created_ms, tagmask) as stored on device
        self._parked: Dict[str, np.ndarray] = {}     # cached vectors of claims currently not scanned
        self._discard: Dict[str, None] = {}          # hashes whose cached vector retract() / add_claim() dropped since the last sync
        # Claims that reached the device without passing through add_claim / from_snapshot (inserted into ``claims`` by
        # hand, or made live again after retract()): the reference has no cache entry for them until a compile() scores
        # them (:113-117).  hash -> the text their row was embedded from.
        self._unscored: Dict[str, str] = {}
        self._tag_bits: Dict[str, int] = {}
        self._tag_overflow = False
        self._key_mode = "hex"   # "hex" | "raw" | "rank": how claim hashes become 32-byte ordering keys (module docstring)
        self._keys_stale = False
        # rank-key mode: the hashes that own a device row, in Python's str order, and each one's 256-bit key.  A new hash
        # takes the midpoint of its neighbours' keys (O(log N) per publish); only a gap that has run out of midpoints --
        # ~250 inserts at the same spot -- or a change of mode relabels every row.
        self._rk_order = _sorted_list()
        self._rk_key: Dict[str, int] = {}
        self._seq_of: Dict[str, int] = {}      # hash -> position stamp in the ``claims`` dict order
        self._next_seq = 0
        self._order_pending: Dict[str, None] = {}  # rows whose stamp is not on the device yet (recency query only)
        self._index_obj: Optional[_hip.Index] = None
        self.claims: Dict[str, Claim] = _ClaimDict(self)
        self._embedding_cache = _EmbeddingCacheView(self)

    # ------------------------------------------------------------------ plumbing

    @property
    def _index(self) -> _hip.Index:
        if self._index_obj is None:
            self._index_obj = _hip.Index()  # raises HipBackendError without liblbhip.so + MI355X
        return self._index_obj

    @_locked
    def _touch(self, h: str) -> None:
        self._dirty[h] = None

    def _owns(self, h: str) -> bool:
        """Whether this graph's device index scans claim ``h`` (always, except in a claim-sharded graph)."""
        return True

    def _tagmask(self, tags: Iterable[str]) -> int:
        m = 0
        for t in tags:
            b = self._tag_bits.get(t)
            if b is None:
                if len(self._tag_bits) >= MAX_TAG_BITS:
                    self._tag_overflow = True
                    continue
                b = len(self._tag_bits)
                self._tag_bits[t] = b
            m |= 1 << b
        return m

    @staticmethod
    def _raw_key(h) -> Optional[bytes]:
        """Zero-padded UTF-8 image of a short hash, or None if it does not order like the string (too long, NUL inside)."""
        if not isinstance(h, str):
            return None
        b = h.encode("utf-8", "surrogatepass")
        if len(b) > _hip.KEY_BYTES or b"\x00" in b:
            return None
        return b.ljust(_hip.KEY_BYTES, b"\x00")

    def _key_bytes(self, h: str) -> bytes:
        if self._key_mode == "hex":
            if isinstance(h, str) and _HEX64.match(h):
                return bytes.fromhex(h)
            # first hash that is not a CAS hash: an empty device index can still switch to raw keys
            self._key_mode = "raw" if not self._id_of and self._raw_key(h) is not None else "rank"
            if self._key_mode == "rank" and self._id_of:
                self._keys_stale = True  # the rows so far carry hex keys
        if self._key_mode == "raw":
            k = self._raw_key(h)
            if k is not None:
                return k
            self._key_mode = "rank"
            self._keys_stale = True  # every row so far carries a key of the old kind
        return self._rank_key_for(h)

    def _forget_rank_key(self, h: str) -> None:
        if self._rk_key.pop(h, None) is not None:
            self._rk_order.remove(h)

    def _rank_key_for(self, h: str) -> bytes:
        """Rank mode: the key of ``h`` -- its stored one, or a new one between its neighbours in str order."""
        k = self._rk_key.get(h)
        if k is None:
            if self._keys_stale:  # a relabel of every row is pending anyway: placeholder
                return b"\x00" * _hip.KEY_BYTES
            i = self._rk_order.bisect_left(h)
            lo = self._rk_key[self._rk_order[i - 1]] if i > 0 else 0
            hi = self._rk_key[self._rk_order[i]] if i < len(self._rk_order) else _RK_SPACE
            k = (lo + hi) // 2
            if k == lo:  # no room left between the neighbours
                self._keys_stale = True
                return b"\x00" * _hip.KEY_BYTES
            self._rk_order.add(h)
            self._rk_key[h] = k
        return k.to_bytes(_hip.KEY_BYTES, "big")

    @property
    def _rank_keys(self) -> bool:
        return self._key_mode == "rank"

    @_locked
    def _refresh_rank_keys(self) -> None:
        """Relabel every row: keys spread evenly over the 256-bit space in Python's str order (what lb/context_graph.py:120
        compares).  Only on entering rank mode or when a gap between two neighbours is used up."""
        hs = sorted(self._id_of.keys())
        self._rk_order = _sorted_list(hs)
        self._rk_key = {}
        if hs:
            step = _RK_SPACE // (len(hs) + 1)
            ids = np.array([self._id_of[h] for h in hs], dtype=np.uint64)
            blob = bytearray()
            for i, h in enumerate(hs):
                k = (i + 1) * step
                self._rk_key[h] = k
                blob += k.to_bytes(_hip.KEY_BYTES, "big")
            self._index.set_keys(ids, np.frombuffer(bytes(blob), dtype=np.uint8).reshape(len(hs), _hip.KEY_BYTES))
        self._keys_stale = False

    @_locked
    def _drop_row(self, h: str, park: bool) -> None:
        while self._readers:  # called under the lock; see _sync
            self._cond.wait()
        cid = self._id_of.pop(h, None)
        if cid is None:
            return
        if park and h not in self._unscored:
            self._parked[h] = self._index.get_embedding(cid)
        self._index.retract(cid)
        self._hash_of.pop(cid, None)
        self._meta.pop(h, None)
        self._unscored.pop(h, None)
        self._forget_rank_key(h)

    @_locked
    def _sync(self) -> None:
        """Bring the device index in line with ``claims`` (only dirty hashes are looked at).

        Device calls are batched: every row that leaves the scanned arrays goes in ONE ``retract_many`` (one compaction
        kernel), every new text in ONE K1 launch.  Bookkeeping is committed only after the device call it describes has
        succeeded; if a call fails, every hash of the batch is dirty again and parked vectors / pending texts are still
        in place, so a later ``compile`` retries instead of silently missing claims."""
        if not self._dirty and not self._keys_stale:
            return
        while self._readers:  # searches running outside the lock map their ids through _hash_of when they return
            self._cond.wait()
        if not self._dirty:
            if self._keys_stale:
                self._refresh_rank_keys()
            return
        dirty = list(self._dirty.keys())
        self._dirty = {}
        try:
            self._reconcile(dirty)
        except BaseException:
            for h in dirty:
                self._dirty.setdefault(h, None)
            raise

    def _reconcile(self, dirty: List[str]) -> None:
        claims = self.claims
        # stage 1: rows that leave the device.  A hash in _discard lost its vector (retract / add_claim of an existing
        # hash, :82 / :89); a claim that is merely not live right now (flag flipped or dict entry removed by hand) keeps
        # its vector parked on the host.
        drop_park: List[str] = []
        drop_free: List[str] = []
        for h in dirty:
            if h not in self._id_of:
                continue
            if h in self._discard:
                drop_free.append(h)
                continue
            c = claims.get(h)
            if c is None or getattr(c, "retracted", False):
                # a claim the reference never scored has no cached vector to keep
                (drop_free if h in self._unscored else drop_park).append(h)
            elif h in self._unscored and getattr(c, "text", None) != self._unscored[h]:
                drop_free.append(h)  # still uncached in the reference: it will embed the text as it is when first scored
        if drop_park or drop_free:
            index = self._index
            vecs = index.get_embeddings([self._id_of[h] for h in drop_park]) if drop_park else None
            index.retract_many([self._id_of[h] for h in drop_park + drop_free])
            for i, h in enumerate(drop_park):
                self._parked[h] = vecs[i].copy()
            for h in drop_park + drop_free:
                self._hash_of.pop(self._id_of.pop(h), None)
                self._meta.pop(h, None)
                self._unscored.pop(h, None)
                self._forget_rank_key(h)
        for h in dirty:
            if self._discard.pop(h, 0) is None:
                self._parked.pop(h, None)
        # stage 2: metadata edits of rows that stay, and the claims that (re-)enter
        new_text: List[Tuple[str, str, Claim]] = []
        new_vec: List[Tuple[str, np.ndarray, Claim]] = []
        for h in dirty:
            c = claims.get(h)
            if c is None or getattr(c, "retracted", False):
                continue  # a text captured by add_claim stays pending: the reference's cache still holds its vector
            if h in self._id_of:
                cm, tm = int(c.created_ms), self._tagmask(c.tags)
                if self._meta.get(h) != (cm, tm):
                    self._index.update_meta(self._id_of[h], cm, tm, self._key_bytes(h))
                    self._meta[h] = (cm, tm)
                continue
            if not self._owns(h):  # claim-sharded graph: another rank scans this claim
                self._pending_text.pop(h, None)
                continue
            if h in self._parked:
                new_vec.append((h, self._parked[h], c))
            else:
                new_text.append((h, self._pending_text.get(h, c.text), c))
        if new_vec:
            hs = [h for h, _, _ in new_vec]
            created = np.array([int(c.created_ms) for _, _, c in new_vec], dtype=np.int64)
            tmask = np.array([self._tagmask(c.tags) for _, _, c in new_vec], dtype=np.uint64)
            keys = self._keys_of(hs)
            ids = self._index.add_vectors(np.stack([v for _, v, _ in new_vec]), created, tmask, keys)
            self._register(hs, ids, created, tmask)
            for h in hs:
                self._parked.pop(h, None)
        if new_text:
            hs = [h for h, _, _ in new_text]
            blob, offsets = _hip.pack_texts([t for _, t, _ in new_text])
            created = np.array([int(c.created_ms) for _, _, c in new_text], dtype=np.int64)
            tmask = np.array([self._tagmask(c.tags) for _, _, c in new_text], dtype=np.uint64)
            keys = self._keys_of(hs)
            ids = self._index.add_texts(blob, offsets, created, tmask, keys)
            self._register(hs, ids, created, tmask)
            for h, t, _ in new_text:
                if self._pending_text.pop(h, None) is None:
                    self._unscored[h] = t
        if self._rank_keys and self._keys_stale:
            self._refresh_rank_keys()

    def _register(self, hs: Sequence[str], ids: np.ndarray, created: np.ndarray, tmask: np.ndarray) -> None:
        idl = ids.tolist()
        self._order_pending.update(zip(hs, [None] * len(idl)))
        self._id_of.update(zip(hs, idl))
        self._hash_of.update(zip(idl, hs))
        self._meta.update(zip(hs, zip(created.tolist(), tmask.tolist())))

    def _keys_of(self, hs: Sequence[str]) -> np.ndarray:
        """32-byte ordering keys of a batch of hashes (one bulk hex decode when they are all sha256 hex)."""
        if self._key_mode == "hex":
            try:
                joined = "".join(hs)
                if len(joined) == 64 * len(hs) and _HEXRUN.match(joined) and all(len(h) == 64 for h in hs):
                    return np.frombuffer(bytes.fromhex(joined), dtype=np.uint8)
            except TypeError:
                pass
        mode = self._key_mode
        keys = [self._key_bytes(h) for h in hs]
        if self._key_mode != mode:  # the batch itself changed the mode: every key of the batch under the final one
            keys = [self._key_bytes(h) for h in hs]
        return np.frombuffer(b"".join(keys), dtype=np.uint8)

    # ------------------------------------------------------------------ reference API

    @_locked
    def add_claim(
        self,
        claim_hash: str,
        text: str,
        tags: List[str],
        evidence: Optional[List[str]] = None,
        *,
        created_ms: Optional[int] = None,
        parent_hash: Optional[str] = None,
    ) -> None:
        if created_ms is None:
            created_ms = int(time.time() * 1000)
        # the reference re-embeds on every add_claim (:82), also when the hash already exists: whatever vector the
        # hash had is discarded at the next sync (batched with every other row that leaves the device)
        self._discard[claim_hash] = None
        self._pending_text[claim_hash] = text
        self.claims[claim_hash] = Claim(
            claim_hash=claim_hash,
            text=text,
            tags=list(tags),
            created_ms=int(created_ms),
            evidence=list(evidence or []),
            parent_hash=parent_hash,
            retracted=False,
        )

    @_locked
    def retract(self, claim_hash: str) -> None:
        c = self.claims.get(claim_hash)
        if c:
            c.retracted = True  # marks the hash dirty
            # the reference frees the cached embedding here (:89); the device row goes at the next sync, in one
            # compaction with every other retract since the last query
            self._discard[claim_hash] = None
            self._pending_text.pop(claim_hash, None)

    def compile(
        self, query: str, *, top_k: int = 8, since_ms: Optional[int] = None, tags: Optional[Sequence[str]] = None,
        lazy: bool = False,
    ) -> Tuple[str, List[str]]:
        """The reference's ``compile`` (lb/context_graph.py:91-133): rank the live claims against ``query``, format the best.

        Extensions, both off by default: ``tags`` keeps only claims carrying at least one of these tags;
        ``lazy=True`` returns a ``LazyContext`` (hashes now, text formatted on first use).
        """
        chosen, _ = self._rank(query, top_k, since_ms, tags)
        if lazy:
            return self._lazy(query, chosen)
        return self._format(query, chosen), chosen

    def _format(self, query: str, chosen: List[str]) -> str:
        """lb/context_graph.py:122-133."""
        claims = self.claims
        return format_context(query, [(claims[h].text, claims[h].tags) for h in chosen])

    def _lazy(self, query: str, chosen: List[str]) -> LazyContext:
        claims = self.claims
        return LazyContext(query, chosen, [(claims[h].text, tuple(claims[h].tags)) for h in chosen])

    def compile_scored(self, query: str, *, top_k: int = 8, since_ms: Optional[int] = None,
                       tags: Optional[Sequence[str]] = None, stats: Optional[_hip.Stats] = None
                       ) -> Tuple[List[str], List[float]]:
        """Ranking only: (claim hashes, fp64 scores) in the reference's order."""
        return self._rank(query, top_k, since_ms, tags, stats)

    def _mark_scored(self, since_ms, tags) -> None:
        """lb/context_graph.py:106-117: a compile() caches the vector of every live claim it scores (those that pass the
        time filter; with the ``tags`` extension, those that also pass it)."""
        if not self._unscored:
            return
        if since_ms is None and tags is None:
            self._unscored.clear()
            return
        want = set(tags) if tags is not None else None
        for h in list(self._unscored):
            c = self.claims.get(h)
            if c is None or (since_ms is not None and c.created_ms < since_ms) or (want is not None and not (want & set(c.tags))):
                continue
            del self._unscored[h]

    def _filters(self, since_ms, tags) -> Tuple[int, int, bool]:
        """(since for the C ABI, tag mask, provably-empty flag)."""
        if since_ms is None:
            since = _hip.SINCE_NONE
        else:
            since = since_ms if isinstance(since_ms, int) else int(np.ceil(since_ms))
            if since > _hip.INT64_MAX:
                return 0, 0, True  # created_ms < since_ms holds for every storable timestamp
            since = max(since, _hip.SINCE_NONE)
        tag_mask = 0
        if tags is not None:
            for t in tags:
                b = self._tag_bits.get(t)
                if b is not None:
                    tag_mask |= 1 << b
                elif self._tag_overflow:
                    raise NotImplementedError("tag filter: this graph uses more than 64 distinct tags")
            if tag_mask == 0:
                return since, 0, True  # no claim carries any of the requested tags
        return since, tag_mask, False

    def _rank(self, query, top_k, since_ms, tags, stats=None) -> Tuple[List[str], List[float]]:
        with self._lock:
            self._sync()
            self._mark_scored(since_ms, tags)
            k = max(0, int(top_k))  # :121
            since, tag_mask, empty = self._filters(since_ms, tags)
            if empty:
                return [], []
            index = self._index  # opens the HIP backend even for an empty graph: no silent CPU path
            if not self._overlap_searches:
                return self._search_text(index, query, k, since, tag_mask, stats)
            if k == 0 or not self._id_of:
                return [], []
            self._readers += 1
        try:  # the device part runs without the graph lock: another thread's compile() may overlap it
            ids, scores, _ = index.search_text(query, k, since, tag_mask, stats)
            with self._lock:
                return [self._hash_of[i] for i in ids.tolist()], scores.tolist()
        finally:
            with self._cond:
                self._readers -= 1
                self._cond.notify_all()

    def _search_text(self, index, query: str, k: int, since: int, tag_mask: int, stats) -> Tuple[List[str], List[float]]:
        """Device part of ``compile``: embed(query) + scan + exact top-k -> (hashes, fp64 scores)."""
        if k == 0 or not self._id_of:
            return [], []
        ids, scores, _ = index.search_text(query, k, since, tag_mask, stats)
        return [self._hash_of[i] for i in ids.tolist()], scores.tolist()

    @_locked
    def recent(self, since_ms, limit: Optional[int] = 100) -> List[str]:
        """Claim hashes of ``BatteryNode.get_recent_claims`` (lb/node.py:750-762) in the reference's order:
        live claims with ``created_ms >= since_ms``, newest first, ties in ``claims`` dict order, then
        ``[:limit]`` with Python's slice semantics (negative = drop from the end, None = all).  One masked
        top-``limit`` over the device's created_ms column (8 B/claim) instead of a Python loop + sort."""
        self._sync()
        if limit is not None:
            limit = operator.index(limit)  # a slice bound: TypeError for floats, like claims[:limit]
        index = self._index
        if isinstance(since_ms, float):
            if since_ms != since_ms or since_ms == float("inf"):
                return []  # created_ms >= since_ms holds for no claim
            since = _hip.SINCE_NONE if since_ms == float("-inf") else int(np.ceil(since_ms))
        else:
            since = int(since_ms)
        if since > _hip.INT64_MAX:
            return []
        since = max(since, _hip.SINCE_NONE)
        if not self._id_of:
            return []
        if self._order_pending:
            hs = [h for h in self._order_pending if h in self._id_of]
            self._order_pending = {}
            if hs:
                index.set_order(np.array([self._id_of[h] for h in hs], dtype=np.uint64),
                                np.array([self._seq_of[h] for h in hs], dtype=np.uint64))
        n = len(self._id_of)
        if limit is None:
            want = n
        elif limit < 0:
            _, _, matched = index.recent(since, 0)
            want = max(0, matched + limit)
        else:
            want = min(limit, n)
        if want == 0:
            return []
        ids, _, _ = index.recent(since, want)
        return [self._hash_of[i] for i in ids.tolist()]

    def recent_claims(self, since_ms, limit: Optional[int] = 100) -> List[Dict[str, Any]]:
        """The list ``BatteryNode.get_recent_claims`` returns (lb/node.py:750-762): ``recent`` picks and orders the claims
        on the GPU, only the chosen ones are turned into the reference's dict literal (:753-760) here."""
        claims = self.claims
        return [{"claim_hash": h, "text": claims[h].text, "tags": list(claims[h].tags), "created_ms": claims[h].created_ms,
                 "parent_hash": claims[h].parent_hash} for h in self.recent(since_ms, limit)]

    @_locked
    def compile_many(self, queries: Sequence[str], *, top_k: int = 8, since_ms: Optional[int] = None,
                     tags: Optional[Sequence[str]] = None, lazy: bool = False) -> List[Tuple[str, List[str]]]:
        """Batched ``compile`` (extension; the reference has no batched entry): the queries are embedded
        in one K1 launch and ranked in one pass over the claim matrix on the matrix cores (K3).  Each
        element equals ``compile(q, top_k=..., since_ms=..., tags=...)`` exactly (``lazy=True``: as ``LazyContext``)."""
        self._sync()
        if queries:
            self._mark_scored(since_ms, tags)
        k = max(0, int(top_k))
        since, tag_mask, empty = self._filters(since_ms, tags)
        index = self._index
        ranked: List[List[str]]
        if empty or k == 0 or not self._id_of or not queries:
            ranked = [[] for _ in queries]
        else:
            qv = _hip.embed_batch(list(queries))
            ids, _, cnt = index.search(qv, k, since, tag_mask)
            ranked = [[self._hash_of[i] for i in ids[j, : int(cnt[j])].tolist()] for j in range(len(queries))]
        if lazy:  # hashes now; each context text is formatted when (and if) its caller reads it
            return [self._lazy(q, chosen) for q, chosen in zip(queries, ranked)]
        return [(self._format(q, chosen), chosen) for q, chosen in zip(queries, ranked)]

    @_locked
    def snapshot(self) -> Dict[str, Any]:
        return {"claims": {h: c.to_dict() for h, c in self.claims.items()}}

    @staticmethod
    def from_snapshot(s: Dict[str, Any]) -> "ContextGraph":
        """lb/context_graph.py:139-148: every non-retracted claim is re-vectorised -- here as one
        batched K1 launch at the first ``compile`` / cache access instead of N Python ``embed`` calls."""
        g = ContextGraph()
        claims = s.get("claims", {}) or {}
        for h, cd in claims.items():
            c = Claim.from_dict(cd)
            g.claims[h] = c
            if not c.retracted:
                g._pending_text[h] = c.text  # :146-147: cached from the text as loaded
        return g
