"""ctypes front-end of the CPU parity oracle (oracle/lb_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg, never by the product package ``lbm_amd``.

Parity status: PINNED against golden vectors generated from the reference
itself (tests/golden/make_golden.py, tests/test_oracle_golden.py).

``OracleGraph`` mirrors the observable behaviour of the reference's
``ContextGraph`` (lb/context_graph.py:46-148): dict-ordered claims, an
embedding cache that ``retract`` empties, on-demand re-embedding in
``compile`` and the exact context-text format of :122-133.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblb_oracle.so")
DIM = 64


def build(force: bool = False) -> str:
    """Compile liblb_oracle.so with gcc (needs only oracle/lb_oracle.c)."""
    src = os.path.join(_HERE, "lb_oracle.c")
    if force or not os.path.exists(_SO) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liblb_oracle.so"])
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u8p, i64p, f64p = (ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int64),
                           ctypes.POINTER(ctypes.c_double))
        L.lbo_sha256.argtypes = [u8p, ctypes.c_int64, u8p]
        L.lbo_sha256.restype = None
        L.lbo_tokenize.argtypes = [u8p, ctypes.c_int64, i64p, i64p, ctypes.c_int64]
        L.lbo_tokenize.restype = ctypes.c_int64
        L.lbo_lut.argtypes = [ctypes.c_int]
        L.lbo_lut.restype = ctypes.c_double
        L.lbo_embed.argtypes = [u8p, ctypes.c_int64, f64p]
        L.lbo_embed.restype = None
        L.lbo_embed_dim.argtypes = [u8p, ctypes.c_int64, ctypes.c_int, f64p]
        L.lbo_embed_dim.restype = None
        L.lbo_embed_batch.argtypes = [u8p, i64p, ctypes.c_int64, f64p]
        L.lbo_embed_batch.restype = None
        L.lbo_cosine.argtypes = [f64p, ctypes.c_int64, f64p, ctypes.c_int64, f64p]
        L.lbo_cosine.restype = ctypes.c_int
        L.lbo_compile.argtypes = [f64p, i64p, u8p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, u8p, i64p,
                                  ctypes.c_int64, f64p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, i64p, f64p]
        L.lbo_compile.restype = ctypes.c_int64
        _lib = L
    return _lib


def _u8(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _i64(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def _f64(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _bytes_arr(b: bytes) -> np.ndarray:
    # +1 so that an empty text still has a valid pointer
    a = np.zeros(len(b) + 1, dtype=np.uint8)
    if b:
        a[: len(b)] = np.frombuffer(b, dtype=np.uint8)
    return a


def sha256(data: bytes) -> bytes:
    a = _bytes_arr(data)
    out = np.zeros(32, dtype=np.uint8)
    lib().lbo_sha256(_u8(a), len(data), _u8(out))
    return out.tobytes()


def tokenize(text: str) -> List[str]:
    """lb/latent.py:37-38 (_tokenize)."""
    b = text.encode("utf-8", "surrogatepass")
    a = _bytes_arr(b)
    n = lib().lbo_tokenize(_u8(a), len(b), None, None, 0)
    starts = np.zeros(max(n, 1), dtype=np.int64)
    lens = np.zeros(max(n, 1), dtype=np.int64)
    lib().lbo_tokenize(_u8(a), len(b), _i64(starts), _i64(lens), n)
    return [b[starts[i]: starts[i] + lens[i]].decode("ascii").lower() for i in range(n)]


def embed_array(text: str) -> np.ndarray:
    b = text.encode("utf-8", "surrogatepass")
    a = _bytes_arr(b)
    out = np.zeros(DIM, dtype=np.float64)
    lib().lbo_embed(_u8(a), len(b), _f64(out))
    return out


def embed(text: str, dim: int = DIM) -> List[float]:
    """lb/latent.py:41-57, any dim: the first `dim` digest bytes, norm over `dim` components; dim > 64 with at
    least one token is the reference's IndexError (:52 reads b[i] past the 64 digest bytes)."""
    if dim == DIM:
        return embed_array(text).tolist()
    if dim > DIM:
        if tokenize(text):
            raise IndexError("index out of range")
        return [0.0] * dim
    if dim <= 0:
        return []  # [0.0] * dim; with tokens the loops over range(dim) do nothing and the norm of [] is 0.0
    b = text.encode("utf-8", "surrogatepass")
    a = _bytes_arr(b)
    out = np.zeros(dim, dtype=np.float64)
    lib().lbo_embed_dim(_u8(a), len(b), int(dim), _f64(out))
    return out.tolist()


def pack_texts(texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """UTF-8 blob + CSR offsets (n+1), the layout the C ABI uses."""
    enc = [t.encode("utf-8", "surrogatepass") for t in texts]
    offsets = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc])
    blob = _bytes_arr(b"".join(enc))
    return blob, offsets


def embed_batch(texts: Sequence[str]) -> np.ndarray:
    blob, offsets = pack_texts(texts)
    return embed_batch_packed(blob, offsets)


def embed_batch_packed(blob: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    n = len(offsets) - 1
    out = np.zeros((n, DIM), dtype=np.float64)
    if n:
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        lib().lbo_embed_batch(_u8(blob), _i64(offsets), n, _f64(out))
    return out


def cosine(a: Sequence[float], b: Sequence[float]) -> float:
    """lb/latent.py:60-63."""
    if len(a) != len(b):
        raise ValueError("dimension mismatch")
    aa = np.ascontiguousarray(a, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64)
    out = ctypes.c_double(0.0)
    lib().lbo_cosine(_f64(aa), len(aa), _f64(bb), len(bb), ctypes.byref(out))
    return out.value


def compile_arrays(emb: np.ndarray, created_ms: np.ndarray, retracted: Optional[np.ndarray],
                   keys: Sequence[str], q: np.ndarray, top_k: int, since_ms: Optional[int] = None,
                   tagmask: Optional[np.ndarray] = None, tag_filter: int = 0,
                   packed_keys: Optional[Tuple[np.ndarray, np.ndarray]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """lb/context_graph.py:104-121 on arrays; returns (indices, scores) of the chosen claims."""
    n = emb.shape[0]
    emb = np.ascontiguousarray(emb, dtype=np.float64)
    created_ms = np.ascontiguousarray(created_ms, dtype=np.int64)
    if retracted is None:
        retracted = np.zeros(n, dtype=np.uint8)
    retracted = np.ascontiguousarray(retracted, dtype=np.uint8)
    if packed_keys is None:
        packed_keys = pack_texts(keys)
    kblob, koff = packed_keys
    q = np.ascontiguousarray(q, dtype=np.float64)
    k = max(0, int(top_k))
    kk = min(k, n)
    out_idx = np.zeros(max(kk, 1), dtype=np.int64)
    out_sc = np.zeros(max(kk, 1), dtype=np.float64)
    tm = None
    if tagmask is not None:
        tagmask = np.ascontiguousarray(tagmask, dtype=np.uint64)
        tm = tagmask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    cnt = lib().lbo_compile(_f64(emb), _i64(created_ms), _u8(retracted), tm, ctypes.c_uint64(int(tag_filter)),
                            _u8(kblob), _i64(koff), n, _f64(q), 1 if since_ms is not None else 0,
                            int(since_ms) if since_ms is not None else 0, kk, _i64(out_idx), _f64(out_sc))
    return out_idx[:cnt].copy(), out_sc[:cnt].copy()


class OracleClaim:
    __slots__ = ("claim_hash", "text", "tags", "created_ms", "evidence", "parent_hash", "retracted")

    def __init__(self, claim_hash, text, tags, created_ms, evidence, parent_hash=None, retracted=False):
        self.claim_hash, self.text, self.tags = claim_hash, text, list(tags)
        self.created_ms, self.evidence = int(created_ms), list(evidence)
        self.parent_hash, self.retracted = parent_hash, retracted


class OracleGraph:
    """Behavioural restatement of lb/context_graph.py:46-148 on top of the C oracle."""

    def __init__(self) -> None:
        self.claims: Dict[str, OracleClaim] = {}
        self._embedding_cache: Dict[str, np.ndarray] = {}

    def add_claim(self, claim_hash: str, text: str, tags: List[str], evidence: Optional[List[str]] = None, *,
                  created_ms: Optional[int] = None, parent_hash: Optional[str] = None) -> None:
        if created_ms is None:
            created_ms = int(time.time() * 1000)
        self.claims[claim_hash] = OracleClaim(claim_hash, text, tags, created_ms, evidence or [], parent_hash)
        self._embedding_cache[claim_hash] = embed_array(text)  # :82

    def retract(self, claim_hash: str) -> None:
        c = self.claims.get(claim_hash)
        if c:
            c.retracted = True
            self._embedding_cache.pop(claim_hash, None)  # :89

    def snapshot(self) -> Dict[str, object]:
        """lb/context_graph.py:135-137 (+ Claim.to_dict, :20-31: parent_hash only when set)."""
        out = {}
        for h, c in self.claims.items():
            d = {"claim_hash": c.claim_hash, "text": c.text, "tags": list(c.tags), "created_ms": int(c.created_ms),
                 "evidence": list(c.evidence), "retracted": bool(c.retracted)}
            if c.parent_hash is not None:
                d["parent_hash"] = c.parent_hash
            out[h] = d
        return {"claims": out}

    @staticmethod
    def from_snapshot(s: Dict[str, object]) -> "OracleGraph":
        """lb/context_graph.py:139-148: every non-retracted claim is re-embedded from its text."""
        g = OracleGraph()
        for h, cd in (s.get("claims", {}) or {}).items():
            c = OracleClaim(str(cd["claim_hash"]), str(cd.get("text", "")), list(cd.get("tags", [])),
                            int(cd.get("created_ms", 0)), list(cd.get("evidence", [])), cd.get("parent_hash"),
                            bool(cd.get("retracted", False)))
            g.claims[h] = c
            if not c.retracted:
                g._embedding_cache[h] = embed_array(c.text)
        return g

    def compile(self, query: str, *, top_k: int = 8, since_ms: Optional[int] = None,
                tags: Optional[Sequence[str]] = None) -> Tuple[str, List[str]]:
        chosen, _ = self.compile_scored(query, top_k=top_k, since_ms=since_ms, tags=tags)
        return format_context(query, [self.claims[h] for h in chosen]), chosen

    def compile_scored(self, query: str, *, top_k: int = 8, since_ms: Optional[int] = None,
                       tags: Optional[Sequence[str]] = None) -> Tuple[List[str], List[float]]:
        qv = embed_array(query)  # :104
        hashes = list(self.claims.keys())
        n = len(hashes)
        emb = np.zeros((n, DIM), dtype=np.float64)
        created = np.zeros(n, dtype=np.int64)
        retr = np.zeros(n, dtype=np.uint8)
        keep = np.ones(n, dtype=np.uint8)
        for i, h in enumerate(hashes):
            c = self.claims[h]
            created[i] = c.created_ms
            retr[i] = 1 if c.retracted else 0
            if c.retracted:
                continue
            if tags is not None and not (set(tags) & set(c.tags)):
                keep[i] = 0  # tag extension: any-of match (absent from the reference)
                continue
            if since_ms is not None and c.created_ms < since_ms:
                continue  # reference skips before touching the cache (:110)
            if h not in self._embedding_cache:  # :113-117
                self._embedding_cache[h] = embed_array(c.text)
            emb[i] = self._embedding_cache[h]
        retr = np.maximum(retr, 1 - keep).astype(np.uint8)
        idx, sc = compile_arrays(emb, created, retr, hashes, qv, top_k, since_ms)
        return [hashes[i] for i in idx], [float(s) for s in sc]


def recent_claims(claims: Dict[str, "OracleClaim"], since_ms, limit=100) -> List[Dict[str, object]]:
    """Restatement of BatteryNode.get_recent_claims, lb/node.py:750-762 (lb/node.py itself does not import
    here: its `cryptography` dependency is absent, so this function is pinned by the expectations of the
    reference's own tests -- tests/test_edge_cases.py:560-598, tests/test_multi_agent.py:170-193 -- which
    tests/test_recent.py replays, not by generated golden vectors)."""
    out = []
    for h, c in claims.items():  # :751
        if not c.retracted and c.created_ms >= since_ms:  # :752
            out.append({"claim_hash": h, "text": c.text, "tags": list(c.tags), "created_ms": c.created_ms,
                        "parent_hash": c.parent_hash})
    out.sort(key=lambda x: x["created_ms"], reverse=True)  # :761 (stable: ties keep dict order)
    return out[:limit]  # :762


def recent_arrays(created_ms: np.ndarray, order: np.ndarray, since_ms: int, limit: int) -> np.ndarray:
    """Array form of the same ordering for large cases: indices of rows with created_ms >= since_ms sorted by
    (created_ms descending, order ascending), cut to `limit` (>= 0)."""
    created_ms = np.asarray(created_ms, dtype=np.int64)
    idx = np.nonzero(created_ms >= since_ms)[0]
    # (created desc, order asc) without negating int64 (INT64_MIN has no negative): two stable sorts
    by_order = idx[np.argsort(np.asarray(order, dtype=np.uint64)[idx], kind="stable")]
    rev = by_order[::-1]                                  # order descending
    asc = rev[np.argsort(created_ms[rev], kind="stable")]  # created ascending, ties in order DESCENDING
    return asc[::-1][:max(0, int(limit))]                  # created descending, ties in order ascending


def format_context(query: str, chosen_claims) -> str:
    """lb/context_graph.py:122-133."""
    lines = ["Context slice", f"Query: {query}", ""]
    for i, c in enumerate(chosen_claims, 1):
        tag_str = (", ".join(c.tags)) if c.tags else ""
        if tag_str:
            lines.append(f"{i}. {c.text}  [tags: {tag_str}]")
        else:
            lines.append(f"{i}. {c.text}")
    return "\n".join(lines).strip() + "\n"
