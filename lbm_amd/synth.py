"""Deterministic synthetic claim corpora (SURVEY.md §8d / BASELINE.md "Inputs").

Vocabulary of 50 000 tokens ``"w%05d"`` drawn Zipf(s = 1.07); claims of 8-24
tokens; ``created_ms = 1_700_000_000_000 + 1000 * i``; 0-3 tags out of a 64-tag
universe (bit j of the u64 mask = tag ``"t%02d" % j``); 1 % retracted; 0.1 %
exact duplicate texts (forces hash tie-breaks); queries of 3-6 tokens.

Generation is chunked (each chunk is self-contained and seeded by
``(seed, chunk_index)``) so a 10 M- or 100 M-claim corpus can be streamed to the
GPU without ever holding all the text on the host.  Everything here is plain
numpy on the host: it produces *inputs*, it is not part of the retrieval path.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import Iterator, List, Optional

import numpy as np

VOCAB = 50_000
ZIPF_S = 1.07
N_TAGS = 64
BASE_MS = 1_700_000_000_000
TOKEN_BYTES = 6  # "w%05d"

_cdf_cache = {}


def _zipf_cdf(vocab: int = VOCAB, s: float = ZIPF_S) -> np.ndarray:
    key = (vocab, s)
    if key not in _cdf_cache:
        p = 1.0 / np.power(np.arange(1, vocab + 1, dtype=np.float64), s)
        c = np.cumsum(p)
        _cdf_cache[key] = c / c[-1]
    return _cdf_cache[key]


def _draw_tokens(rng: np.random.Generator, total: int) -> np.ndarray:
    ids = np.searchsorted(_zipf_cdf(), rng.random(total), side="left")
    return np.minimum(ids, VOCAB - 1).astype(np.int32)


def tokens_to_text_blob(tok: np.ndarray, lens: np.ndarray):
    """Ragged token ids -> (uint8 blob, int64 offsets[n+1]); tokens joined by one space."""
    total = int(tok.shape[0])
    n = int(lens.shape[0])
    mat = np.empty((total, TOKEN_BYTES + 1), dtype=np.uint8)
    mat[:, 0] = ord("w")
    t = tok.astype(np.int64)
    for d in range(5):
        mat[:, 5 - d] = (t % 10 + ord("0")).astype(np.uint8)
        t //= 10
    mat[:, 6] = ord(" ")
    ends = np.cumsum(lens.astype(np.int64))
    keep = np.ones(total * (TOKEN_BYTES + 1), dtype=bool)
    nz = lens > 0
    keep[ends[nz] * (TOKEN_BYTES + 1) - 1] = False  # drop the space after each claim's last token
    blob = mat.reshape(-1)[keep]
    sizes = lens.astype(np.int64) * (TOKEN_BYTES + 1) - nz.astype(np.int64)
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(sizes)
    return blob, offsets


@dataclass
class CorpusChunk:
    start: int              # global index of the first claim of this chunk
    blob: np.ndarray        # uint8 UTF-8 text of all claims, concatenated
    offsets: np.ndarray     # int64 [n+1]
    created_ms: np.ndarray  # int64 [n]
    tagmask: np.ndarray     # uint64 [n]
    retracted: np.ndarray   # bool [n]
    keys: np.ndarray        # uint8 [n, 32] raw claim-hash bytes (hex of these = the claim hash string)
    tok: np.ndarray         # int32 ragged token ids (kept for cross-checks)
    lens: np.ndarray        # int32 [n]

    @property
    def n(self) -> int:
        return int(self.lens.shape[0])

    def text(self, i: int) -> str:
        return bytes(self.blob[self.offsets[i]: self.offsets[i + 1]]).decode("ascii")

    def texts(self) -> List[str]:
        return [self.text(i) for i in range(self.n)]

    def hash_hex(self, i: int) -> str:
        return bytes(self.keys[i]).hex()

    def hashes(self) -> List[str]:
        return [self.hash_hex(i) for i in range(self.n)]

    def tags(self, i: int) -> List[str]:
        m = int(self.tagmask[i])
        return ["t%02d" % j for j in range(N_TAGS) if (m >> j) & 1]


def make_chunk(start: int, n: int, seed: int = 1234, chunk_index: int = 0, *, real_sha_keys: bool = True,
               dup_frac: float = 0.001, retract_frac: float = 0.01) -> CorpusChunk:
    """Claims [start, start+n).  ``real_sha_keys``: hash = sha256(f"{seed}|{i}"); for
    multi-million-claim bench corpora pass False to draw the 32 key bytes from the RNG instead
    (same distribution, ~100x faster to generate)."""
    rng = np.random.default_rng([seed, chunk_index])
    lens = rng.integers(8, 25, size=n, dtype=np.int32)
    # duplicates: claim i copies the token list of an earlier claim of the same chunk
    n_dup = int(round(n * dup_frac))
    dup_dst = np.sort(rng.choice(np.arange(1, n), size=n_dup, replace=False)) if n_dup and n > 1 else np.zeros(0, np.int64)
    dup_src = (rng.random(dup_dst.shape[0]) * dup_dst).astype(np.int64)  # uniform in [0, dst)
    for d, s in zip(dup_dst, dup_src):  # sorted ascending, so chained copies resolve in order
        lens[d] = lens[s]
    tok = _draw_tokens(rng, int(lens.sum()))
    starts = np.zeros(n + 1, dtype=np.int64)
    starts[1:] = np.cumsum(lens.astype(np.int64))
    for d, s in zip(dup_dst, dup_src):
        tok[starts[d]: starts[d + 1]] = tok[starts[s]: starts[s + 1]]
    blob, offsets = tokens_to_text_blob(tok, lens)
    idx = np.arange(start, start + n, dtype=np.int64)
    created = BASE_MS + 1000 * idx
    n_tags = rng.integers(0, 4, size=n)
    tagmask = np.zeros(n, dtype=np.uint64)
    for j in range(3):
        bit = rng.integers(0, N_TAGS, size=n).astype(np.uint64)
        tagmask |= np.where(n_tags > j, np.uint64(1) << bit, np.uint64(0)).astype(np.uint64)
    retracted = rng.random(n) < retract_frac
    if real_sha_keys:
        keys = np.frombuffer(b"".join(hashlib.sha256(f"{seed}|{i}".encode()).digest() for i in idx.tolist()),
                             dtype=np.uint8).reshape(n, 32).copy()
    else:
        keys = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    return CorpusChunk(start, blob, offsets, created, tagmask, retracted, keys, tok, lens)


def iter_corpus(n_total: int, chunk: int = 1_000_000, seed: int = 1234, *, real_sha_keys: Optional[bool] = None,
                first_chunk_index: int = 0, start: int = 0) -> Iterator[CorpusChunk]:
    if real_sha_keys is None:
        real_sha_keys = n_total <= 200_000
    done, ci = 0, first_chunk_index
    while done < n_total:
        m = min(chunk, n_total - done)
        yield make_chunk(start + done, m, seed, ci, real_sha_keys=real_sha_keys)
        done += m
        ci += 1


def make_queries(nq: int, seed: int = 4321) -> List[str]:
    rng = np.random.default_rng([seed, 0])
    lens = rng.integers(3, 7, size=nq, dtype=np.int32)
    tok = _draw_tokens(rng, int(lens.sum()))
    out, p = [], 0
    for L in lens.tolist():
        out.append(" ".join("w%05d" % t for t in tok[p: p + L].tolist()))
        p += L
    return out
