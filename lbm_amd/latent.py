"""Drop-in for the reference's ``lb/latent.py`` backed by HIP kernels.

Same public surface -- ``_tokenize``, ``embed(text, dim=64)``, ``cosine(a, b)`` --
same return types (Python lists / floats) and the same bits: ``embed`` runs K1
(tokenise + SHA-256 feature hashing + in-order fp64 accumulation + normalise,
lbm_amd/csrc/embed.hip) and ``cosine`` the sequential fp64 dot kernel, through
the C ABI in include/lbhip.h.  No CPU fallback: without liblbhip.so and an
MI355X these raise ``HipBackendError``.
"""
from __future__ import annotations

import re
from typing import List, Sequence

import numpy as np

from . import _hip

_WORD_RE = re.compile(r"[A-Za-z0-9_]+")  # lb/latent.py:34 (host mirror of the device tokeniser)


def _tokenize(text: str) -> List[str]:
    """lb/latent.py:37-38.  Not numeric: kept on the host for callers that inspect tokens."""
    return [t.lower() for t in _WORD_RE.findall(text)]


def embed(text: str, dim: int = 64) -> List[float]:
    """lb/latent.py:41-57.  ``dim`` < 64 truncates the accumulation exactly like the reference (first ``dim``
    digest bytes, norm over the first ``dim`` components; computed by the same kernel); dim > 64 is the
    reference's IndexError once the text has a token; dim <= 0 is its empty list."""
    if dim == _hip.DIM:
        return _hip.embed_batch([text])[0].tolist()
    if dim > _hip.DIM:
        if _tokenize(text):
            raise IndexError("index out of range")  # b[i] with i >= 64, lb/latent.py:52
        return [0.0] * dim
    if dim <= 0:
        return [0.0] * dim
    return _hip.embed_batch([text], dim)[0, :dim].tolist()


def embed_many(texts: Sequence[str]) -> np.ndarray:
    """Batched ``embed``: (n, 64) float64, one kernel launch."""
    return _hip.embed_batch(texts)


def cosine(a: Sequence[float], b: Sequence[float]) -> float:
    """lb/latent.py:60-63 (plain dot product of already-normalised vectors)."""
    if len(a) != len(b):
        raise ValueError("dimension mismatch")
    aa = np.ascontiguousarray(a, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64)
    return _hip.cosine(aa, bb)
