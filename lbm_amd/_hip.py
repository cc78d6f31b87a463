"""ctypes binding of liblbhip.so (C ABI: include/lbhip.h).

This is the only door between Python and the GPU.  There is deliberately no CPU
fallback: if the shared library is missing, or no gfx950 device can be opened,
every entry point raises ``HipBackendError`` -- loudly, at first use.

PyTorch is not needed here; where a caller wants torch to own a device buffer it
passes ``tensor.data_ptr()`` to the ``*_dev`` entry points.
"""
from __future__ import annotations

import ctypes
import os
import sys
import threading
from typing import Optional, Sequence, Tuple

import numpy as np

DIM = 64
KEY_BYTES = 32
MAX_PAGE = 64
SINCE_NONE = -(2 ** 63)
INT64_MAX = 2 ** 63 - 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LBM_HIP_LIB") or os.path.join(_HERE, "liblbhip.so")  # LBM_HIP_LIB: A/B builds

EXPORTS = [
    "lbh_init", "lbh_shutdown", "lbh_last_error", "lbh_version", "lbh_device_name", "lbh_set_option",
    "lbh_host_alloc", "lbh_host_free", "lbh_embed_batch", "lbh_embed_batch_dim", "lbh_embed_batch_dev", "lbh_cosine",
    "lbh_index_create", "lbh_index_destroy", "lbh_index_reserve", "lbh_index_add", "lbh_index_retract",
    "lbh_index_retract_many", "lbh_index_update_meta", "lbh_index_set_keys", "lbh_index_size", "lbh_index_get_embedding",
    "lbh_index_get_embeddings",
    "lbh_index_search", "lbh_index_search_text", "lbh_index_search_shard",
    "lbh_index_recent", "lbh_index_set_order", "lbh_merge_topk", "lbh_search_dev_noalloc",
    "lbh_comm_unique_id", "lbh_comm_init", "lbh_comm_destroy", "lbh_index_search_sharded",
    "lbh_index_search_sharded_many", "lbh_comm_allgather",
]


class HipBackendError(RuntimeError):
    """The HIP backend is unavailable or a call into it failed."""


class Stats(ctypes.Structure):
    _fields_ = [
        ("want_timing", ctypes.c_int32), ("path", ctypes.c_int32),
        ("rows_scanned", ctypes.c_int64), ("bytes_per_row", ctypes.c_int64), ("candidates", ctypes.c_int64),
        ("scan_launches", ctypes.c_int32), ("fallbacks", ctypes.c_int32),
        ("scan_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
        ("max_score_err", ctypes.c_float), ("margin", ctypes.c_float),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


_lib = None
_lib_lock = threading.RLock()
_ctx = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7) and ask for it as plain `libamdhip64.so`, which the dynamic linker does not match
    against an already loaded /opt/rocm copy: if liblbhip.so pulled in the system runtime first, torch
    would bring up a second runtime that finds no GPU ("ProcessGroupNCCL ... no GPUs found").  Loading
    the bundled copy first makes liblbhip.so's `libamdhip64.so.7` dependency bind to it, and torch later
    re-uses the same file.  Without torch (or with LBM_HIP_SYSTEM_RUNTIME=1) the RUNPATH copy is used."""
    if os.environ.get("LBM_HIP_SYSTEM_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the RUNPATH runtime; torch then has to be imported before lbm_amd


def load_library() -> ctypes.CDLL:
    """dlopen liblbhip.so and declare every prototype of include/lbhip.h.  Needs no GPU."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipBackendError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C lbm_amd/csrc`). lbm_amd has no CPU fallback.")
        _preload_torch_hip_runtime()
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # missing ROCm runtime etc.
            raise HipBackendError(f"cannot load {LIB_PATH}: {e}") from e
        L.lbh_last_error.restype = ctypes.c_char_p
        L.lbh_last_error.argtypes = []
        L.lbh_version.restype = ctypes.c_int
        L.lbh_version.argtypes = []
        L.lbh_init.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(_vp)]
        L.lbh_shutdown.argtypes = [_vp]
        L.lbh_device_name.argtypes = [_vp, ctypes.c_char_p, ctypes.c_int]
        L.lbh_set_option.argtypes = [_vp, ctypes.c_char_p, _i64]
        L.lbh_embed_batch.argtypes = [_vp, _vp, _vp, _i64, _vp]
        L.lbh_host_alloc.argtypes = [_vp, _i64, ctypes.POINTER(_vp)]
        L.lbh_host_free.argtypes = [_vp, _vp]
        L.lbh_embed_batch_dim.argtypes = [_vp, _vp, _vp, _i64, ctypes.c_int, _vp]
        L.lbh_embed_batch_dev.argtypes = [_vp, _vp, _vp, _i64, _i64, _vp, _vp]
        L.lbh_cosine.argtypes = [_vp, _vp, _i64, _vp, _i64, ctypes.POINTER(ctypes.c_double)]
        L.lbh_index_create.argtypes = [_vp, ctypes.POINTER(_vp)]
        L.lbh_index_destroy.argtypes = [_vp]
        L.lbh_index_reserve.argtypes = [_vp, _i64]
        L.lbh_index_add.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.lbh_index_retract.argtypes = [_vp, _u64]
        L.lbh_index_retract_many.argtypes = [_vp, _i64, _vp]
        L.lbh_index_get_embeddings.argtypes = [_vp, _i64, _vp, _vp]
        L.lbh_index_update_meta.argtypes = [_vp, _u64, _i64, _u64, _vp]
        L.lbh_index_set_keys.argtypes = [_vp, _i64, _vp, _vp]
        L.lbh_index_size.argtypes = [_vp, ctypes.POINTER(_i64)]
        L.lbh_index_get_embedding.argtypes = [_vp, _u64, _vp]
        L.lbh_index_search.argtypes = [_vp, _vp, _i64, _i64, _i64, _u64, _vp, _vp, _vp, ctypes.POINTER(Stats)]
        L.lbh_index_search_text.argtypes = [_vp, _vp, _i64, _i64, _i64, _u64, _vp, _vp, ctypes.POINTER(_i64), _vp,
                                            ctypes.POINTER(Stats)]
        L.lbh_index_recent.argtypes = [_vp, _i64, _i64, _vp, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                       ctypes.POINTER(Stats)]
        L.lbh_index_set_order.argtypes = [_vp, _i64, _vp, _vp]
        L.lbh_merge_topk.argtypes = [_vp, _i64, _i64, _vp, ctypes.POINTER(_i64)]
        L.lbh_index_search_shard.argtypes = [_vp, _vp, _i64, _i64, _u64, _vp, _vp, _vp, ctypes.POINTER(_i64),
                                             ctypes.POINTER(Stats)]
        L.lbh_search_dev_noalloc.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, ctypes.c_double, _vp, _i64, _i64, _u64,
                                             _vp, _vp, _vp]
        L.lbh_comm_unique_id.argtypes = [_vp]
        L.lbh_comm_init.argtypes = [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]
        L.lbh_comm_destroy.argtypes = [_vp]
        L.lbh_index_search_sharded.argtypes = [_vp, _vp, _vp, _i64, _i64, _u64, _u64, _u64, _vp, _vp, _vp, ctypes.POINTER(_i64),
                                               ctypes.POINTER(Stats)]
        L.lbh_index_search_sharded_many.argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _u64, _u64, _u64, _vp, _vp, _vp, _vp,
                                                    ctypes.POINTER(Stats)]
        L.lbh_comm_allgather.argtypes = [_vp, _vp, _i64, _vp]
        for name in EXPORTS:
            if name not in ("lbh_last_error", "lbh_version"):
                getattr(L, name).restype = ctypes.c_int
        _lib = L
        return L


def _check(rc: int) -> None:
    if rc != 0:
        msg = load_library().lbh_last_error().decode("utf-8", "replace")
        if rc == -1 and msg == "dimension mismatch":
            raise ValueError("dimension mismatch")  # lb/latent.py:62
        raise HipBackendError(f"liblbhip error {rc}: {msg}")


def default_device() -> int:
    """The GPU this process drives.  ``LB_HIP_DEVICES`` (the reference's ``LB_*`` style, lb/config.py:49-235) lists the
    ordinals a node may use, e.g. "0,1,2,3": rank ``LOCAL_RANK`` takes entry ``LOCAL_RANK`` (one process per GPU), a
    single process the first.  Without it: ``LBM_HIP_DEVICE``, else ``LOCAL_RANK``, else 0."""
    devs = os.environ.get("LB_HIP_DEVICES", "").strip()
    if devs:
        try:
            lst = [int(x) for x in devs.replace(" ", "").split(",") if x != ""]
        except ValueError:
            raise HipBackendError(f"LB_HIP_DEVICES={devs!r}: expected a comma-separated list of GPU ordinals") from None
        if not lst or min(lst) < 0:
            raise HipBackendError(f"LB_HIP_DEVICES={devs!r}: expected a comma-separated list of GPU ordinals")
        lr = os.environ.get("LOCAL_RANK", "").strip()
        i = int(lr) if lr else 0
        if i >= len(lst):
            raise HipBackendError(f"LB_HIP_DEVICES lists {len(lst)} GPUs but LOCAL_RANK is {i}")
        return lst[i]
    for var in ("LBM_HIP_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return int(v)
    return 0


def check_backend_choice() -> None:
    """``LB_RETRIEVAL_BACKEND`` = auto | hip | python.  This package IS the HIP backend and carries no CPU path, so
    "auto" and "hip" mean the same (fail loudly without liblbhip.so + an MI355X) and "python" is refused: a deployment
    that wants the reference's Python loop imports ``lb.latent`` / ``lb.context_graph`` instead of this package."""
    v = os.environ.get("LB_RETRIEVAL_BACKEND", "auto").strip().lower() or "auto"
    if v in ("auto", "hip"):
        return
    if v == "python":
        raise HipBackendError("LB_RETRIEVAL_BACKEND=python: lbm_amd has no CPU path (import the reference's lb.latent / "
                              "lb.context_graph for that); set it to 'hip' or 'auto' to use this package")
    raise HipBackendError(f"LB_RETRIEVAL_BACKEND={v!r}: expected auto, hip or python")


def context():
    """Process-wide context on one GPU (LBM_HIP_DEVICE, else LOCAL_RANK, else 0)."""
    global _ctx
    L = load_library()
    with _lib_lock:
        if _ctx is None:
            check_backend_choice()
            dev = (ctypes.c_int * 1)(default_device())
            h = _vp()
            _check(L.lbh_init(dev, 1, ctypes.byref(h)))
            _ctx = h
        return _ctx


def set_option(name: str, value: int) -> None:
    _check(load_library().lbh_set_option(context(), name.encode(), int(value)))


def merge_topk(lists: np.ndarray, k: int) -> np.ndarray:
    """lists: int64 [n_lists, k + 1, 6] (sharded.pack_local layout) -> int64 [m, 6], the global top-k entries."""
    lists = np.ascontiguousarray(lists, dtype=np.int64)
    k = int(k)
    if lists.ndim != 3 or lists.shape[1] != k + 1 or lists.shape[2] != 6:
        raise ValueError("lists must have shape [n_lists, k + 1, 6]")
    out = np.empty((max(k, 1), 6), dtype=np.int64)
    cnt = _i64(0)
    _check(load_library().lbh_merge_topk(_ptr(lists), lists.shape[0], k, _ptr(out), ctypes.byref(cnt)))
    return out[:cnt.value]


def device_name() -> str:
    buf = ctypes.create_string_buffer(128)
    _check(load_library().lbh_device_name(context(), buf, 128))
    return buf.value.decode()


def _since(since_ms) -> int:
    """since_ms as the C ABI's int64: anything at or below INT64_MIN means "no bound" (LBH_SINCE_NONE); a bound no
    int64 timestamp can reach is an error here -- callers (ContextGraph._filters, Index.recent) answer it themselves."""
    v = max(int(since_ms), SINCE_NONE)
    if v > INT64_MAX:
        raise ValueError("since_ms exceeds the int64 range of created_ms")
    return v


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_vp)


class PinnedArray:
    """A numpy array over page-locked host memory from ``lbh_host_alloc``: arrays handed to ``Index.add_texts`` /
    ``add_vectors`` from here cross PCIe by asynchronous DMA.  ``.a`` is the array; keep the object alive while it is used."""

    def __init__(self, shape, dtype) -> None:
        self._L = load_library()
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
        p = _vp()
        _check(self._L.lbh_host_alloc(context(), max(n * dt.itemsize, 1), ctypes.byref(p)))
        self._p = p
        buf = (ctypes.c_char * max(n * dt.itemsize, 1)).from_address(p.value)
        self.a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    def close(self) -> None:
        if getattr(self, "_p", None):
            self.a = None
            self._L.lbh_host_free(context(), self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pinned_like(arr: np.ndarray) -> PinnedArray:
    """Copy of ``arr`` in pinned memory."""
    p = PinnedArray(arr.shape, arr.dtype)
    p.a[...] = arr
    return p


def pack_texts(texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """UTF-8 blob + int64 CSR offsets [n+1].  Lone surrogates are kept (they are separators anyway)."""
    enc = [t.encode("utf-8", "surrogatepass") for t in texts]
    offsets = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc])
    blob = np.frombuffer(b"".join(enc) + b"\x00", dtype=np.uint8).copy()
    return blob, offsets


def embed_batch_packed(blob: np.ndarray, offsets: np.ndarray, dim: int = DIM) -> np.ndarray:
    n = int(offsets.shape[0]) - 1
    out = np.empty((n, DIM), dtype=np.float64)
    if n > 0:
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        _check(load_library().lbh_embed_batch_dim(context(), _ptr(blob), _ptr(offsets), n, int(dim), _ptr(out)))
    return out


def embed_batch(texts: Sequence[str], dim: int = DIM) -> np.ndarray:
    """(n, 64) float64; with dim < 64 the components >= dim are 0 (callers slice [:dim])."""
    blob, offsets = pack_texts(texts)
    return embed_batch_packed(blob, offsets, dim)


def embed_batch_dev(utf8_ptr: int, offsets_ptr: int, n: int, total_bytes: int, out64_ptr: int, out32_ptr: int = 0):
    _check(load_library().lbh_embed_batch_dev(context(), utf8_ptr, offsets_ptr, n, total_bytes, out64_ptr,
                                              out32_ptr or None))


def search_dev_noalloc(emb32_ptr: int, emb64_ptr: int, created_ptr: int, tags_ptr: int, keywords_ptr: int, n: int,
                       max_row_norm: float, q: np.ndarray, k: int, since_ms: int, tag_mask: int, out_rows_ptr: int,
                       out_scores_ptr: int, out_count_ptr: int) -> None:
    """lbh_search_dev_noalloc: device pointers in and out (e.g. torch tensors' data_ptr()), no allocation inside."""
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(DIM)
    _check(load_library().lbh_search_dev_noalloc(context(), emb32_ptr or None, emb64_ptr or None, created_ptr or None,
                                                 tags_ptr or None, keywords_ptr or None, int(n), float(max_row_norm),
                                                 _ptr(q), int(k), _since(since_ms), int(tag_mask), out_rows_ptr or None,
                                                 out_scores_ptr or None, out_count_ptr or None))


def cosine(a: np.ndarray, b: np.ndarray) -> float:
    out = ctypes.c_double(0.0)
    _check(load_library().lbh_cosine(context(), _ptr(a), a.shape[0], _ptr(b), b.shape[0], ctypes.byref(out)))
    return out.value


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """Rendezvous id of a claim-sharded search group (rank 0 draws it and hands it to the other ranks)."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().lbh_comm_unique_id(buf))
    return buf.raw


class Comm:
    """One rank's end of the sharded search's exchange (``lbh_comm*``: an RCCL communicator on this process's GPU).
    Creating it is collective: it returns once all ``world`` ranks have called it with the same id."""

    def __init__(self, uid: bytes, rank: int, world: int) -> None:
        if len(uid) != COMM_ID_BYTES:
            raise ValueError("uid must be the %d bytes comm_unique_id() returned" % COMM_ID_BYTES)
        self._L = load_library()
        self.rank, self.world = int(rank), int(world)
        h = _vp()
        _check(self._L.lbh_comm_init(context(), ctypes.c_char_p(uid), self.rank, self.world, ctypes.byref(h)))
        self._h = h

    def allgather(self, payload: bytes) -> list:
        """Every rank's ``payload`` (any length), rank-major, through the communicator: one all-gather of the lengths,
        one of the payloads padded to the longest (``lbh_comm_allgather``).  Collective."""
        W = self.world
        mine = np.array([len(payload)], dtype=np.int64)
        sizes = np.zeros(W, dtype=np.int64)
        _check(self._L.lbh_comm_allgather(self._h, _ptr(mine), 8, _ptr(sizes)))
        width = int(sizes.max())
        if width == 0:
            return [b""] * W
        send = np.zeros(width, dtype=np.uint8)
        send[:len(payload)] = np.frombuffer(payload, dtype=np.uint8)
        recv = np.zeros(W * width, dtype=np.uint8)
        _check(self._L.lbh_comm_allgather(self._h, _ptr(send), width, _ptr(recv)))
        return [recv[r * width: r * width + int(sizes[r])].tobytes() for r in range(W)]

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.lbh_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """Thin owner of an ``lbh_index*`` (an HBM-resident claim index)."""

    def __init__(self) -> None:
        self._L = load_library()
        h = _vp()
        _check(self._L.lbh_index_create(context(), ctypes.byref(h)))
        self._h = h
        self._tls = threading.local()  # cached output buffers of search_shard / search_sharded: per index AND per thread

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.lbh_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, capacity: int) -> None:
        _check(self._L.lbh_index_reserve(self._h, int(capacity)))

    def size(self) -> int:
        out = _i64(0)
        _check(self._L.lbh_index_size(self._h, ctypes.byref(out)))
        return out.value

    def add_texts(self, blob: np.ndarray, offsets: np.ndarray, created_ms: np.ndarray, tagmask: Optional[np.ndarray],
                  keys: np.ndarray) -> np.ndarray:
        n = int(offsets.shape[0]) - 1
        ids = np.empty(n, dtype=np.uint64)
        if n == 0:
            return ids
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        created_ms = np.ascontiguousarray(created_ms, dtype=np.int64)
        keys = np.ascontiguousarray(keys, dtype=np.uint8).reshape(n, KEY_BYTES)
        if tagmask is not None:
            tagmask = np.ascontiguousarray(tagmask, dtype=np.uint64)
        _check(self._L.lbh_index_add(self._h, n, None, _ptr(blob), _ptr(offsets), _ptr(created_ms), _ptr(tagmask),
                                     _ptr(keys), _ptr(ids)))
        return ids

    def add_vectors(self, emb: np.ndarray, created_ms: np.ndarray, tagmask: Optional[np.ndarray],
                    keys: np.ndarray) -> np.ndarray:
        emb = np.ascontiguousarray(emb, dtype=np.float64).reshape(-1, DIM)
        n = emb.shape[0]
        ids = np.empty(n, dtype=np.uint64)
        if n == 0:
            return ids
        created_ms = np.ascontiguousarray(created_ms, dtype=np.int64)
        keys = np.ascontiguousarray(keys, dtype=np.uint8).reshape(n, KEY_BYTES)
        if tagmask is not None:
            tagmask = np.ascontiguousarray(tagmask, dtype=np.uint64)
        _check(self._L.lbh_index_add(self._h, n, _ptr(emb), None, None, _ptr(created_ms), _ptr(tagmask), _ptr(keys),
                                     _ptr(ids)))
        return ids

    def retract(self, cid: int) -> None:
        _check(self._L.lbh_index_retract(self._h, int(cid)))

    def retract_many(self, cids) -> None:
        ids = np.ascontiguousarray(cids, dtype=np.uint64).reshape(-1)
        if ids.shape[0]:
            _check(self._L.lbh_index_retract_many(self._h, ids.shape[0], _ptr(ids)))

    def update_meta(self, cid: int, created_ms: int, tagmask: int, key: bytes) -> None:
        k = np.frombuffer(key, dtype=np.uint8).copy()
        _check(self._L.lbh_index_update_meta(self._h, int(cid), int(created_ms), int(tagmask), _ptr(k)))

    def set_keys(self, ids: np.ndarray, keys: np.ndarray) -> None:
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        keys = np.ascontiguousarray(keys, dtype=np.uint8).reshape(ids.shape[0], KEY_BYTES)
        _check(self._L.lbh_index_set_keys(self._h, ids.shape[0], _ptr(ids), _ptr(keys)))

    def set_order(self, ids: np.ndarray, order: np.ndarray) -> None:
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        order = np.ascontiguousarray(order, dtype=np.uint64).reshape(ids.shape[0])
        _check(self._L.lbh_index_set_order(self._h, ids.shape[0], _ptr(ids), _ptr(order)))

    def recent(self, since_ms: int = SINCE_NONE, limit: int = 100, stats: Optional[Stats] = None):
        """Newest-first claim ids with created_ms >= since_ms.  Returns (ids, created_ms, matched)."""
        limit = max(0, min(int(limit), self.size()))
        since_ms = max(int(since_ms), SINCE_NONE)
        if since_ms > INT64_MAX:  # no storable timestamp reaches the bound
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.int64), 0
        ids = np.zeros(max(limit, 1), dtype=np.uint64)
        created = np.zeros(max(limit, 1), dtype=np.int64)
        cnt, matched = _i64(0), _i64(0)
        _check(self._L.lbh_index_recent(self._h, int(since_ms), limit, _ptr(ids), _ptr(created), ctypes.byref(cnt),
                                        ctypes.byref(matched), ctypes.byref(stats) if stats is not None else None))
        return ids[:cnt.value], created[:cnt.value], int(matched.value)

    def get_embedding(self, cid: int) -> np.ndarray:
        out = np.empty(DIM, dtype=np.float64)
        _check(self._L.lbh_index_get_embedding(self._h, int(cid), _ptr(out)))
        return out

    def get_embeddings(self, cids) -> np.ndarray:
        ids = np.ascontiguousarray(cids, dtype=np.uint64).reshape(-1)
        out = np.empty((ids.shape[0], DIM), dtype=np.float64)
        if ids.shape[0]:
            _check(self._L.lbh_index_get_embeddings(self._h, ids.shape[0], _ptr(ids), _ptr(out)))
        return out

    def search(self, q: np.ndarray, k: int, since_ms: int = SINCE_NONE, tag_mask: int = 0,
               stats: Optional[Stats] = None):
        """q: (nq, 64) float64.  Returns (ids [nq, k] uint64, scores [nq, k] float64, counts [nq] int64)."""
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, DIM)
        nq = q.shape[0]
        k = max(0, int(k))
        kk = min(k, max(self.size(), 1))
        ids = np.zeros((nq, kk), dtype=np.uint64)
        sc = np.zeros((nq, kk), dtype=np.float64)
        cnt = np.zeros(nq, dtype=np.int64)
        if nq and k:
            _check(self._L.lbh_index_search(self._h, _ptr(q), nq, kk, _since(since_ms), int(tag_mask), _ptr(ids),
                                            _ptr(sc), _ptr(cnt), ctypes.byref(stats) if stats is not None else None))
        return ids, sc, cnt

    def search_text(self, query: str, k: int, since_ms: int = SINCE_NONE, tag_mask: int = 0,
                    stats: Optional[Stats] = None, want_qvec: bool = False):
        b = query.encode("utf-8", "surrogatepass")  # (handed over as it is: ctypes passes a bytes object's buffer for a void*)
        k = max(0, int(k))
        kk = min(k, max(self.size(), 1)) if k > 4096 else k  # (the library clamps k to the live claims itself)
        # output buffers and their ctypes pointers are kept per thread and per k: building them costs ~8 us a call
        ids, sc, _, cnt, p_ids, p_sc, _, p_cnt, _, _ = self._out_bufs(kk)
        qv = np.zeros(DIM, dtype=np.float64) if want_qvec else None
        _check(self._L.lbh_index_search_text(self._h, b, len(b), kk, _since(since_ms), int(tag_mask), p_ids,
                                             p_sc, p_cnt, _ptr(qv),
                                             ctypes.byref(stats) if stats is not None else None))
        n = cnt.value
        return ids[:n].copy(), sc[:n].copy(), qv

    def search_sharded(self, comm: "Comm", q: np.ndarray, k: int, since_ms: int = SINCE_NONE, tag_mask: int = 0,
                       stats: Optional[Stats] = None, gid_mul: Optional[int] = None, gid_add: Optional[int] = None):
        """Collective over ``comm``: global top-k of the claim-sharded corpus -> (global ids, scores, keys[n, 32]),
        identical on every rank.  Global id = local id * gid_mul + gid_add (default: world, rank = round-robin deal)."""
        if not (isinstance(q, np.ndarray) and q.dtype == np.float64 and q.shape == (DIM,) and q.flags.c_contiguous):
            q = np.ascontiguousarray(q, dtype=np.float64).reshape(DIM)
        k = max(0, int(k))
        mul = comm.world if gid_mul is None else int(gid_mul)
        add = comm.rank if gid_add is None else int(gid_add)
        ids, sc, keys, cnt, p_ids, p_sc, p_keys, p_cnt, _, _ = self._out_bufs(k)
        _check(self._L.lbh_index_search_sharded(self._h, comm._h, q.ctypes.data, k, _since(since_ms), int(tag_mask), mul, add,
                                                p_ids, p_sc, p_keys, p_cnt, ctypes.byref(stats) if stats is not None else None))
        n = cnt.value
        return ids[:n].copy(), sc[:n].copy(), keys[:n].copy()

    def search_sharded_many(self, comm: "Comm", q: np.ndarray, k: int, since_ms: int = SINCE_NONE, tag_mask: int = 0,
                            stats: Optional[Stats] = None, gid_mul: Optional[int] = None, gid_add: Optional[int] = None):
        """Collective over ``comm``: the global top-k of every query of a batch (local K3 pass on each rank's shard, ONE
        all-gather, nq merges) -> (global ids [nq, k], scores [nq, k], keys [nq, k, 32], counts [nq])."""
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, DIM)
        nq, k = q.shape[0], max(0, int(k))
        mul = comm.world if gid_mul is None else int(gid_mul)
        add = comm.rank if gid_add is None else int(gid_add)
        kk = max(k, 1)
        ids = np.zeros((nq, kk), dtype=np.uint64)
        sc = np.zeros((nq, kk), dtype=np.float64)
        keys = np.zeros((nq, kk, KEY_BYTES), dtype=np.uint8)
        cnt = np.zeros(max(nq, 1), dtype=np.int64)
        _check(self._L.lbh_index_search_sharded_many(self._h, comm._h, _ptr(q), nq, k, _since(since_ms), int(tag_mask), mul, add,
                                                     _ptr(ids), _ptr(sc), _ptr(keys), _ptr(cnt),
                                                     ctypes.byref(stats) if stats is not None else None))
        return ids[:, :k], sc[:, :k], keys[:, :k], cnt[:nq]

    def _out_bufs(self, k: int):
        """Output arrays and their ctypes pointers, kept per thread and per k (building them costs more than the call)."""
        bufs = getattr(self._tls, "bufs", None)
        if bufs is None:
            bufs = self._tls.bufs = {}
        buf = bufs.get(k)
        if buf is None:
            if len(bufs) > 64:
                bufs.clear()
            kk = max(k, 1)
            ids = np.zeros(kk, dtype=np.uint64)
            sc = np.zeros(kk, dtype=np.float64)
            keys = np.zeros((kk, KEY_BYTES), dtype=np.uint8)
            cnt = _i64(0)
            qb = np.zeros(DIM, dtype=np.float64)  # the query is copied here: 0.3 us, where q.ctypes.data costs 2
            buf = bufs[k] = (ids, sc, keys, cnt, _ptr(ids), _ptr(sc), _ptr(keys), ctypes.byref(cnt), qb, _ptr(qb))
        return buf

    def search_shard(self, q: np.ndarray, k: int, since_ms: int = SINCE_NONE, tag_mask: int = 0,
                     stats: Optional[Stats] = None):
        """One query, local top-k with hash keys: (ids, scores, keys[n, 32]).  The latency path of the bench:
        output buffers and their ctypes pointers are kept per k (building them costs more than the call)."""
        k = max(0, int(k))
        if k > 4096:  # never allocate for more results than there are claims
            k = min(k, self.size())
        ids, sc, keys, cnt, p_ids, p_sc, p_keys, p_cnt, qb, p_q = self._out_bufs(k)
        if isinstance(q, np.ndarray) and q.shape == (DIM,) and q.dtype == np.float64:
            np.copyto(qb, q)
        else:
            qb[:] = np.asarray(q, dtype=np.float64).reshape(DIM)
        # the library clamps k to the number of live claims itself
        _check(self._L.lbh_index_search_shard(self._h, p_q, k, _since(since_ms), int(tag_mask), p_ids, p_sc,
                                              p_keys, p_cnt, ctypes.byref(stats) if stats is not None else None))
        n = cnt.value
        return ids[:n].copy(), sc[:n].copy(), keys[:n].copy()
