"""Oracle-backed stand-in for ``lbm_amd._hip.Index`` (TEST INFRASTRUCTURE: lets the CPU tests drive the host-side logic
that sits above the C ABI -- the claim-sharded graph, bench.py's launcher -- on a box without a GPU).  Same method
names and return shapes as the real class; every number comes from oracle/lb_oracle.c."""
import numpy as np

from lbm_amd import _hip
from oracle import lb_oracle as orc


class OracleIndex:
    def __init__(self):
        self.emb = np.zeros((0, 64))
        self.created = np.zeros(0, dtype=np.int64)
        self.tags = np.zeros(0, dtype=np.uint64)
        self.keys = np.zeros((0, 32), dtype=np.uint8)
        self.live = np.zeros(0, dtype=bool)
        self.order = np.zeros(0, dtype=np.uint64)

    def reserve(self, n):
        pass

    def close(self):
        pass

    def _append(self, emb, created_ms, tagmask, keys):
        first = self.emb.shape[0]
        n = emb.shape[0]
        self.emb = np.concatenate([self.emb, emb])
        self.created = np.concatenate([self.created, np.asarray(created_ms, dtype=np.int64)])
        self.tags = np.concatenate([self.tags, np.zeros(n, np.uint64) if tagmask is None else np.asarray(tagmask, dtype=np.uint64)])
        self.keys = np.concatenate([self.keys, np.asarray(keys, dtype=np.uint8).reshape(-1, 32)])
        self.live = np.concatenate([self.live, np.ones(n, dtype=bool)])
        self.order = np.concatenate([self.order, np.arange(first, first + n, dtype=np.uint64)])  # default: the claim id
        return np.arange(first, first + n, dtype=np.uint64)

    def add_texts(self, blob, offsets, created_ms, tagmask, keys):
        return self._append(orc.embed_batch_packed(blob, offsets), created_ms, tagmask, keys)

    def add_vectors(self, emb, created_ms, tagmask, keys):
        return self._append(np.asarray(emb, dtype=np.float64).reshape(-1, 64), created_ms, tagmask, keys)

    def retract_many(self, ids):
        self.live[np.asarray(ids, dtype=np.int64)] = False

    def retract(self, cid):
        self.live[int(cid)] = False

    def update_meta(self, cid, created_ms, tagmask, key):
        self.created[int(cid)], self.tags[int(cid)] = created_ms, tagmask
        self.keys[int(cid)] = np.frombuffer(key, dtype=np.uint8)

    def set_keys(self, ids, keys):
        self.keys[np.asarray(ids, dtype=np.int64)] = np.asarray(keys, dtype=np.uint8).reshape(-1, 32)

    def set_order(self, ids, order):
        self.order[np.asarray(ids, dtype=np.int64)] = np.asarray(order, dtype=np.uint64)

    def recent(self, since_ms=_hip.SINCE_NONE, limit=100, stats=None):
        """lbh_index_recent: live claims with created_ms >= since_ms, newest first, ties in ascending order stamp."""
        ok = self.live & (self.created >= max(int(since_ms), _hip.SINCE_NONE))
        idx = np.nonzero(ok)[0]
        idx = idx[np.lexsort((self.order[idx], -self.created[idx]))][: max(0, int(limit))]
        return idx.astype(np.uint64), self.created[idx].copy(), int(ok.sum())

    def get_embedding(self, cid):
        return self.emb[int(cid)].copy()

    def get_embeddings(self, ids):
        return self.emb[np.asarray(ids, dtype=np.int64)].copy()

    def size(self):
        return int(self.live.sum())

    def search_shard(self, q, k, since_ms=_hip.SINCE_NONE, tag_mask=0, stats=None):
        hexes = [bytes(r).hex() for r in self.keys]  # 64 hex chars order like the 32 key bytes
        idx, sc = orc.compile_arrays(self.emb, self.created, (~self.live).astype(np.uint8), hexes, q, k,
                                     None if since_ms == _hip.SINCE_NONE else since_ms, self.tags, tag_mask)
        if stats is not None:
            stats.scan_ms = 1e-3
        return idx.astype(np.uint64), sc, self.keys[idx]
