"""lbh_search_dev_noalloc (the allocation-free entry point; no counterpart in the reference) against the entry point
it is a variant of, lbh_index_search_shard: same kernels, scratch in the library's file-scope device array instead of
the index's hipMalloc'd workspace.  Bit-exact rows and fp64 scores, on two calls in a row, the second with a smaller
input -- the shared array keeps the first call's contents, so whatever must start at zero has to be cleared."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corpus(hip, torch, rng, n):
    v = rng.standard_normal((n, 64))
    v[1::40] = v[0::40][: v[1::40].shape[0]]  # exact duplicates: ties broken by key
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    ms = rng.integers(0, 10_000, size=n).astype(np.int64)
    tags = rng.integers(0, 2 ** 63, size=n).astype(np.uint64)
    keys = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ix = hip.Index()
    ids = ix.add_vectors(v, ms, tags, keys)
    assert ids.tolist() == list(range(n))  # no retracts: claim id == row
    dev = torch.device("cuda", 0)
    t64 = torch.from_numpy(v).to(dev)
    words = keys.view(">u8").astype(np.uint64).view(np.int64).reshape(n, 4)  # word w = big-endian value of key bytes 8w..8w+7
    cols = {"e64": t64, "e32": t64.to(torch.float32), "ms": torch.from_numpy(ms).to(dev),
            "tags": torch.from_numpy(tags.view(np.int64)).to(dev), "keys": torch.from_numpy(words.copy()).to(dev)}
    return ix, cols, (v, ms, tags, keys)


def _noalloc(hip, torch, cols, n, q, k, since, tm, out):
    rows, scores, count = out
    count.fill_(-7)
    torch.cuda.synchronize()  # the call runs on the library's own stream: nothing else orders this fill ahead of it
    hip.search_dev_noalloc(cols["e32"].data_ptr(), cols["e64"].data_ptr(), cols["ms"].data_ptr(), cols["tags"].data_ptr(),
                           cols["keys"].data_ptr(), n, 1.0, q, k, since, tm, rows.data_ptr(), scores.data_ptr(), count.data_ptr())
    c = int(count.item())
    return rows[:max(c, 0)].cpu().numpy(), scores[:max(c, 0)].cpu().numpy(), c


def test_noalloc_entry_point_equals_search_shard():
    import torch
    from lbm_amd import _hip as hip
    rng = np.random.default_rng(11)
    dev = torch.device("cuda", 0)
    out = (torch.zeros(64, dtype=torch.int64, device=dev), torch.zeros(64, dtype=torch.float64, device=dev),
           torch.zeros(1, dtype=torch.int64, device=dev))
    qs = rng.standard_normal((5, 64))
    qs /= np.linalg.norm(qs, axis=1, keepdims=True)
    # two calls in a row, the second on a smaller input; then a large one again, then a tiny one
    for n in (300_000, 4_000, 120_000, 70):
        ix, cols, (v, ms, tags, keys) = _corpus(hip, torch, rng, n)
        st = hip.Stats()
        if n <= 4_000:  # and once against the CPU oracle itself, so that this file stands on its own
            from oracle import lb_oracle as orc
            hexes = [bytes(r).hex() for r in keys]
            for q in (qs[0], np.zeros(64)):
                for k, since, tm in ((8, None, 0), (64, 5000, 0xFF)):
                    oi, osc = orc.compile_arrays(v, ms, np.zeros(n, np.uint8), hexes, q, k, since, tags, tm)
                    gi, gs, c = _noalloc(hip, torch, cols, n, q, k, hip.SINCE_NONE if since is None else since, tm, out)
                    assert c == len(oi) and gi.tolist() == oi.tolist() and gs.tobytes() == osc.tobytes(), (n, k, since, tm)
        for q in list(qs) + [np.zeros(64)]:  # the zero query ties every row at 0.0: the exact fp64 path must take over
            for k, since, tm in ((8, hip.SINCE_NONE, 0), (64, 5000, 0), (8, hip.SINCE_NONE, 0xFF), (1, 9000, 0xF0F0), (0, hip.SINCE_NONE, 0)):
                wi, ws, _ = ix.search_shard(q, k, since, tm, st)
                gi, gs, c = _noalloc(hip, torch, cols, n, q, k, since, tm, out)
                assert c == len(wi), (n, k, since, tm)
                assert gi.tolist() == wi.tolist(), (n, k, since, tm)
                assert gs.tobytes() == ws.tobytes(), (n, k, since, tm)
                if not q.any() and k > 0 and n > 1000 and since == hip.SINCE_NONE and tm == 0:
                    assert st.path == 1 and st.fallbacks == 1  # the reference entry point took the exact path too
        ix.close()
    # limits are rejected with a clear error, nothing is paged or allocated
    with pytest.raises(hip.HipBackendError, match="top_k <= 64"):
        hip.search_dev_noalloc(cols["e32"].data_ptr(), cols["e64"].data_ptr(), 0, 0, cols["keys"].data_ptr(), 70, 1.0, qs[0], 65,
                               hip.SINCE_NONE, 0, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    with pytest.raises(hip.HipBackendError, match="at most"):
        hip.search_dev_noalloc(cols["e32"].data_ptr(), cols["e64"].data_ptr(), 0, 0, cols["keys"].data_ptr(), 2 ** 32, 1.0, qs[0], 8,
                               hip.SINCE_NONE, 0, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    with pytest.raises(hip.HipBackendError, match="filter needs its column"):
        hip.search_dev_noalloc(cols["e32"].data_ptr(), cols["e64"].data_ptr(), 0, 0, cols["keys"].data_ptr(), 70, 1.0, qs[0], 8,
                               123, 0, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
