"""Every 64-row tile of the matrix is scanned exactly once, whatever way the single-query scan divides the tiles
among its blocks (contiguous weighted ranges, `lbh_split` in lbm_amd/csrc/search.hip, or the grid-stride loop below
two tiles per wave).  Sizes sit on the switches: grid-stride <-> ranges (2 x 256 blocks x 4 waves = 2 048 tiles =
131 072 rows on a 256-CU part), one <-> two blocks per CU (3 500 000 rows), ragged last tiles.

Method: one "needle" row per tile, at a random position inside it, equal to one of the query vectors (score exactly
1.0, far above the 0.01-norm background); tile t belongs to query t mod nq, nq chosen so that a query owns at most
64 tiles.  The top-64 of query j must then be exactly its needles, ordered by key (all scores tie at 1.0): a tile that
no wave visits loses its needle, a tile visited twice would return its needle twice."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [131_007, 131_072, 131_073, 131_136, 262_207, 1_000_017, 3_500_000, 3_500_001, 3_500_127]


@pytest.mark.parametrize("n", SIZES)
def test_every_tile_is_scanned_once(n):
    from lbm_amd import _hip as hip
    rng = np.random.default_rng(n)
    tiles = (n + 63) // 64
    nq = (tiles + 63) // 64
    q = rng.standard_normal((nq, 64))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    keys = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    ix = hip.Index()
    ix.reserve(n)
    needle_rows = np.empty(tiles, dtype=np.int64)
    chunk = 64 * 8192  # whole tiles per chunk
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        v = 0.01 * rng.standard_normal((m, 64)) / 8.0  # |x| ~ 0.01: no background row comes near a needle
        for t in range(s // 64, (s + m + 63) // 64):
            r = int(rng.integers(t * 64, min(t * 64 + 64, n)))
            needle_rows[t] = r
            v[r - s] = q[t % nq]
        ix.add_vectors(v, np.arange(s, s + m, dtype=np.int64), None, keys[s:s + m])
    st = hip.Stats()
    for j in range(nq):
        mine = needle_rows[j::nq]
        ids, sc, _ = ix.search_shard(q[j], 64, hip.SINCE_NONE, 0, st)
        assert st.path == 0 and st.fallbacks == 0  # the fp32 scan + exact re-rank, no exact-path rescue
        want = sorted(mine.tolist(), key=lambda r: keys[r].tobytes())
        assert ids[: len(mine)].tolist() == want, (n, j)
        assert np.all(sc[: len(mine)] == sc[0]) and abs(sc[0] - 1.0) < 1e-12
        if len(mine) < 64 and len(ids) > len(mine):
            assert sc[len(mine)] < 0.999  # the next best is background or another query's needle
    ix.close()
