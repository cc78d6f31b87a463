import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_parity as t, test_recent as r
from lbm_amd import _hip
_hip.context()
bad = 0
for seed in range(100, 160):
    for hexh in (True, False):
        try:
            t.test_randomised_incremental_ops_vs_oracle(_hip, seed, hexh)
        except Exception as e:
            bad += 1; print('FAIL ops', seed, hexh, repr(e)[:300])
    try:
        r.test_randomised_graph_vs_oracle(_hip, seed)
    except Exception as e:
        bad += 1; print('FAIL recent', seed, repr(e)[:300])
print('campaign done, failures:', bad)
