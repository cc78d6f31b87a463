"""Graph-level randomised campaign beyond the seeds the test suite runs (GPU box, repo root): incremental-ops and
recency tests over 60 more seeds, hex and free-form hashes, single-query scan over the fp32 matrix and the fp16 copy."""
import sys
import traceback
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_parity as t, test_recent as r
from lbm_amd import _hip
_hip.context()
bad = 0
base = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for seed in range(base, base + count):
    for hexh in (True, False):
        try:
            t.test_randomised_incremental_ops_vs_oracle(_hip, seed, hexh, seed & 1)
        except Exception as e:
            bad += 1; print('FAIL ops', seed, hexh, repr(e)[:300], '|', traceback.format_exc().strip().splitlines()[-2].strip())
    try:
        r.test_randomised_graph_vs_oracle(_hip, seed)
    except Exception as e:
        bad += 1; print('FAIL recent', seed, repr(e)[:300], '|', traceback.format_exc().strip().splitlines()[-2].strip())
print('campaign done, failures:', bad)
