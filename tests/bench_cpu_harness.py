"""CPU rehearsal of bench.py's launcher and N > 1 flow (test infrastructure, run by tests/test_bench_launcher.py).

bench.py itself has no CPU path: without a GPU ``lbm_amd._hip.Index()`` raises.  To exercise what does NOT need a GPU
-- ``--gpus N`` spawning N ranks, the gloo rendezvous, barriers, max-over-ranks timing, summed claim counts, one JSON
line from rank 0, the failure when WORLD_SIZE and --gpus disagree -- this script swaps the rank-local index for the
CPU oracle (tests may use it) and then runs bench.main() unchanged.  The numbers it prints mean nothing."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from lbm_amd import _hip  # noqa: E402
from oracle import lb_oracle as orc  # noqa: E402
from tests.oracle_index import OracleIndex  # noqa: E402


_hip.Index = OracleIndex
_hip.embed_batch = orc.embed_batch
_hip.set_option = lambda *a: None

if __name__ == "__main__":
    bench.main()
