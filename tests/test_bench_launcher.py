"""bench.py --gpus N must run N ranks: it starts them itself when no launcher did, refuses a rank count that does
not match --gpus, and rank 0 prints one JSON line with n_gpus = N.  CPU rehearsal over gloo (tests/bench_cpu_harness.py
replaces the rank-local GPU index by the oracle; everything else is bench.py's own code); the same flow with the real
index on one GPU is test_bench_two_ranks_on_one_gpu below."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(REPO, "tests", "bench_cpu_harness.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def _json_line(out: str):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_gpus_flag_launches_that_many_ranks_cpu_rehearsal():
    cmd = [sys.executable, HARNESS, "--gpus", "2", "--claims", "1500", "--steps", "6", "--warmup", "2", "--queries", "4",
           "--cpu-sample", "0"]
    r = subprocess.run(cmd, env=_env(LBM_BENCH_REHEARSE_GLOO="1"), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["config"]["claims_per_gpu"] == 1500 and line["config"]["claims_total"] == 3000
    assert 2 * 1500 * 0.97 <= line["config"]["live_claims_total"] <= 2 * 1500  # both ranks' live claims, ~1 % retracted
    assert "REHEARSAL" in line["data"]


@pytest.mark.timeout(300)
def test_strong_scaling_splits_one_corpus_cpu_rehearsal():
    """--scaling strong: --claims is the TOTAL, dealt out in N contiguous parts; said so on the line."""
    cmd = [sys.executable, HARNESS, "--gpus", "3", "--scaling", "strong", "--claims", "3001", "--steps", "4", "--warmup", "1",
           "--queries", "3", "--cpu-sample", "0"]
    r = subprocess.run(cmd, env=_env(LBM_BENCH_REHEARSE_GLOO="1"), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 3 and line["scaling"] == "strong"
    assert line["config"]["claims_total"] == 3001 and line["config"]["claims_per_gpu"] == 1000  # rank 0's part
    assert 3001 * 0.97 <= line["config"]["live_claims_total"] <= 3001
    assert "in total" in line["config"]["workload"] and "strong scaling" in line["config"]["workload"]


@pytest.mark.timeout(120)
def test_rank_count_must_match_gpus_flag():
    cmd = [sys.executable, HARNESS, "--gpus", "4", "--claims", "500", "--steps", "2", "--warmup", "1", "--cpu-sample", "0"]
    r = subprocess.run(cmd, env=_env(LBM_BENCH_REHEARSE_GLOO="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=100)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_two_ranks_on_one_gpu():
    """The real bench (HIP index, K1 ingest, search_shard) with two ranks sharing GPU 0, exchange over gloo."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--claims", "300000", "--steps", "20", "--warmup", "5",
           "--cpu-sample", "0"]
    r = subprocess.run(cmd, env=_env(LBM_BENCH_REHEARSE_GLOO="1"), capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["claims_per_gpu"] == 300000 and line["value"] > 0
    assert line["roofline"]["kernel"] == "scan_f32_kernel" and line["roofline"]["achieved"] > 0
