"""CPU-only checks: the C-ABI library loads, exports every symbol include/lbhip.h declares, refuses to
run without a GPU (no silent CPU path), and the host-side mirror logic that needs no device."""
import os
import re

import numpy as np
import pytest

from lbm_amd import _hip, synth
from lbm_amd.context_graph import Claim, ContextGraph, _ClaimDict
from lbm_amd.latent import _tokenize

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "lbhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lbh_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    L = _hip.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/lbhip.h but not exported"
    assert sorted(_hip.EXPORTS) == names
    assert L.lbh_version() == 1


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_no_cpu_fallback():
    with pytest.raises(_hip.HipBackendError, match="no CPU fallback|no HIP device"):
        _hip.embed_batch(["hello"])
    g = ContextGraph()
    g.add_claim("h1", "some text", [], created_ms=1)
    with pytest.raises(_hip.HipBackendError):
        g.compile("text")
    with pytest.raises(_hip.HipBackendError):
        ContextGraph().compile("query")  # even an empty graph opens the backend


def test_tokenize_mirror_matches_survey_examples():
    assert _tokenize("Héllo wörld_1 FOO-bar 42") == ["h", "llo", "w", "rld_1", "foo", "bar", "42"]
    assert _tokenize("Hello_World 42!x") == ["hello_world", "42", "x"]
    assert _tokenize("") == []


def test_claim_dataclass_roundtrip_like_reference():
    c = Claim("h", "t", ["a"], 5, ["e"])
    d = c.to_dict()
    assert d == {"claim_hash": "h", "text": "t", "tags": ["a"], "created_ms": 5, "evidence": ["e"], "retracted": False}
    c2 = Claim.from_dict({**d, "parent_hash": "p"})
    assert c2.parent_hash == "p" and c2.to_dict()["parent_hash"] == "p"
    assert Claim.from_dict({"claim_hash": 7}).claim_hash == "7"
    assert c == Claim("h", "t", ["a"], 5, ["e"])


def test_claims_dict_tracks_mutations_without_a_gpu():
    g = ContextGraph()
    assert isinstance(g.claims, dict) and isinstance(g.claims, _ClaimDict)
    g.add_claim("a" * 64, "alpha", ["x"], created_ms=1)
    g.claims["raw"] = Claim("raw", "inserted directly", [], 2, [])
    assert list(g._dirty) == ["a" * 64, "raw"]
    g._dirty.clear()
    g.claims["raw"].retracted = True
    assert list(g._dirty) == ["raw"]
    g._dirty.clear()
    g.claims["raw"].created_ms = 9
    del g.claims["a" * 64]
    assert list(g._dirty) == ["raw", "a" * 64]
    snap = g.snapshot()
    assert snap == {"claims": {"raw": g.claims["raw"].to_dict()}}
    g2 = ContextGraph.from_snapshot(snap)
    assert list(g2.claims) == ["raw"] and g2.claims["raw"].retracted


def test_synth_generator_is_deterministic_and_well_formed():
    a = synth.make_chunk(0, 500, seed=1234)
    b = synth.make_chunk(0, 500, seed=1234)
    assert np.array_equal(a.blob, b.blob) and np.array_equal(a.keys, b.keys)
    for i in (0, 17, 499):
        toks = a.text(i).split(" ")
        assert 8 <= len(toks) <= 24 and all(re.fullmatch(r"w\d{5}", t) for t in toks)
        assert len(a.hash_hex(i)) == 64 and len(a.tags(i)) <= 3
    assert a.created_ms[3] == synth.BASE_MS + 3000
    qs = synth.make_queries(5)
    assert all(3 <= len(q.split()) <= 6 for q in qs) and qs == synth.make_queries(5)


def test_env_knobs_device_list_and_backend_choice(monkeypatch):
    """LB_HIP_DEVICES / LB_RETRIEVAL_BACKEND (the reference's LB_* style): host logic only."""
    for var in ("LB_HIP_DEVICES", "LBM_HIP_DEVICE", "LOCAL_RANK", "LB_RETRIEVAL_BACKEND"):
        monkeypatch.delenv(var, raising=False)
    assert _hip.default_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert _hip.default_device() == 3
    monkeypatch.setenv("LB_HIP_DEVICES", "4, 5,6,7")
    assert _hip.default_device() == 7
    monkeypatch.setenv("LOCAL_RANK", "4")
    with pytest.raises(_hip.HipBackendError, match="lists 4 GPUs"):
        _hip.default_device()
    monkeypatch.delenv("LOCAL_RANK")
    assert _hip.default_device() == 4
    monkeypatch.setenv("LB_HIP_DEVICES", "0,x")
    with pytest.raises(_hip.HipBackendError, match="comma-separated"):
        _hip.default_device()
    _hip.check_backend_choice()
    for ok in ("auto", "hip", "HIP"):
        monkeypatch.setenv("LB_RETRIEVAL_BACKEND", ok)
        _hip.check_backend_choice()
    monkeypatch.setenv("LB_RETRIEVAL_BACKEND", "python")
    with pytest.raises(_hip.HipBackendError, match="no CPU path"):
        _hip.check_backend_choice()
    monkeypatch.setenv("LB_RETRIEVAL_BACKEND", "cuda")
    with pytest.raises(_hip.HipBackendError, match="expected auto, hip or python"):
        _hip.check_backend_choice()
