import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_embed():
    return load_golden("embed.json")["data"]


@pytest.fixture(scope="session")
def golden_compile():
    return load_golden("compile.json")["data"]


@pytest.fixture(scope="session")
def golden_synth():
    return load_golden("synth2k.json")["data"]


@pytest.fixture(scope="session")
def golden_snapshot():
    return load_golden("snapshot.json")["data"]


@pytest.fixture(scope="session")
def golden_embed_dim():
    return load_golden("embed_dim.json")["data"]
