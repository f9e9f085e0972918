import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the CPU checker is test infrastructure

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scenarios():
    with open(os.path.join(GOLDEN, "scenarios.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def sparse_goldens():
    with open(os.path.join(GOLDEN, "sparse_graphs.json")) as f:
        return json.load(f)["graphs"]


@pytest.fixture(scope="session")
def threshold_kats():
    with open(os.path.join(GOLDEN, "threshold_kats.json")) as f:
        return json.load(f)["kats"]
