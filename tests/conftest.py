"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU needed).
`-m gpu`       : parity tests proper — HIP path (through the C-ABI) vs the CPU oracle / golden fixtures.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def golden_trace():
    with open(os.path.join(GOLDEN, "state_transfer_trace.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_counts():
    with open(os.path.join(GOLDEN, "bfs_counts.json")) as f:
        return {c["label"]: c for c in json.load(f)}
