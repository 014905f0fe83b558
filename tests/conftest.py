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


FP_VERSION = 2     # fingerprint function of this build (oracle/vsr_oracle.hpp FP_VERSION)


@pytest.fixture(scope="session")
def oracle_levels():
    """Whole-workload fixtures written by the CPU oracle (tools/make_oracle_levels.py, run on the GPU box's host cores):
    label -> fixture.  `checksums` says whether its fingerprint values were made with this build's fingerprint function
    (the counts are valid for any)."""
    import glob
    out = {}
    for path in sorted(glob.glob(os.path.join(GOLDEN, "oracle_levels_*.json"))):
        with open(path) as f:
            d = json.load(f)
        d["checksums"] = d.get("fp_version", 1) == FP_VERSION
        d["file"] = os.path.basename(path)
        key = d["label"].split(" ")[0]
        if key not in out or d["checksums"]:
            out[key] = d
    return out
