"""The multi-threaded oracle driver (oracle/vsr_oracle_mt, used by bench.py's cpu_baseline leg) against the single-threaded
oracle and its committed per-level fixtures: identical new / generated / deadlock counts and fingerprint xor / sum per level for any thread count."""
import json
import subprocess

import pytest

from oracle import orc


def _run(args):
    orc.build()
    out = subprocess.run([orc.BIN_MT] + [str(a) for a in args], capture_output=True, text=True, check=True).stdout
    lines = [json.loads(l) for l in out.strip().splitlines()]
    return lines[:-1], lines[-1]


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_mt_oracle_reproduces_the_level_fixtures(golden_counts, threads):
    g = golden_counts["config2 (3,1,{v1,v2},2)"]
    levels, summary = _run([3, 1, 2, 2, "--threads", threads, "--max-depth", 12])
    assert summary["threads"] == threads and summary["stop"] == "max-depth" and summary["error"] == ""
    assert len(levels) == 12
    for lv, want in zip(levels, g["levels"]):
        assert (lv["level"], lv["new"], lv["generated"], lv["deadlocks"], lv["ties"], lv["fp_xor"], lv["fp_sum"]) == \
               (want["level"], want["new"], want["generated"], want["deadlocks"], 0, want["fp_xor"], want["fp_sum"])
        assert sum(lv["act_generated"]) == lv["generated"]
    assert summary["distinct"] == sum(w["new"] for w in g["levels"][:12])


def test_mt_oracle_whole_space_and_eval_error():
    _, s = _run([2, 1, 2, 2, "--threads", 4])
    assert (s["stop"], s["distinct"], s["depth"], s["viol_mask"]) == ("exhausted", 2073, 27, 0)
    _, s = _run([2, 1, 2, 2, "--threads", 4, "--no-symmetry"])
    assert s["distinct"] == 4034
    _, s = _run([3, 2, 3, 3, "--threads", 2])                         # VSR.tla:421 with two clients
    assert s["stop"] == "error" and "VSR.tla:421" in s["error"] and s["depth"] == 2
