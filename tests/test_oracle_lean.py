"""The memory-lean oracle driver (oracle/vsr_oracle_lean: levels beyond a stored base level are regenerated from the base level's
records; its fingerprint works on the encoded record and is checked against fingerprint() of vsr_oracle.cpp) against the ordinary
multi-threaded driver: identical per-level figures, identical violating fingerprint, for any base level and thread count; the probe
pass (nothing inserted) finds the same violation as the level it stands in for."""
import json
import os
import subprocess

import pytest

from oracle import orc

LEAN = os.path.join(os.path.dirname(orc.BIN_MT), "vsr_oracle_lean")
KEYS = ("level", "new", "generated", "ties", "deadlocks", "distinct", "max_bag", "fp_xor", "fp_sum", "act_generated")


def _run(exe, args):
    orc.build()
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, check=True).stdout
    return [json.loads(l) for l in out.strip().splitlines()]


@pytest.mark.parametrize("base,threads", [(3, 1), (7, 4), (11, 8)])
def test_lean_levels_equal_the_ordinary_driver(base, threads):
    want = _run(orc.BIN_MT, [3, 1, 2, 2, "--threads", 4, "--max-depth", 13])
    got = _run(LEAN, [3, 1, 2, 2, "--threads", threads, "--max-depth", 13, "--base-level", base, "--slots", 1 << 21, "--verify-fp-all"])
    assert len(got) == len(want) == 14
    for g, w in zip(got[:-1], want[:-1]):
        assert {k: g[k] for k in KEYS} == {k: w[k] for k in KEYS}
    assert (got[-1]["distinct"], got[-1]["generated"], got[-1]["stop"]) == (want[-1]["distinct"], want[-1]["generated"], "max-depth")
    assert got[-1]["base_level"] == base and got[-1]["error"] == ""


def test_lean_with_symmetry_of_three_values_and_whole_space():
    # three values = six permutations: the encoded-record fingerprint against fingerprint() on every successor
    want = _run(orc.BIN_MT, [3, 1, 3, 3, "--threads", 4, "--max-depth", 10])
    got = _run(LEAN, [3, 1, 3, 3, "--threads", 3, "--max-depth", 10, "--base-level", 6, "--slots", 1 << 20, "--verify-fp-all"])
    for g, w in zip(got[:-1], want[:-1]):
        assert {k: g[k] for k in KEYS} == {k: w[k] for k in KEYS}
    # a whole small space: the search ends by exhaustion inside the regenerated levels
    got = _run(LEAN, [2, 1, 2, 2, "--threads", 2, "--base-level", 9, "--slots", 1 << 16, "--verify-fp-all"])
    assert (got[-1]["stop"], got[-1]["distinct"], got[-1]["depth"], got[-1]["viol_mask"]) == ("exhausted", 2073, 27, 0)


def test_lean_violation_and_probe_pass():
    # (3,1,{v1,v2},1) violates AcknowledgedWritesExistOnMajority (inv bit 1) at depth 19 after 146 935 states: the ordinary driver's
    # violating fingerprint must come out of the lean driver both when level 19 is inserted and when it is only probed
    cfg = [3, 1, 2, 1]
    want = _run(orc.BIN_MT, cfg + ["--threads", 2, "--inv-mask", 2])
    s = want[-1]
    assert (s["stop"], s["depth"], s["distinct"]) == ("violation", 19, 146935)
    depth = s["depth"]
    got = _run(LEAN, cfg + ["--threads", 3, "--inv-mask", 2, "--base-level", 14, "--slots", 1 << 19, "--verify-fp-all"])
    assert (got[-1]["stop"], got[-1]["depth"], got[-1]["viol_fp"], got[-1]["distinct"]) == ("violation", depth, s["viol_fp"], s["distinct"])
    got = _run(LEAN, cfg + ["--threads", 3, "--inv-mask", 2, "--base-level", 14, "--slots", 1 << 19, "--probe-level", depth])
    probe = [g for g in got if "probe_level" in g][0]
    assert (probe["probe_level"], probe["viol_fp"], probe["viol_mask"]) == (depth, s["viol_fp"], 2) and probe["violating_successors"] >= 1
    assert probe["generated"] == want[-2]["generated"] and probe["deadlocks"] == want[-2]["deadlocks"]
    assert (got[-1]["stop"], got[-1]["depth"], got[-1]["viol_fp"]) == ("violation", depth, s["viol_fp"])


def test_collision_hunt_on_a_shortened_audit_fingerprint():
    # the second model's lean driver with a 12-bit audit fingerprint (--hunt-mask): collisions galore.  Every reported state must really have
    # the audit fingerprint it is reported under (recomputed through the oracle's binding, seed = the audit seed), the members printed for the
    # first collision are different states, and the hunt changes nothing in the level figures
    import numpy as np
    from oracle import orc2
    exe = os.path.join(os.path.dirname(orc.BIN_MT), "vrst_oracle_lean")
    args = [3, 1, 2, 2, "--base-level", 6, "--slots", 1 << 19, "--max-depth", 10, "--threads", 3, "--inv-mask", 14, "--no-symmetry"]
    plain = _run(exe, args)
    got = _run(exe, args + ["--hunt-seed", "5eed5eed5eed5eed", "--hunt-slots", 1 << 18, "--hunt-mask", "fff"])
    levels = [g for g in got if "level" in g and "words" not in g]
    assert [{k: g[k] for k in KEYS} for g in levels] == [{k: w[k] for k in KEYS} for w in plain[:-1]]
    col = [g for g in got if g.get("fp_collision")]
    mem = [g for g in got if g.get("fp_collision_member")]
    assert got[-1]["audit_collisions"] == len(col) > 100 and mem
    first = col[0]["audit_fp"]
    assert {g["audit_fp"] for g in mem} == {first}
    assert len({tuple(g["words"]) for g in mem}) >= 2
    P = orc2.Params(3, 2, 2, invariant_mask=14)
    orc2.set_fp_seed(0x5EED5EED5EED5EED)
    try:
        for g in col[:200] + mem:
            rec = np.array([int(w, 16) for w in g["words"]], dtype=np.uint64)
            assert (orc2.fingerprint(P, rec)[0] & 0xFFF) | 1 == int(g["audit_fp"], 16)
        orc2.set_fp_seed(0)
        for g in col[:200]:                                      # "fp" = the run's own fingerprint of the state (seed 0 here)
            rec = np.array([int(w, 16) for w in g["words"]], dtype=np.uint64)
            assert orc2.fingerprint(P, rec)[0] == int(g["fp"], 16)
    finally:
        orc2.set_fp_seed(0)
