"""The README defect configuration of the reference (3 replicas, Values = {v1, v2, v3}, StartViewOnTimerLimit = 3; README:13-18,
BASELINE configs[2]): the 24-state counter-example the GPU BFS found on one MI355X (tests/golden/config3_violation.json); its level
table, the probe of level 24 and the violating fingerprint equal the CPU oracle's own run (oracle_levels_config3.json).

CPU: both restatements of the spec accept every step under the recorded action name; AcknowledgedWriteNotLost holds in states
1..23 and fails in state 24; the violating fingerprint is the oracle's; the level counts agree with the oracle's own BFS as deep
as that went.  The reference's own trace for this configuration also has 24 states: BFS depth 24 is the shallowest violation.
GPU (-m gpu): the same walk through vsrmc_model_check_trace."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(GOLDEN, "config3_violation.json")) as f:
        return json.load(f)


def test_config3_counterexample_is_a_behaviour(fx, golden_counts, golden_trace):
    from oracle import orc, pycodec, pyoracle as po
    P = orc.Params(3, 1, 3, 3)
    M = po.Model(3, 1, ("v1", "v2", "v3"), 3)
    recs = [np.array([int(w, 16) for w in t["words"]], dtype=np.uint64) for t in fx["trace"]]
    norm = lambda w: tuple(int(x) for x in orc.normalise(P, w))   # noqa: E731
    assert len(recs) == fx["depth"] == 24 == len(golden_trace["states"])
    assert norm(recs[0]) == norm(orc.init_record(P))
    for i in range(len(recs) - 1):
        hits = [s for s in orc.successors(P, recs[i]) if norm(s["words"]) == norm(recs[i + 1])]
        assert hits and orc.ACTIONS[hits[0]["action"]] == fx["trace"][i + 1]["action"], i
        cur = pycodec.unpack(M, [int(x) for x in recs[i]])
        nxt = pycodec.normalise(M, [int(x) for x in recs[i + 1]])
        names = [n for n, t in po.successors(M, cur) if pycodec.normalise(M, pycodec.pack(M, t)) == nxt]
        assert fx["trace"][i + 1]["action"] in names, i
    assert [orc.invariants(P, r) for r in recs] == [0] * 23 + [1]
    assert not po.AcknowledgedWriteNotLost(M, pycodec.unpack(M, [int(x) for x in recs[-1]]))
    fp, _ = orc.fingerprint(P, recs[-1])
    assert "%016x" % fp == fx["viol_fp"]
    # level counts against the oracle's own BFS of this configuration, as deep as it went
    mine = fx["levels"]
    assert [l["level"] for l in mine] == list(range(1, 24)) and sum(l["n_new"] for l in mine) == fx["distinct_through_level_23"]
    # no level figure of this fixture has a GPU run as its only source any more (round 3: levels 22-23, the probe of level 24 and the
    # violating fingerprint come from the memory-lean oracle driver, tests/golden/oracle_levels_config3.json)
    assert all(l["source"] == "oracle" for l in mine) and fx["probe"]["source"] == "oracle"
    with open(os.path.join(GOLDEN, "oracle_levels_config3.json")) as f:
        o = json.load(f)
    assert len(o["levels"]) == 23 and o["probe"]["viol_fp"] == fx["viol_fp"] and o["probe"]["generated"] == fx["probe"]["generated"]
    assert [(l["new"], l["generated"], l["deadlocks"]) for l in o["levels"]] == [(l["n_new"], l["generated"], l["deadlocks"]) for l in mine]
    g = golden_counts["config3 (3,1,{v1,v2,v3},3)"]
    assert len(g["levels"]) >= 12
    for lv, m in zip(g["levels"], mine):
        assert (lv["new"], lv["generated"], lv["deadlocks"]) == (m["n_new"], m["generated"], m["deadlocks"])


@pytest.mark.gpu
def test_config3_counterexample_on_the_gpu(fx):
    import vsr_tlaplus_amd as vt
    m = vt.Model.from_constants(R=3, C_=1, n=3, L=3)
    recs = [np.array([int(w, 16) for w in t["words"]], dtype=np.uint64) for t in fx["trace"]]
    res = m.check_trace(recs)
    assert res["ok"] and res["actions"] == [t["action"] for t in fx["trace"]] and res["inv_mask_last"] == 1
    fps, _ = m.fingerprints(recs[-1], np.array([0, len(recs[-1])], dtype=np.uint64))
    assert "%016x" % int(fps[0]) == fx["viol_fp"]
