"""The search beyond the record buffers (csrc/vsr_deep.hpp: vsrmc_checker_deepen / _advance / vsrmc_check) against the CPU oracle.

Levels that exist in the seen-set only are regenerated from the newest stored level, slice inside slice; every pass inserts one more
level and probes the one after it.  Held against the oracle level by level — new states, successors in total and per action,
deadlocks, largest bag, xor / sum of the level's fingerprints — for many levels past the base, to exhaustion and to a violation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
M64 = (1 << 64) - 1


def _oracle_levels(orc, P, depth):
    ob = orc.Bfs(P)
    rows = []
    for _ in range(depth):
        n = ob.step()
        if n == 0:
            rows.append(dict(n_new=0))
            break
        fps = ob.level_fps(ob.info["depth"])
        rows.append(dict(level=ob.info["depth"], n_new=n, generated=ob.info["generated"], deadlocks=ob.info["deadlocks"],
                         fp_xor=int(np.bitwise_xor.reduce(fps)), fp_sum=int(fps.astype(object).sum()) & M64, viol=ob.info["viol_mask"]))
        if ob.info["viol_mask"]:
            break
    return rows, ob


def _same(d, want):
    assert (d["level"], d["n_new"], d["generated"], d["deadlocks"]) == (want["level"], want["n_new"], want["generated"], want["deadlocks"]), (d["level"], want)
    assert (d["fp_xor"], d["fp_sum"]) == (want["fp_xor"], want["fp_sum"]), d["level"]


def test_rolling_deep_search_exhausts_a_small_space():
    """(2,1,{v1,v2},2): 2 073 states, 40 levels.  Three stored levels, then everything through the seen-set alone: 36 passes, each
    regenerating every level between the base and the one it inserts; the last pass comes back empty = exhausted."""
    import vsr_tlaplus_amd as vt
    from oracle import orc
    P = orc.Params(2, 1, 2, 2)
    want, _ = _oracle_levels(orc, P, 100)
    m = vt.Model.from_constants(R=2, C_=1, n=2, L=2)
    mc = vt.ModelChecker(m, table_log2=16, frontier_words=1 << 20, frontier_states=1 << 14, pending_entries=1 << 15)
    for _ in range(3):
        d = mc.step()
    probed = {}
    for w in want[3:]:
        a, b = mc.deepen()
        if w["n_new"] == 0:
            assert a["n_new"] == 0 and a["level"] == mc.depth
            break
        _same(a, w)
        assert a["viol_mask"] == 0 and (b is None or b["viol_mask"] == 0)
        if b is not None:
            probed[b["level"]] = b
    assert mc.distinct == 2073 and mc.depth == want[-2]["level"]
    for lvl, b in probed.items():                                # a probed level generates what the inserted level generates a pass later
        w = [x for x in want if x.get("level") == lvl]
        if w:
            assert (b["generated"], b["deadlocks"]) == (w[0]["generated"], w[0]["deadlocks"]), lvl
    with pytest.raises(vt.VsrmcError):
        mc.step()                                                # levels without a frontier: stepping is refused
    mc.reset()
    assert mc.run() == "exhausted" and mc.distinct == 2073       # and the ordinary way round
    mc.close()


@pytest.mark.parametrize("base", [9, 12])
def test_deep_search_finds_the_violation_many_levels_past_the_base(base):
    """(3,1,{v1,v2},1) with AcknowledgedWritesExistOnMajority: violated at depth 19 after 109 878 states.  The base level is 9 / 12;
    levels beyond it are held against the oracle one by one, the violation is found by whichever pass reaches it — as a probed level
    one pass before it would be inserted — and the counter-example is reconstructed through the seen-set."""
    import vsr_tlaplus_amd as vt
    from oracle import orc
    P = orc.Params(3, 1, 2, 1, invariant_mask=2)
    want, ob = _oracle_levels(orc, P, 19)
    assert want[-1]["viol"] == 2 and want[-1]["level"] == 19
    words, off = ob.frontier()
    viol = min(orc.fingerprint(P, words[int(off[i]): int(off[i + 1])])[0] for i in range(len(off) - 1)
               if orc.invariants(P, words[int(off[i]): int(off[i + 1])]))
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=1, invariant_mask=2)
    mc = vt.ModelChecker(m, table_log2=20, frontier_words=1 << 21, frontier_states=1 << 15, pending_entries=1 << 16)
    for _ in range(base - 1):
        mc.step()
    assert mc.level == base
    found = None
    while found is None:
        a, b = mc.deepen()
        if a["viol_mask"]:
            found = a
            break
        _same(a, want[a["level"] - 2])
        if b is not None and b["viol_mask"]:
            found = b
    assert (found["level"], found["viol_mask"], found["viol_fp"]) == (19, 2, viol)
    assert found is b and a["level"] == 18                       # found by the probe of the pass that inserted level 18
    tr = mc.probe_trace()
    assert len(tr) == 19
    rec = orc.init_record(P)
    assert np.array_equal(tr[0][1], rec)
    for act, w in tr[1:]:
        f = orc.fingerprint(P, w)[0]                                # (a bag has no order: states are compared by their view's fingerprint)
        nxt = [s for s in orc.successors(P, rec) if s["fp"] == f]
        assert nxt, "a state of the counter-example is not a successor of its predecessor (%s)" % act
        rec, inv = nxt[0]["words"], nxt[0]["inv"]
    assert inv == 2
    mc.close()


def test_automatic_scheme_needs_no_level_numbers():
    """run() with record buffers that hold config 2 only up to about level 16: the levels are stored while the next one is predicted to
    fit, the search goes on through the seen-set alone and every level has the oracle fixture's figures; stopped at depth 21."""
    import json
    import os
    import vsr_tlaplus_amd as vt
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_levels_config2.json")) as f:
        g = json.load(f)
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
    mc = vt.ModelChecker(m, table_log2=26, frontier_words=1 << 25, frontier_states=1 << 20, pending_entries=1 << 16)
    kinds = []
    while mc.depth < 21:
        kind, a, b = mc.advance()
        kinds.append(kind)
        lv = g["levels"][a["level"] - 1]
        assert (a["n_new"], a["generated"], a["deadlocks"], a["max_bag"]) == (lv["new"], lv["generated"], lv["deadlocks"], lv["max_bag"]), a["level"]
        assert [int(x) for x in a["act_generated"][1:16]] == lv["act_generated"][1:16], a["level"]
        if kind == "deep":
            assert ("%016x" % a["fp_xor"], "%016x" % a["fp_sum"]) == (lv["fp_xor"], lv["fp_sum"]), a["level"]
            if b is not None and b["level"] <= len(g["levels"]):
                assert b["generated"] == g["levels"][b["level"] - 1]["generated"]
    assert "level" in kinds and "deep" in kinds and kinds == sorted(kinds, key=lambda k: k == "deep")   # one switch
    assert 14 <= kinds.index("deep") + 2 <= 19, kinds            # stored as long as it fits (level 16: 838 162 states x 36 words)
    assert mc.distinct == sum(lv["new"] for lv in g["levels"][:21])
    mc.close()


def test_probe2_and_probe3_are_steps_of_the_same_machinery():
    import vsr_tlaplus_amd as vt
    from oracle import orc
    P = orc.Params(3, 1, 2, 2)
    want, _ = _oracle_levels(orc, P, 14)
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
    kw = dict(table_log2=20, frontier_words=1 << 22, frontier_states=1 << 16, pending_entries=1 << 16)
    mc = vt.ModelChecker(m, **kw)
    for _ in range(9):
        mc.step()
    v1, v2, p = mc.probe3()
    _same(v1, want[9])
    _same(v2, want[10])
    assert (p["level"], p["generated"], p["deadlocks"], p["viol_mask"]) == (13, want[11]["generated"], want[11]["deadlocks"], 0)
    a, b = mc.deepen()                                           # ... and on from there
    _same(a, want[11])
    assert b["generated"] == want[12]["generated"]
    mc.close()
    mc = vt.ModelChecker(m, **kw)
    for _ in range(9):
        mc.step()
    v, p = mc.probe2()
    _same(v, want[9])
    assert (p["level"], p["generated"], p["deadlocks"], p["viol_mask"]) == (12, want[10]["generated"], want[10]["deadlocks"], 0)
    mc.close()
