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


@pytest.mark.parametrize("base,probe_env", [(9, None), (12, None), (9, ("VSRMC_PROBE_CCAP", "32")), (11, ("VSRMC_NO_PROBE_KERNEL", "1"))])
def test_deep_search_finds_the_violation_many_levels_past_the_base(base, probe_env, monkeypatch):
    """(3,1,{v1,v2},1) with AcknowledgedWritesExistOnMajority: violated at depth 19 after 109 878 states.  The base level is 9 / 12;
    levels beyond it are held against the oracle one by one, the violation is found by whichever pass reaches it — as a probed level
    one pass before it would be inserted — and the counter-example is reconstructed through the seen-set.  The probe passes run the
    probe-only instantiation (k_expand<.., 6> + k_probe_resolve, round 6); once with a work list of 32 entries, so that its tiles
    overflow and are taken again in halves, and once without it (the mode-capable kernel, rounds 4-5)."""
    import vsr_tlaplus_amd as vt
    from oracle import orc
    if probe_env:
        monkeypatch.setenv(*probe_env)
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


# ---------------------------------------------------------------------------------------------------------------------
# round 5: no fatal mispredictions, re-basing, checkpoints of a deep search
# ---------------------------------------------------------------------------------------------------------------------
def _fixture(key):
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_levels_%s.json" % key)) as f:
        return json.load(f)


@pytest.mark.parametrize("words_log2", [17, 19, 21])
def test_a_level_that_overflows_its_record_buffer_is_kept_in_the_seen_set(words_log2):
    """A wrong "it fits" is not fatal any more.  Config 2 with record buffers of 2^17 / 2^19 / 2^21 words, stepped BLINDLY (vsrmc_checker_step, no
    prediction) until a level runs out of buffer: the step reports "frontier full" as it always did — and the next advance() adopts that very
    level from the seen-set (k_expand went on claiming, counting and checking after the buffers were exhausted): its size, successors in total
    and per action, deadlocks, largest bag and both fingerprint checksums are the oracle fixture's, and so is every level the search then
    reaches through the seen-set alone."""
    import vsr_tlaplus_amd as vt
    g = _fixture("config2")
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
    mc = vt.ModelChecker(m, table_log2=24, frontier_words=1 << words_log2, frontier_states=1 << (words_log2 - 4), pending_entries=1 << 16)
    failed_at = None
    for _ in range(20):
        try:
            d = mc.step()
        except vt.VsrmcError as e:
            assert "frontier" in str(e).lower() or "device error 21" in str(e), str(e)
            failed_at = mc.level + 1
            break
        lv = g["levels"][d["level"] - 1]
        assert (d["n_new"], d["generated"]) == (lv["new"], lv["generated"])
    assert failed_at is not None and 8 <= failed_at <= 19, failed_at
    kind, a, b = mc.advance()                                    # the overflowed level, adopted
    assert kind == "deep" and b is None and a["level"] == failed_at
    seen = 0
    while True:
        lv = g["levels"][a["level"] - 1]
        assert (a["n_new"], a["generated"], a["deadlocks"], a["max_bag"]) == (lv["new"], lv["generated"], lv["deadlocks"], lv["max_bag"]), a["level"]
        assert [int(x) for x in a["act_generated"][1:16]] == lv["act_generated"][1:16], a["level"]
        assert ("%016x" % a["fp_xor"], "%016x" % a["fp_sum"]) == (lv["fp_xor"], lv["fp_sum"]), a["level"]
        seen += 1
        if seen == 4:
            break
        kind, a, b = mc.advance()
        assert kind == "deep"
    assert mc.depth == failed_at + 3 and mc.distinct == sum(x["new"] for x in g["levels"][: failed_at + 3])
    mc.close()


def test_overflow_inside_the_automatic_scheme_and_a_seen_set_that_grows():
    """The whole loop with both mispredictions forced: a seen-set of 2^10 slots for a search of 109 878 states (it is re-hashed into twice the
    slots again and again, never "incomplete" while the device has memory), record buffers the levels outgrow, to the violation of
    AcknowledgedWritesExistOnMajority at depth 19 of (3,1,{v1,v2},1) — the oracle's level figures all the way, the oracle's violating
    fingerprint, and a counter-example that still walks back to Init through the re-hashed table."""
    import vsr_tlaplus_amd as vt
    from oracle import orc
    P = orc.Params(3, 1, 2, 1, invariant_mask=2)
    want, ob = _oracle_levels(orc, P, 19)
    words, off = ob.frontier()
    viol = min(orc.fingerprint(P, words[int(off[i]): int(off[i + 1])])[0] for i in range(len(off) - 1)
               if orc.invariants(P, words[int(off[i]): int(off[i + 1])]))
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=1, invariant_mask=2)
    mc = vt.ModelChecker(m, table_log2=10, frontier_words=1 << 18, frontier_states=1 << 14, pending_entries=1 << 16)
    grown = 0
    while mc.violation is None:
        st = mc.room()
        assert st != 2, "the seen-set must grow, not give up, while the device has memory"
        grown += st == 1
        kind, a, b = mc.advance()
        if a["viol_mask"]:
            break
        _same(a, want[a["level"] - 2]) if kind == "deep" else None
        w = want[a["level"] - 2]
        assert (a["n_new"], a["generated"], a["deadlocks"]) == (w["n_new"], w["generated"], w["deadlocks"]), a["level"]
    assert grown >= 2 and int(mc.options.table_log2) >= 17, (grown, int(mc.options.table_log2))   # (one call may double the table more than once)
    assert mc.violation["mask"] == 2 and mc.violation["fp"] == viol and mc.violation["level"] == 19
    tr = mc.violation_trace()
    assert len(tr) == 19 and np.array_equal(tr[0][1], orc.init_record(P))
    rec = tr[0][1]
    for act, w in tr[1:]:
        f = orc.fingerprint(P, w)[0]
        nxt = [s for s in orc.successors(P, rec) if s["fp"] == f]
        assert nxt, act
        rec = nxt[0]["words"]
    mc.close()
    # and the native loop: vsrmc_check on the same sizes
    mc = vt.ModelChecker(m, table_log2=10, frontier_words=1 << 18, frontier_states=1 << 14, pending_entries=1 << 16)
    assert mc.check() == "violation" and mc.violation["fp"] == viol and int(mc.options.table_log2) >= 17
    mc.close()


def test_rebasing_when_the_levels_shrink():
    """(2,1,{v1,v2},2): 2 073 states in 40 levels that grow to level 14 and shrink from there.  Three stored levels, two passes through the
    seen-set alone — then advance() finds that the newest seen-set-only level fits the idle record buffer, spends one descent on regenerating
    it INTO that buffer (k_export packs the regenerated slices), and the rest of the run is ordinary stored levels again: every level against
    the oracle to exhaustion, and the re-based level itself state by state (fingerprint set and records)."""
    import vsr_tlaplus_amd as vt
    from oracle import orc
    P = orc.Params(2, 1, 2, 2)
    want, _ = _oracle_levels(orc, P, 100)
    m = vt.Model.from_constants(R=2, C_=1, n=2, L=2)
    mc = vt.ModelChecker(m, table_log2=16, frontier_words=1 << 20, frontier_states=1 << 14, pending_entries=1 << 15)
    for _ in range(3):
        mc.step()
    for k in (3, 4):
        a, _b = mc.deepen()
        _same(a, want[k])
    assert mc.depth == 6 and mc.level == 4 and not mc.rebased
    kinds = []
    for w in want[5:]:
        kind, a, b = mc.advance()
        kinds.append(kind)
        if w["n_new"] == 0:
            assert a["n_new"] == 0
            break
        assert (a["level"], a["n_new"], a["generated"], a["deadlocks"]) == (w["level"], w["n_new"], w["generated"], w["deadlocks"]), w["level"]
        if kind == "level":
            x, s_, n = mc.level_checksum()
            assert (x, s_, n) == (w["fp_xor"], w["fp_sum"], w["n_new"]), w["level"]
    assert mc.rebased and mc.rebased[0]["level"] == 6 and mc.rebased[0]["n"] == want[4]["n_new"], mc.rebased
    assert kinds and all(k == "level" for k in kinds), kinds       # stored levels from the re-based one on
    assert mc.distinct == 2073
    mc.close()
    # the re-based level itself: regenerated into the record buffer = the oracle's level, fingerprints and records
    mc = vt.ModelChecker(m, table_log2=16, frontier_words=1 << 20, frontier_states=1 << 14, pending_entries=1 << 15)
    for _ in range(3):
        mc.step()
    mc.deepen()
    mc.deepen()
    ob = orc.Bfs(P)
    for _ in range(5):
        ob.step()
    a, b, what = vt.capi.LevelInfo(), vt.capi.LevelInfo(), __import__("ctypes").c_int32()
    vt.capi.check(vt.capi.load().vsrmc_checker_advance(mc._h, a, b, what))
    assert what.value == 3 and a.level == 6 and a.n_new == want[4]["n_new"]
    mc.level, mc.n_frontier, mc.depth = a.level, a.n_new, a.level
    assert np.array_equal(mc.level_fps(), ob.level_fps(6))
    words, off = mc.frontier()
    got = sorted(orc.fingerprint(P, words[int(off[i]): int(off[i + 1])])[0] for i in range(len(off) - 1))
    assert np.array_equal(np.array(got, dtype=np.uint64), ob.level_fps(6))
    d = mc.step()                                                # and the search steps on from it the ordinary way
    assert (d["level"], d["n_new"], d["generated"]) == (7, want[5]["n_new"], want[5]["generated"])
    mc.close()


def test_a_deep_search_is_checkpointed_and_recovered_in_another_process(tmp_path):
    """a search that has gone beyond its record buffers is saved BETWEEN TWO PASSES (seen-set, base level, the descriptors of the
    levels that exist in the seen-set only) by one process and recovered by another one, which reproduces every later level figure of the
    uninterrupted run — the oracle's — to the violation at depth 19, with its counter-example."""
    import subprocess
    import sys
    import os
    import vsr_tlaplus_amd as vt
    from oracle import orc
    P = orc.Params(3, 1, 2, 1, invariant_mask=2)
    want, ob = _oracle_levels(orc, P, 19)
    words, off = ob.frontier()
    viol = min(orc.fingerprint(P, words[int(off[i]): int(off[i + 1])])[0] for i in range(len(off) - 1)
               if orc.invariants(P, words[int(off[i]): int(off[i + 1])]))
    path = str(tmp_path / "deep.chk")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = ("import sys; sys.path.insert(0, %r)\n"
              "import vsr_tlaplus_amd as vt\n"
              "m = vt.Model.from_constants(R=3, C_=1, n=2, L=1, invariant_mask=2)\n"
              "mc = vt.ModelChecker(m, table_log2=20, frontier_words=1 << 21, frontier_states=1 << 15, pending_entries=1 << 16)\n"
              "for _ in range(8): mc.step()\n"
              "for _ in range(4): a, b = mc.deepen()\n"
              "mc.save(%r)\n"
              "print('SAVED', mc.level, mc.depth, mc.distinct)\n") % (root, path)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SAVED 9 13" in r.stdout, r.stdout + r.stderr
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=1, invariant_mask=2)
    mc = vt.ModelChecker(m, table_log2=20, frontier_words=1 << 21, frontier_states=1 << 15, pending_entries=1 << 16, recover=path)
    assert (mc.level, mc.depth) == (9, 13) and mc.distinct == sum(w["n_new"] for w in want[:12]) + 1
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
    assert len(mc.probe_trace()) == 19
    mc.close()
    with pytest.raises(vt.VsrmcError):                           # another model's constants: refused as before
        vt.ModelChecker(vt.Model.from_constants(R=3, C_=1, n=2, L=2), table_log2=20, frontier_words=1 << 21, frontier_states=1 << 15, recover=path)
