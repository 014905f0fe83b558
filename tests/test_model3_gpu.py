"""Third model ("then 04-application-state") on the GPU: analysis/04-application-state/VR_APP_STATE.tla lowered by
csrc/vras_actions.hpp behind the same kernels, checked bit for bit against its CPU oracle (oracle/vras_oracle.cpp via
oracle/orc3.py): per-level fingerprint sets, new / generated / deadlock counts, per-state successor multisets (action, record,
fingerprint, auxkey, invariant verdict), and the outcome on VR_APP_STATE.cfg as deep as the oracle went — no invariant is violated."""
import collections

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vt():
    import vsr_tlaplus_amd as vt
    assert vt.load().vsrmc_device_count() >= 1, "no HIP device visible"
    return vt


@pytest.fixture(scope="module")
def orc2():
    from oracle import orc3
    return orc3


def _norm(orc2, P, words):
    return tuple(int(x) for x in orc2.normalise(P, words))


def _compare_levels(vt, orc2, R, n, L, max_depth, exact=False, sizes=None):
    P = orc2.Params(R, n, L)
    m = vt.Model.third_model(R=R, n=n, L=L)
    mc = vt.ModelChecker(m, exact_ties=exact, **(sizes or dict(table_log2=22, frontier_words=1 << 24, frontier_states=1 << 19,
                                                               pending_entries=1 << 21)))
    ob = orc2.Bfs(P)
    level = 1
    while level < max_depth:
        assert np.array_equal(mc.level_fps(), ob.level_fps(level)), "fingerprint sets differ at level %d" % level
        d = mc.step()
        nn = ob.step()
        assert (d["n_new"], d["generated"], d["deadlocks"], d["viol_mask"]) == (nn, ob.info["generated"], ob.info["deadlocks"], 0), level
        assert ob.info["ties"] == 0 and ob.info["viol_mask"] == 0
        if nn == 0:
            break
        level += 1
    total = mc.distinct
    mc.close()
    ob.close()
    return total, level


@pytest.mark.parametrize("exact", [False, True])
def test_model3_small_spaces_whole(vt, orc2, exact):
    assert _compare_levels(vt, orc2, 2, 1, 1, 100, exact=exact) == (76, 14)
    total, level = _compare_levels(vt, orc2, 2, 2, 2, 100, exact=exact)
    assert (total, level) == (14735, 27)


def test_model3_shipped_cfg_prefix_and_other_sizes(vt, orc2):
    """VR_APP_STATE.cfg:4-7 (3 replicas, {a,b}, limit 2): the specialised kernel k_expand<true, 2302>; then the generic one on
    (3, {a}, 1) whole, (3, {a,b}, 1) to depth 16 and (3, {a,b,c}, 2) in the two-kernel scheme"""
    total, level = _compare_levels(vt, orc2, 3, 2, 2, 12)
    assert level == 12
    assert _compare_levels(vt, orc2, 3, 1, 1, 100) == (42738, 24)
    _compare_levels(vt, orc2, 3, 2, 1, 16)
    _compare_levels(vt, orc2, 3, 3, 2, 9, exact=True)


def test_model3_successors_state_by_state(vt, orc2):
    """every state of (2, {a,b}, 2) and the states of (3, {a,b}, 2) at levels 9-13 in which the state-transfer actions fire
    (vsrmc_checker_select): successor multisets of the HIP path == the oracle's"""
    P = orc2.Params(2, 2, 2)
    m = vt.Model.third_model(R=2, n=2, L=2)
    b = orc2.Bfs(P)
    checked = 0
    acts = collections.Counter()
    while True:
        words, off = b.frontier() if b.info["depth"] > 1 else (orc2.init_record(P), np.array([0, len(orc2.init_record(P))], dtype=np.uint64))
        by = collections.defaultdict(list)
        for s in m.get_next_states(words, off):
            assert s["err"] == 0
            by[s["parent"]].append((s["action"], s["fp"], s["auxkey"], s["inv"], _norm(orc2, P, s["words"])))
        for i in range(len(off) - 1):
            osucc = orc2.successors(P, words[int(off[i]): int(off[i + 1])])
            assert sorted(by.get(i, [])) == sorted((s["action"], s["fp"], s["auxkey"], s["inv"], _norm(orc2, P, s["words"])) for s in osucc)
            acts.update(s["action"] for s in osucc)
            checked += 1
        if b.step() == 0 or checked > 4000:
            break
    assert checked > 1000 and all(acts[a] > 0 for a in (1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12)), acts
    # the state-transfer actions (13 SendGetState, 14 ReceiveGetState, 15 ReceiveNewState) on the shipped constants
    P = orc2.Params(3, 2, 2)
    m = vt.Model.third_model(R=3, n=2, L=2)
    mc = vt.ModelChecker(m, table_log2=24, frontier_words=1 << 27, frontier_states=1 << 22, pending_entries=1 << 15, keep_trace=False)
    seen = collections.Counter()
    while mc.level < 15 and any(seen[a] < 50 for a in (13, 14, 15)):
        mc.step()
        if mc.level < 8:
            continue
        for a in (13, 14, 15):
            words, off, total = mc.select(1 << a, 150)
            if len(off) < 2:
                continue
            by = collections.defaultdict(list)
            for s in m.get_next_states(words, off):
                by[s["parent"]].append((s["action"], s["fp"], s["auxkey"], s["inv"], _norm(orc2, P, s["words"])))
            for i in range(len(off) - 1):
                osucc = orc2.successors(P, words[int(off[i]): int(off[i + 1])])
                assert sorted(by.get(i, [])) == sorted((s["action"], s["fp"], s["auxkey"], s["inv"], _norm(orc2, P, s["words"])) for s in osucc)
                k = sum(1 for s in osucc if s["action"] == a)
                assert k >= 1
                seen[a] += k
    mc.close()
    assert all(seen[a] >= 50 for a in (13, 14, 15)), seen


def test_model3_whole_workload_against_the_oracle(vt, oracle_levels):
    """the shipped VR_APP_STATE.cfg as deep as the CPU oracle went (tests/golden/oracle_levels_model3.json: 29 levels, 2.5e9 states — levels 23-29 from the memory-lean driver)"""
    g = oracle_levels["model3"]
    p = g["params"]
    m = vt.Model.third_model(R=p["R"], n=p["n"], L=p["L"], invariant_mask=p["inv_mask"])
    mc = vt.ModelChecker.auto(m)                                 # sized from the free HBM; every level of the fixture fits the record buffers
    for lv in g["levels"][1:]:
        kind, d, _ = mc.advance()                                  # a level that does not fit the record buffers lives in the seen-set only ("deep")
        assert (d["level"], d["n_new"], d["generated"], d["deadlocks"], d["max_bag"], d["viol_mask"]) == \
            (lv["level"], lv["new"], lv["generated"], lv["deadlocks"], lv["max_bag"], 0), lv["level"]
        assert [int(x) for x in d["act_generated"][1:16]] == lv["act_generated"][1:16], lv["level"]
        x, s = (mc.level_checksum()[:2]) if kind == "level" else (d["fp_xor"], d["fp_sum"])
        assert ("%016x" % x, "%016x" % s) == (lv["fp_xor"], lv["fp_sum"]), lv["level"]
    mc.close()


def test_model3_simulation_and_printer(vt, orc2):
    """random walks find no violation; the TLC-style printer names every variable"""
    m = vt.Model.third_model(R=3, n=2, L=2)
    r = m.simulate(n_walkers=1 << 14, max_depth=40, seed=7, max_seconds=2.0)
    assert r["found"] == 0 and r["steps"] > 10 ** 6
    txt = m.format_state(m.init_state())
    for var in ("aux_client_acked", "aux_svc", "messages", "no_progress", "no_progress_ctr", "rep_commit_number", "rep_last_normal_view",
                "rep_log", "rep_op_number", "rep_peer_op_number", "rep_sent_dvc", "rep_sent_sv", "rep_status", "rep_view_number", "replicas",
                "rep_app_state", "rep_recv_dvc", "rep_rec_number", "rep_rec_recv", "aux_restart"):
        assert ("\n%s |-> " % var) in txt
    assert "rep_last_normal_view |-> <<1, 1, 1>>" in txt and "rep_status |-> <<Normal, Normal, Normal>>" in txt


def test_model3_cli(vt, orc2, tmp_path):
    """vsrmc -config <VR_APP_STATE.cfg> (no .tla: recognised by its constants and NoAppStateDivergence) to depth 12: TLC-style
    progress lines, the oracle's distinct-state count, no error"""
    import os
    import subprocess
    from test_model3_host_cpu import _cfg
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vsr_tlaplus_amd", "vsrmc")
    r = subprocess.run([cli, "-config", _cfg(tmp_path), "-maxDepth", "12", "-tableLog2", "22", "-frontierGiB", "0.5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ob = orc2.Bfs(orc2.Params(3, 2, 2))
    while ob.info["depth"] < 12:
        ob.step()
    assert "VR_APP_STATE.tla lowered" in r.stdout and "invariant mask 30" in r.stdout
    assert "%d distinct states found" % ob.info["distinct"] in r.stdout, r.stdout[-1500:]
    ob.close()


def test_model3_validate_trace_cli(vt, tmp_path):
    """a 15-state behaviour of VR_APP_STATE (GPU successors, an arbitrary but fixed choice per step), printed as a TLA+ trace
    expression by the product's printer, is read back and re-walked by `vsrmc -validateTrace`; with one state damaged it is refused
    at that state"""
    import os
    import subprocess
    from test_model3_host_cpu import _cfg
    m = vt.Model.third_model(R=3, n=2, L=2)
    rec = m.init_state()
    path = [("Initial predicate", rec)]
    from vsr_tlaplus_amd.checker import ACTION_NAMES
    for t in range(14):
        succ = m.get_next_states(rec, np.array([0, len(rec)], dtype=np.uint64))
        assert succ
        s = succ[(5 * t + 3) % len(succ)]
        rec = s["words"]
        path.append((ACTION_NAMES[s["action"]], rec))
    body = ",\n".join("[\n _TEAction |-> [\n   position |-> %d,\n   name |-> \"%s\",\n   location |-> \"Unknown location\"\n ],\n%s"
                       % (k + 1, a, m.format_state(r).split("\n", 1)[1]) for k, (a, r) in enumerate(path))
    f = tmp_path / "m3.trace"
    f.write_text("<<\n" + body + "\n>>\n")
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vsr_tlaplus_amd", "vsrmc")
    r = subprocess.run([cli, "-config", _cfg(tmp_path), "-validateTrace", str(f)], capture_output=True, text=True, timeout=300)
    assert "15 states read" in r.stdout and "The trace is a behaviour of the model." in r.stdout, r.stdout + r.stderr
    bad = f.read_text().replace("rep_view_number |-> <<", "rep_view_number |-> <<7, ", 1)
    bad = bad.replace("<<7, 1, 1, 1>>", "<<7, 1, 1>>", 1)
    f.write_text(bad)
    r = subprocess.run([cli, "-config", _cfg(tmp_path), "-validateTrace", str(f)], capture_output=True, text=True, timeout=300)
    assert "The trace is a behaviour of the model." not in r.stdout, r.stdout


def test_checkpoints_of_the_analysis_models_do_not_cross(vt, tmp_path):
    """VR_STATE_TRANSFER and VR_APP_STATE share R / n / L / inv-mask-compatible headers but not the record layout (one vs two words per
    replica) nor the action table: a checkpoint names its module, layout and fingerprint version and is refused by the other module."""
    m2 = vt.Model.second_model(R=3, n=2, L=2, invariant_mask=14)
    m3 = vt.Model.third_model(R=3, n=2, L=2, invariant_mask=14)
    a = vt.ModelChecker(m2, table_log2=18, frontier_words=1 << 21, frontier_states=1 << 16, pending_entries=1 << 15)
    for _ in range(6):
        a.step()
    path = str(tmp_path / "m2.chk")
    a.save(path)
    lvl, distinct = a.level, a.distinct
    a.close()
    with pytest.raises(vt.VsrmcError):
        vt.ModelChecker(m3, table_log2=18, frontier_words=1 << 21, frontier_states=1 << 16, pending_entries=1 << 15, recover=path)
    b = vt.ModelChecker(m2, table_log2=18, frontier_words=1 << 21, frontier_states=1 << 16, pending_entries=1 << 15, recover=path)
    assert (b.level, b.distinct) == (lvl, distinct)
    b.close()
