"""Second model on the GPU: analysis/03-state-transfer/VR_STATE_TRANSFER.tla lowered by csrc/vrst_actions.hpp behind
the same kernels, checked bit for bit against its CPU oracle (oracle/vrst_oracle.cpp via oracle/orc2.py): per-level fingerprint
sets, new / generated / deadlock counts, per-state successor multisets (action, record, fingerprint, auxkey, invariant verdict),
and the expected outcome of VR_STATE_TRANSFER.cfg — no invariant is violated."""
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
    from oracle import orc2
    return orc2


def _norm(orc2, P, words):
    return tuple(int(x) for x in orc2.normalise(P, words))


def _compare_levels(vt, orc2, R, n, L, max_depth, exact=False, sizes=None):
    P = orc2.Params(R, n, L)
    m = vt.Model.second_model(R=R, n=n, L=L)
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
def test_model2_small_spaces_whole(vt, orc2, exact):
    assert _compare_levels(vt, orc2, 2, 1, 1, 100, exact=exact) == (76, 14)
    total, level = _compare_levels(vt, orc2, 2, 2, 2, 100, exact=exact)
    assert total > 2000 and level > 20


def test_model2_shipped_cfg_prefix_and_other_sizes(vt, orc2):
    """VR_STATE_TRANSFER.cfg:4-7 (3 replicas, {v1,v2}, limit 2): the specialised kernel k_expand<true, 1302>; then the generic one on
    (3, {v1}, 1) whole, (4, {v1,v2}, 1) and (5, {v1}, 1) prefixes"""
    total, level = _compare_levels(vt, orc2, 3, 2, 2, 12)
    assert level == 12
    total, level = _compare_levels(vt, orc2, 3, 1, 1, 100)
    assert level > 15
    _compare_levels(vt, orc2, 4, 2, 1, 9)
    _compare_levels(vt, orc2, 5, 1, 1, 8)
    _compare_levels(vt, orc2, 3, 3, 2, 9, exact=True)


def test_model2_successors_state_by_state(vt, orc2):
    """every state of (2, {v1,v2}, 2) and the states of (3, {v1,v2}, 2) at levels 9-13 in which the state-transfer actions fire
    (vsrmc_checker_select): successor multisets of the HIP path == the oracle's"""
    P = orc2.Params(2, 2, 2)
    m = vt.Model.second_model(R=2, n=2, L=2)
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
    m = vt.Model.second_model(R=3, n=2, L=2)
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


def test_model2_whole_workload_against_the_oracle(vt, oracle_levels):
    """the shipped VR_STATE_TRANSFER.cfg as deep as the CPU oracle went on the GPU box's host (tests/golden/oracle_levels_model2.json)"""
    if "model2" not in oracle_levels:
        pytest.skip("no oracle fixture for the second model yet")
    g = oracle_levels["model2"]
    p = g["params"]
    m = vt.Model.second_model(R=p["R"], n=p["n"], L=p["L"], invariant_mask=p["inv_mask"])
    mc = vt.ModelChecker.auto(m)                                 # sized from the free HBM; every level of the fixture fits the record buffers
    for lv in g["levels"][1:]:
        kind, d, _ = mc.advance()                                  # a level that does not fit the record buffers lives in the seen-set only ("deep")
        assert (d["level"], d["n_new"], d["generated"], d["deadlocks"], d["max_bag"], d["viol_mask"]) == \
            (lv["level"], lv["new"], lv["generated"], lv["deadlocks"], lv["max_bag"], 0), lv["level"]
        assert [int(x) for x in d["act_generated"][1:16]] == lv["act_generated"][1:16], lv["level"]
        x, s = (mc.level_checksum()[:2]) if kind == "level" else (d["fp_xor"], d["fp_sum"])
        assert ("%016x" % x, "%016x" % s) == (lv["fp_xor"], lv["fp_sum"]), lv["level"]
    mc.close()


def test_model2_simulation_and_printer(vt, orc2):
    """random walks find no violation (VR_STATE_TRANSFER.cfg: the invariants hold); the TLC-style printer names every variable"""
    m = vt.Model.second_model(R=3, n=2, L=2)
    r = m.simulate(n_walkers=1 << 14, max_depth=40, seed=7, max_seconds=2.0)
    assert r["found"] == 0 and r["steps"] > 10 ** 6
    txt = m.format_state(m.init_state())
    for var in ("aux_client_acked", "aux_svc", "messages", "no_progress", "no_progress_ctr", "rep_commit_number", "rep_last_normal_view",
                "rep_log", "rep_op_number", "rep_peer_op_number", "rep_sent_dvc", "rep_sent_sv", "rep_status", "rep_view_number", "replicas"):
        assert ("\n%s |-> " % var) in txt
    assert "rep_last_normal_view |-> <<1, 1, 1>>" in txt and "rep_status |-> <<Normal, Normal, Normal>>" in txt
