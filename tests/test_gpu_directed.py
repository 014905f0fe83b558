"""Directed parity tests (`-m gpu`) for the actions the whole-space tests barely reach: the state-transfer actions
SendGetState / ReceiveGetState / ReceiveNewState (VSR.tla:496-567) — the ones behind the defect the reference exists to show
(README:9-18) — and ReceiveHigherDVC (VSR.tla:677-688).

Part 1 harvests, from the GPU BFS of BASELINE config 2, states of levels 14.. in which these actions are enabled
(vsrmc_checker_select, a coverage filter) and compares, state by state, the successor multiset of the HIP path (k_successors
through the C ABI: action, record, fingerprint, auxkey, invariant verdict) with the C++ oracle's, and for a sample with the
independent Python restatement's.  >= 1000 directly compared instances per action.

Part 2 feeds hand-built records through all three: the log-merge branches of VSR.tla:557-561, SendOnce blocked by a key with
delivery count 0 (VSR.tla:250-252), the MinVal truncation of VSR.tla:504-507.
"""
import collections

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A_ReceiveHigherDVC, A_SendGetState, A_ReceiveGetState, A_ReceiveNewState = 5, 13, 14, 15


@pytest.fixture(scope="module")
def vt():
    import vsr_tlaplus_amd as vt
    assert vt.load().vsrmc_device_count() >= 1, "no HIP device visible"
    return vt


@pytest.fixture(scope="module")
def orc():
    from oracle import orc
    return orc


def _norm(orc, P, words):
    return tuple(int(x) for x in orc.normalise(P, words))


def _gpu_by_parent(orc, P, m, words, off):
    by = collections.defaultdict(list)
    for s in m.get_next_states(words, off):
        assert s["err"] == 0
        by[s["parent"]].append((s["action"], s["fp"], s["auxkey"], s["inv"], _norm(orc, P, s["words"])))
    return by


def test_state_transfer_actions_state_by_state(vt, orc):
    from oracle import pycodec, pyoracle as po
    P = orc.Params(3, 1, 2, 2)
    PM = po.Model(3, 1, ("v1", "v2"), 2)
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
    mc = vt.ModelChecker(m, table_log2=27, frontier_words=1 << 30, frontier_states=1 << 25, pending_entries=1 << 15, keep_trace=False)
    want = 1000
    instances = collections.Counter()          # directly compared instances per action
    py_checked = collections.Counter()
    targets = (A_ReceiveHigherDVC, A_SendGetState, A_ReceiveGetState, A_ReceiveNewState)
    while mc.level < 23 and any(instances[a] < want for a in targets):
        d = mc.step()
        assert d["n_new"] > 0 and not d["viol_mask"]
        if mc.level < 14:
            continue
        for a in targets:
            if instances[a] >= want:
                continue
            words, off, total = mc.select(1 << a, 400)
            assert total >= len(off) - 1
            if len(off) < 2:
                continue
            gpu = _gpu_by_parent(orc, P, m, words, off)
            for i in range(len(off) - 1):
                rec = words[int(off[i]): int(off[i + 1])]
                osucc = orc.successors(P, rec)
                mine = sorted(gpu.get(i, []))
                theirs = sorted((s["action"], s["fp"], s["auxkey"], s["inv"], _norm(orc, P, s["words"])) for s in osucc)
                assert mine == theirs, (mc.level, vt.ACTION_NAMES[a], i)
                k = sum(1 for s in osucc if s["action"] == a)
                assert k >= 1, "vsrmc_checker_select returned a state without the action"
                instances[a] += k
                if py_checked[a] < 60:                      # the independent Python restatement on a sample
                    st = pycodec.unpack(PM, [int(x) for x in rec])
                    ps = sorted((n, tuple(pycodec.normalise(PM, pycodec.pack(PM, t)))) for n, t in po.successors(PM, st))
                    cs = sorted((vt.ACTION_NAMES[s["action"]], tuple(pycodec.normalise(PM, [int(v) for v in s["words"]]))) for s in osucc)
                    assert ps == cs
                    py_checked[a] += 1
    mc.close()
    for a in targets:
        assert instances[a] >= want, (vt.ACTION_NAMES[a], instances[a])


# ---------------------------------------------------------------------------------------------------------------------
# hand-built records
# ---------------------------------------------------------------------------------------------------------------------
def _three_way(vt, orc, R, n, L, states):
    """successor multisets of python states: HIP == C++ oracle == Python restatement; returns the oracle's lists."""
    from oracle import pycodec, pyoracle as po
    vals = tuple("v%d" % (i + 1) for i in range(n))
    PM = po.Model(R, 1, vals, L)
    P = orc.Params(R, 1, n, L)
    m = vt.Model.from_constants(R=R, C_=1, n=n, L=L)
    recs = [np.array(pycodec.pack(PM, s), dtype=np.uint64) for s in states]
    words = np.concatenate(recs)
    off = np.cumsum([0] + [len(r) for r in recs]).astype(np.uint64)
    gpu = _gpu_by_parent(orc, P, m, words, off)
    out = []
    for i, (s, rec) in enumerate(zip(states, recs)):
        osucc = orc.successors(P, rec)
        theirs = sorted((x["action"], x["fp"], x["auxkey"], x["inv"], _norm(orc, P, x["words"])) for x in osucc)
        assert sorted(gpu.get(i, [])) == theirs, i
        ps = sorted((nm, tuple(pycodec.normalise(PM, pycodec.pack(PM, t)))) for nm, t in po.successors(PM, s))
        cs = sorted((vt.ACTION_NAMES[x["action"]], tuple(pycodec.normalise(PM, [int(v) for v in x["words"]]))) for x in osucc)
        assert ps == cs, i
        out.append([(vt.ACTION_NAMES[x["action"]], pycodec.unpack(PM, [int(v) for v in x["words"]])) for x in osucc])
    return out


def _base_state(po, PM, view=2):
    """all replicas Normal in `view`, empty logs, empty bag"""
    s = po.Init(PM)
    R = PM.R
    s = po.upd(s, rep_view_number=tuple(view for _ in range(R)), rep_last_normal_view=tuple(view for _ in range(R)), aux_svc=view - 1)
    return s


def test_hand_built_state_transfer_records(vt, orc):
    from oracle import pyoracle as po
    PM = po.Model(3, 1, ("v1", "v2", "v3"), 3)
    e = lambda view, v, req: po.rec(view_number=view, operation=v, client_id=1, request_number=req)   # noqa: E731
    e1, e2, e3 = e(1, "v1", 1), e(2, "v2", 2), e(2, "v3", 3)
    acked = {"v1": True, "v2": False, "v3": False}
    states = []

    def mk(view, logs, ops, commits, msgs, **kw):
        s = _base_state(po, PM, view)
        s = po.upd(s, rep_log=tuple(tuple(l) for l in logs), rep_op_number=tuple(ops), rep_commit_number=tuple(commits),
                   messages=dict(msgs), aux_client_acked=dict(acked), **kw)
        states.append(s)
        return s

    def newstate(view, dest, source, first, log, op, commit):
        return po.rec(type="NewStateMsg", view_number=view, first_op=first, log=tuple((first + i, x) for i, x in enumerate(log)),
                      op_number=op, commit_number=commit, dest=dest, source=source)

    # ---- ReceiveNewState, VSR.tla:551-567 / log merge :557-561
    # (a) replica 3 holds [e1], the message brings ops 2..3: the log becomes [e1, e2, e3]
    mk(2, [[e1, e2, e3], [e1, e2, e3], [e1]], [3, 3, 1], [1, 1, 1], {newstate(2, 3, 2, 2, [e2, e3], 3, 1): 1})
    # (b) replica 3 holds nothing, first_op = 1: the whole log arrives
    mk(2, [[e1, e2], [e1, e2], []], [2, 2, 0], [1, 1, 0], {newstate(2, 3, 2, 1, [e1, e2], 2, 1): 1})
    # (c) first_op does not continue the replica's log (op 1 held, first_op 3): not enabled
    mk(2, [[e1, e2, e3], [e1, e2, e3], [e1]], [3, 3, 1], [1, 1, 1], {newstate(2, 3, 2, 3, [e3], 3, 1): 1})
    # (d) the same message for another view, and with delivery count 0: not enabled
    mk(3, [[e1, e2, e3], [e1, e2, e3], [e1]], [3, 3, 1], [1, 1, 1], {newstate(2, 3, 2, 2, [e2, e3], 3, 1): 1})
    mk(2, [[e1, e2, e3], [e1, e2, e3], [e1]], [3, 3, 1], [1, 1, 1], {newstate(2, 3, 2, 2, [e2, e3], 3, 1): 0})
    # (e) two copies in flight: one is consumed, one stays
    mk(2, [[e1, e2, e3], [e1, e2, e3], [e1]], [3, 3, 1], [1, 1, 1], {newstate(2, 3, 2, 2, [e2, e3], 3, 1): 2})

    # ---- SendGetState, VSR.tla:496-516: a Prepare of a higher view that skips an op; MinVal truncation :504-507
    def prepare(view, dest, source, op, commit, entry):
        return po.rec(type="PrepareMsg", view_number=view, message=entry, op_number=op, commit_number=commit, dest=dest, source=source)

    def getstate(view, dest, source, op):
        return po.rec(type="GetStateMsg", view_number=view, op_number=op, dest=dest, source=source)

    def lagging(extra_msgs, log3, op3, commit3):
        """replica 3 still Normal in view 1 (primary of view 2 = replica 2 sends a Prepare for op 3 in view 2)"""
        s = _base_state(po, PM, 2)
        s = po.upd(s, rep_view_number=(2, 2, 1), rep_last_normal_view=(2, 2, 1),
                   rep_log=((e1, e2, e3), (e1, e2, e3), tuple(log3)), rep_op_number=(3, 3, op3), rep_commit_number=(1, 1, commit3),
                   messages=dict([(prepare(2, 3, 2, 3, 1, e3), 1)] + list(extra_msgs)), aux_client_acked=dict(acked))
        states.append(s)
        return s

    i_sgs = len(states)
    lagging([], [e1], 1, 1)                                   # commit = Len(log): nothing truncated, GetState(op 1) to 1 and 2
    lagging([], [e1], 1, 0)                                   # MinVal: the log is cut back to the commit number 0 -> <<>>
    lagging([], [], 0, 0)
    lagging([(getstate(2, 1, 3, 1), 0)], [e1], 1, 1)          # SendOnce: the key to replica 1 exists with count 0 -> only rDest = 2
    lagging([(getstate(2, 1, 3, 1), 0), (getstate(2, 2, 3, 1), 1)], [e1], 1, 1)   # both keys present -> not enabled
    lagging([(getstate(2, 1, 3, 0), 0)], [e1], 1, 1)          # a key for another op number does not block

    # ---- ReceiveGetState, VSR.tla:526-543: the reply carries ops mop+1 .. op
    i_rgs = len(states)
    mk(2, [[e1, e2, e3], [e1, e2, e3], [e1]], [3, 3, 1], [2, 1, 1], {getstate(2, 1, 3, 1): 1})
    mk(2, [[e1, e2, e3], [e1, e2, e3], []], [3, 3, 0], [2, 1, 0], {getstate(2, 2, 3, 0): 1})
    mk(2, [[e1], [e1, e2, e3], [e1]], [1, 3, 1], [1, 1, 1], {getstate(2, 1, 3, 1): 1})     # op_number not larger: not enabled

    succ = _three_way(vt, orc, 3, 3, 3, states)
    names = [[a for a, _ in s] for s in succ]
    assert names[0].count("ReceiveNewState") == 1 and names[1].count("ReceiveNewState") == 1
    got = [t for a, t in succ[0] if a == "ReceiveNewState"][0]
    assert got["rep_log"][2] == (e1, e2, e3) and got["rep_op_number"][2] == 3
    got = [t for a, t in succ[1] if a == "ReceiveNewState"][0]
    assert got["rep_log"][2] == (e1, e2) and got["rep_op_number"][2] == 2
    assert all("ReceiveNewState" not in names[k] for k in (2, 3, 4)) and names[5].count("ReceiveNewState") == 1
    assert names[i_sgs].count("SendGetState") == 2 and names[i_sgs + 1].count("SendGetState") == 2
    cut = [t for a, t in succ[i_sgs + 1] if a == "SendGetState"][0]
    assert cut["rep_log"][2] == () and cut["rep_op_number"][2] == 0 and cut["rep_view_number"][2] == 2
    assert names[i_sgs + 3].count("SendGetState") == 1 and names[i_sgs + 4].count("SendGetState") == 0
    assert names[i_sgs + 5].count("SendGetState") == 2
    assert names[i_rgs].count("ReceiveGetState") == 1 and names[i_rgs + 1].count("ReceiveGetState") == 1
    assert names[i_rgs + 2].count("ReceiveGetState") == 0
