"""Second model: /root/reference/vsr-revisited/paper/analysis/03-state-transfer/VR_STATE_TRANSFER.{tla,cfg}.
Two independent CPU readings — the C++ oracle (oracle/vrst_oracle.cpp) and the Python restatement (oracle/pyoracle2.py) — must
agree on whole small state spaces: level sizes, generated counts, per-state successor multisets (action, record), invariant
verdicts, and fingerprint = VIEW identity (no collisions, no false splits).  Expected outcome of the model: no violation."""
import numpy as np
import pytest

from oracle import orc2, pyoracle2 as po


def _cpp_levels(P, max_depth):
    b = orc2.Bfs(P)
    out = [dict(new=1, generated=0)]
    while b.info["depth"] < max_depth:
        nn = b.step()
        if nn == 0:
            break
        assert b.info["viol_mask"] == 0 and b.info["ties"] == 0
        out.append(dict(new=nn, generated=b.info["generated"]))
    total = b.info["distinct"]
    b.close()
    return out, total


@pytest.mark.parametrize("R,vals,L,depth", [(2, ("v1",), 1, 99), (2, ("v1", "v2"), 1, 99), (2, ("v1", "v2"), 2, 14), (3, ("v1",), 1, 11), (3, ("v1", "v2"), 2, 7)])
def test_level_counts_agree(R, vals, L, depth):
    M = po.Model(R, vals, L)
    levels, gen, viol = po.bfs(M, max_depth=depth)
    assert viol is None                                     # VR_STATE_TRANSFER.cfg: the invariants hold
    cpp, total = _cpp_levels(orc2.Params(R, len(vals), L), depth)
    assert [l["new"] for l in cpp] == [len(l) for l in levels]
    assert [l["generated"] for l in cpp] == gen
    if depth == 99:
        assert total == sum(len(l) for l in levels)


def test_successor_sets_agree_on_every_state_of_a_small_space():
    M = po.Model(2, ("v1", "v2"), 2)
    P = orc2.Params(2, 2, 2)
    levels, _, _ = po.bfs(M, max_depth=16)
    fp_of_view = {}
    acts = set()
    for lvl in levels:
        for s in lvl[::6]:
            w = np.array(po.pack(M, s), dtype=np.uint64)
            succ = orc2.successors(P, w)
            cs = sorted((orc2.ACTIONS[x["action"]], tuple(po.normalise(M, [int(v) for v in x["words"]])), x["inv"]) for x in succ)
            ps = sorted((n, tuple(po.normalise(M, po.pack(M, t))), po.invariant_mask(M, t)) for n, t in po.successors(M, s))
            assert cs == ps
            acts.update(a for a, _, _ in cs)
            assert orc2.invariants(P, w) == po.invariant_mask(M, s) == 0
            fp, _ = orc2.fingerprint(P, w)
            assert fp_of_view.setdefault(po.view_of(s), fp) == fp
    assert len(set(fp_of_view.values())) == len(fp_of_view) == sum(len(l[::6]) for l in levels)
    assert {"TimerSendSVC", "SendDVC", "SendSV", "ReceiveSV", "ReceiveClientRequest", "ReceivePrepareMsg", "ExecuteOp"} <= acts


def test_state_transfer_actions_of_the_second_model():
    """states of (3, two values, limit 1) in which SendGetState / ReceiveGetState / ReceiveNewState fire — found in the C++ oracle's BFS
    by a cheap look at the bag (an undelivered GetState / NewState, or an undelivered Prepare of a later view): the Python
    restatement produces the same successors"""
    M = po.Model(3, ("v1", "v2"), 1)
    P = orc2.Params(3, 2, 1)
    FIXED = 4                                                   # words before the bag in the wire record
    b = orc2.Bfs(P)
    seen = {13: 0, 14: 0, 15: 0}
    while b.info["depth"] < 17 and min(seen.values()) < 12:
        assert b.step() > 0
        if b.info["depth"] < 9:
            continue
        words, off = b.frontier()
        off = off.astype(np.int64)
        idx = np.arange(len(words), dtype=np.int64)
        rid = np.searchsorted(off, idx, side="right") - 1
        t, cnt, view = words & np.uint64(7), (words >> np.uint64(21)) & np.uint64(3), (words >> np.uint64(3)) & np.uint64(7)
        hot = ((t == 6) | (t == 7) | ((t == 2) & (view >= 2))) & (cnt > 0) & (idx - off[rid] >= FIXED)
        for i in np.unique(rid[hot]):
            rec = words[int(off[i]): int(off[i + 1])]
            succ = orc2.successors(P, rec)
            hit = [x["action"] for x in succ if x["action"] in seen]
            if not hit or all(seen[a] >= 40 for a in hit):
                continue
            for a in hit:
                seen[a] += 1
            s = po.unpack(M, [int(x) for x in rec])
            cs = sorted((orc2.ACTIONS[x["action"]], tuple(po.normalise(M, [int(v) for v in x["words"]]))) for x in succ)
            ps = sorted((n, tuple(po.normalise(M, po.pack(M, t2)))) for n, t2 in po.successors(M, s))
            assert cs == ps
    b.close()
    assert min(seen.values()) >= 12, seen
