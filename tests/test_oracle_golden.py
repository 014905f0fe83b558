"""The CPU oracle(s) against the reference's golden vector (tests/golden/state_transfer_trace.json, derived from
/root/reference/state_transfer_violation_trace.txt by tests/golden/make_golden.py) and against each other."""
import numpy as np

from oracle import orc, pycodec, pyoracle as po


def _words(st):
    return np.array([int(w, 16) for w in st["words"]], dtype=np.uint64)


def test_trace_shape(golden_trace):
    sts = golden_trace["states"]
    assert len(sts) == 24                                   # trace:556 position 24
    assert [s["position"] for s in sts] == list(range(1, 25))
    assert sts[0]["action"] == "Initial predicate"
    assert [s["holds"] for s in sts] == [True] * 23 + [False]


def test_cpp_oracle_replays_golden_trace(golden_trace):
    """Every step of the reference trace is a successor under the C++ oracle, produced by the named action; the
    invariant AcknowledgedWriteNotLost (VSR.tla:945-950) holds in states 1..23 and fails in state 24."""
    p = golden_trace["params"]
    P = orc.Params(p["R"], p["C"], len(p["values"]), p["L"])
    sts = golden_trace["states"]
    assert np.array_equal(orc.normalise(P, orc.init_record(P)), orc.normalise(P, _words(sts[0])))
    for i in range(len(sts) - 1):
        cur, nxt = _words(sts[i]), orc.normalise(P, _words(sts[i + 1]))
        succ = orc.successors(P, cur)
        hits = [s for s in succ if np.array_equal(orc.normalise(P, s["words"]), nxt)]
        assert len(hits) == 1, (i, len(hits))
        assert orc.ACTION_NAMES[hits[0]["action"]] == sts[i + 1]["action"] if hasattr(orc, "ACTION_NAMES") else True
        assert "%016x" % hits[0]["fp"] == sts[i + 1]["fp"]
        assert hits[0]["auxkey"] == sts[i + 1]["auxkey"]
        assert hits[0]["inv"] == sts[i + 1]["inv_mask"]
    assert [orc.invariants(P, _words(s)) for s in sts] == [0] * 23 + [1]


def test_pyoracle_replays_golden_trace(golden_trace):
    p = golden_trace["params"]
    M = po.Model(p["R"], p["C"], tuple(p["values"]), p["L"])
    sts = golden_trace["states"]
    for i in range(len(sts) - 1):
        cur = pycodec.unpack(M, [int(w, 16) for w in sts[i]["words"]])
        nxt = pycodec.normalise(M, [int(w, 16) for w in sts[i + 1]["words"]])
        hits = [n for n, t in po.successors(M, cur) if pycodec.normalise(M, pycodec.pack(M, t)) == nxt]
        assert hits == [sts[i + 1]["action"]], (i, hits)


def test_codec_roundtrip_three_ways(golden_trace):
    """python pack/unpack and C++ decode/encode agree on every golden state."""
    p = golden_trace["params"]
    M = po.Model(p["R"], p["C"], tuple(p["values"]), p["L"])
    P = orc.Params(p["R"], p["C"], len(p["values"]), p["L"])
    for st in golden_trace["states"]:
        w = [int(x, 16) for x in st["words"]]
        assert pycodec.normalise(M, pycodec.pack(M, pycodec.unpack(M, w))) == pycodec.normalise(M, w)
        assert pycodec.normalise(M, [int(x) for x in orc.normalise(P, np.array(w, dtype=np.uint64))]) == pycodec.normalise(M, w)


def test_fingerprint_is_symmetric_and_view_only(golden_trace):
    """fp is invariant under every permutation of Values (VSR.cfg:31) and ignores the aux variables (VSR.tla:149-150)."""
    from itertools import permutations
    p = golden_trace["params"]
    M = po.Model(p["R"], p["C"], tuple(p["values"]), p["L"])
    P = orc.Params(p["R"], p["C"], len(p["values"]), p["L"])
    for st in golden_trace["states"][::3]:
        s = pycodec.unpack(M, [int(x, 16) for x in st["words"]])
        fp0 = int(st["fp"], 16)
        for perm in permutations(M.Values):
            pi = dict(zip(M.Values, perm))
            t = {k: po.permute_value(v, pi) for k, v in s.items()}
            fp, _ = orc.fingerprint(P, np.array(pycodec.pack(M, t), dtype=np.uint64))
            assert fp == fp0
        t = dict(s)
        t["aux_svc"] = (s["aux_svc"] + 1) % 4
        fp, ak = orc.fingerprint(P, np.array(pycodec.pack(M, t), dtype=np.uint64))
        assert fp == fp0 and ak != st["auxkey"]
