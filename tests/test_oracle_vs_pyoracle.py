"""Two independent CPU readings of VSR.tla (C++ oracle with hashed canonical fingerprints vs the Python
restatement with value-level canonicalisation) must agree on whole small state spaces, and the C++ oracle must
reproduce its committed per-level fixtures (tests/golden/bfs_counts.json)."""
import numpy as np
import pytest

from oracle import orc, pycodec, pyoracle as po

M64 = (1 << 64) - 1


def _cpp_levels(P, max_depth):
    b = orc.Bfs(P)
    out = [dict(new=1, generated=0)]
    while b.info["depth"] < max_depth:
        nn = b.step()
        if nn == 0:
            break
        out.append(dict(new=nn, generated=b.info["generated"]))
    total = b.info["distinct"]
    b.close()
    return out, total


@pytest.mark.parametrize("R,C,vals,L,sym,total,depth", [
    (2, 1, ("v1",), 1, True, 76, 14),            # BASELINE config 1
    (2, 1, ("v1", "v2"), 2, True, 2073, 27),
    (2, 1, ("v1", "v2"), 2, False, 4034, 27),
])
def test_whole_space_counts_agree(R, C, vals, L, sym, total, depth):
    M = po.Model(R, C, vals, L)
    levels, gen, viol = po.bfs(M, symmetry=sym)
    assert viol is None
    assert sum(len(l) for l in levels) == total and len(levels) == depth
    cpp, cpp_total = _cpp_levels(orc.Params(R, C, len(vals), L, symmetry=sym), 10 ** 6)
    assert cpp_total == total
    assert [l["new"] for l in cpp] == [len(l) for l in levels]
    assert [l["generated"] for l in cpp] == gen


def test_successor_sets_agree_on_every_state_of_a_small_space():
    """For every reachable state of (2,1,{v1,v2},2): identical successor multiset (action, record), identical
    invariant verdict; and the hashed fingerprint separates exactly the states the value-level canonical form
    separates (no collisions, no false splits)."""
    M = po.Model(2, 1, ("v1", "v2"), 2)
    P = orc.Params(2, 1, 2, 2)
    levels, _, _ = po.bfs(M)
    fp_of_view = {}
    for lvl in levels:
        for s in lvl:
            w = np.array(pycodec.pack(M, s), dtype=np.uint64)
            cs = sorted((orc.ACTIONS[x["action"]], tuple(pycodec.normalise(M, [int(v) for v in x["words"]])))
                        for x in orc.successors(P, w))
            ps = sorted((n, tuple(pycodec.normalise(M, pycodec.pack(M, t)))) for n, t in po.successors(M, s))
            assert cs == ps
            assert (orc.invariants(P, w) == 0) == po.AcknowledgedWriteNotLost(M, s)
            fp, _ = orc.fingerprint(P, w)
            cv = po.canonical_view(M, s)
            assert fp_of_view.setdefault(cv, fp) == fp
    assert len(set(fp_of_view.values())) == len(fp_of_view) == 2073


def test_config2_prefix_agrees_with_pyoracle():
    M = po.Model(3, 1, ("v1", "v2"), 2)
    levels, gen, _ = po.bfs(M, max_depth=7)
    cpp, _ = _cpp_levels(orc.Params(3, 1, 2, 2), 7)
    assert [l["new"] for l in cpp] == [len(l) for l in levels] == [1, 3, 10, 35, 124, 403, 1200]
    assert [l["generated"] for l in cpp] == gen


@pytest.mark.parametrize("label,depth", [
    ("config1 (2,1,{v1},1)", 99), ("(2,1,{v1,v2},2)", 99), ("(2,1,{v1,v2},2) no symmetry", 99),
    ("config2 (3,1,{v1,v2},2)", 11), ("config3 (3,1,{v1,v2,v3},3)", 9), ("config5 (5,1,{v1,v2},2)", 6),
    ("config4 assume-commit-number", 7),
])
def test_cpp_oracle_reproduces_committed_fixture(golden_counts, label, depth):
    c = golden_counts[label]
    p = c["params"]
    P = orc.Params(p["R"], p["C"], p["n"], p["L"], symmetry=p["symmetry"],
                   assume_commit_number="assume" in label)
    b = orc.Bfs(P)
    for lv in c["levels"][1:depth]:
        nn = b.step()
        fps = b.level_fps(b.info["depth"])
        x, s = 0, 0
        for v in fps.tolist():
            x ^= v
            s = (s + v) & M64
        assert (nn, b.info["generated"], "%016x" % x, "%016x" % s) == (lv["new"], lv["generated"], lv["fp_xor"], lv["fp_sum"])
        assert b.info["ties"] == 0            # no same-level VIEW ties with different aux
    b.close()


def test_config4_strict_raises_tlc_evaluation_error(golden_counts):
    """ClientCount=2: VSR.tla:421 selects the nonexistent field `commit` -> TLC aborts."""
    P = orc.Params(3, 2, 3, 3)
    b = orc.Bfs(P)
    b.step()
    with pytest.raises(orc.OracleError) as e:
        b.step()
    assert e.value.code == -1 and "VSR.tla:421" in str(e.value)
    assert golden_counts["config4 (3,2,{v1,v2,v3},3) strict"]["error"]["at_level"] == 3
    M = po.Model(3, 2, ("v1", "v2", "v3"), 3)
    with pytest.raises(po.EvalError):
        po.bfs(M, max_depth=4)
