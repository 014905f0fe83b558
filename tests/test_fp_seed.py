"""Second-hash audit (TLC: a run repeated with another `-fp N` polynomial).

Oracle and HIP path key their seen-sets by the SAME 64-bit function, so a false merge of two states would drop one of them from
both sides and every count would still agree.  `fp_seed` (vsrmc_model_set_fp_seed / oracle `set_fp_seed`) xors a seed into every
salt of the view hash: an independent member of the same hash family.  A collision under one seed is (up to 2^-64) not a collision
under another, so any count that depends on the fingerprint function shows up as a difference between two seeds.

CPU part: the seeded oracle (fingerprints change, counts do not).  GPU part: the seeded HIP path equals the seeded oracle set by set
on small spaces, and the whole config-2 workload has the fixture's per-level counts under other seeds."""

import numpy as np
import pytest

SEEDS = [0x5EED5EED5EED5EED, 0x0123456789ABCDEF]


@pytest.fixture()
def orc_seeded():
    from oracle import orc
    yield orc
    orc.set_fp_seed(0)


def _bfs_counts(orc, P, depth):
    ob = orc.Bfs(P)
    rows, sets = [], [set(int(x) for x in ob.level_fps(1))]
    for _ in range(depth):
        n = ob.step()
        if n == 0:
            break
        rows.append((n, ob.info["generated"], ob.info["deadlocks"], ob.info["distinct"]))
        sets.append(set(int(x) for x in ob.level_fps(ob.info["depth"])))
    ob.close()
    return rows, sets


def test_oracle_seed_changes_fingerprints_not_counts(orc_seeded, golden_trace):
    orc = orc_seeded
    P = orc.Params(2, 1, 2, 2)
    base_rows, base_sets = _bfs_counts(orc, P, 60)
    assert base_rows[-1][3] == 2073                                        # the whole space of (2,1,{v1,v2},2)
    for seed in SEEDS:
        orc.set_fp_seed(seed)
        assert orc.fp_seed() == seed
        rows, sets = _bfs_counts(orc, P, 60)
        assert rows == base_rows                                           # n_new, generated, deadlocks, distinct per level
        assert all(len(a) == len(b) for a, b in zip(sets, base_sets))
        assert not (set().union(*sets) & set().union(*base_sets))           # not one fingerprint in common
    # the golden trace's states: 24 distinct fingerprints under every seed, symmetric states still share one
    recs = [np.array([int(w, 16) for w in st["words"]], dtype=np.uint64) for st in golden_trace["states"]]
    gp = golden_trace["params"]
    Pg = orc.Params(gp["R"], gp["C"], len(gp["values"]), gp["L"])
    seen = []
    for seed in [0] + SEEDS:
        orc.set_fp_seed(seed)
        fps = [orc.fingerprint(Pg, r)[0] for r in recs]
        assert len(set(fps)) == len(recs)
        seen.append(fps)
    assert seen[0] != seen[1] != seen[2]
    orc.set_fp_seed(0)
    assert [orc.fingerprint(Pg, r)[0] for r in recs] == seen[0]            # seed 0 = the fixtures' function


def test_analysis_oracles_take_the_seed():
    from oracle import orc2, orc3
    for o, P in ((orc2, orc2.Params(2, 2, 1)), (orc3, orc3.Params(2, 2, 1))):
        try:
            base = _bfs_counts(o, P, 12)
            o.set_fp_seed(SEEDS[0])
            other = _bfs_counts(o, P, 12)
            assert other[0] == base[0] and not (set().union(*other[1]) & set().union(*base[1]))
        finally:
            o.set_fp_seed(0)


@pytest.mark.gpu
@pytest.mark.parametrize("params", [(2, 1, 2, 2), (3, 1, 3, 3), (3, 1, 2, 2)])
def test_seeded_hip_path_equals_seeded_oracle(orc_seeded, params):
    """level fingerprint SETS under a seed: HIP = oracle (the seed reaches every incremental term of hash_child, the staged parents'
    canonical fingerprints and the Init record's hash_full)."""
    import vsr_tlaplus_amd as vt
    orc = orc_seeded
    R, C_, n, L = params
    depth = 40 if params == (2, 1, 2, 2) else 9
    for seed in SEEDS:
        m = vt.Model.from_constants(R=R, C_=C_, n=n, L=L).set_fp_seed(seed)
        assert m.fp_seed == seed
        mc = vt.ModelChecker(m, table_log2=20, frontier_words=1 << 22, frontier_states=1 << 17, pending_entries=1 << 16)
        orc.set_fp_seed(seed)
        ob = orc.Bfs(orc.Params(R, C_, n, L))
        for lvl in range(1, depth + 1):
            assert np.array_equal(mc.level_fps(), ob.level_fps(lvl)), (seed, lvl)
            d = mc.step()
            nn = ob.step()
            assert (d["n_new"], d["generated"], d["deadlocks"]) == (nn, ob.info["generated"], ob.info["deadlocks"])
            if nn == 0:
                break
        mc.close()
        ob.close()


@pytest.mark.gpu
def test_whole_config2_workload_under_other_seeds(oracle_levels):
    """319 228 361 distinct states, 28 levels: every per-level count of the CPU oracle's fixture under two more hash functions — a
    false merge under the fixtures' function (n^2 / 2^65 = 2.8e-3 for this run) would have to repeat under both."""
    import vsr_tlaplus_amd as vt
    g = oracle_levels["config2"]
    p = g["params"]
    for seed in SEEDS:
        m = vt.Model.from_constants(R=p["R"], C_=p["C"], n=p["n"], L=p["L"], symmetry=p["symmetry"], invariant_mask=p["inv_mask"]).set_fp_seed(seed)
        mc = vt.ModelChecker(m, table_log2=30, frontier_words=int(3.6e9), frontier_states=int(1.1e8), pending_entries=1 << 15, keep_trace=False)
        for lv in g["levels"][1:]:
            d = mc.step()
            assert (d["level"], d["n_new"], d["generated"], d["deadlocks"], d["max_bag"]) == \
                (lv["level"], lv["new"], lv["generated"], lv["deadlocks"], lv["max_bag"]), (seed, lv["level"])
            assert [int(x) for x in d["act_generated"][1:16]] == lv["act_generated"][1:16], (seed, lv["level"])
        assert mc.distinct == g["distinct"] and mc.violation is not None and mc.violation["mask"] == g["viol_mask"]
        tr = mc.trace_fp(mc.violation["level"], mc.violation["fp"])
        assert len(tr) == 28                                                # a shortest counter-example whatever the tie-breaks
        mc.close()


@pytest.mark.gpu
def test_checkpoint_of_another_seed_is_refused(tmp_path):
    import vsr_tlaplus_amd as vt
    m1 = vt.Model.from_constants(R=3, C_=1, n=2, L=2).set_fp_seed(SEEDS[0])
    mc = vt.ModelChecker(m1, table_log2=20, frontier_words=1 << 22, frontier_states=1 << 17, pending_entries=1 << 16)
    for _ in range(6):
        mc.step()
    path = str(tmp_path / "seeded.chk")
    mc.save(path)
    mc.close()
    kw = dict(table_log2=20, frontier_words=1 << 22, frontier_states=1 << 17, pending_entries=1 << 16)
    again = vt.ModelChecker(m1, recover=path, **kw)
    assert again.level == 7
    again.close()
    with pytest.raises(vt.VsrmcError):
        vt.ModelChecker(vt.Model.from_constants(R=3, C_=1, n=2, L=2), recover=path, **kw)
