"""The probe level runs only the actions inside the invariants' footprint (ModelOps<0>::probe_actions, csrc/vsr_kernels.hpp): the invariants
(VSR.tla:933-950) read rep_log and aux_client_acked only, so a successor whose action writes neither has the verdict of its (passing) parent.
Held here against the ORACLE's successors: over whole small state spaces and the first levels of the BASELINE configurations, every successor
by an action outside the set leaves every replica's log and the acknowledged map untouched and has its parent's invariant verdict; the actions
inside the set do change them (the set is not vacuous).  The GPU side of the claim is the probe fixtures: violating successors and smallest
violating fingerprint of the probed levels equal the oracle's (tests/test_gpu_parity.py, bench.py)."""
import pytest

PROBE_ACTIONS = {"SendSV", "ExecuteOp", "ReceiveClientRequest", "SendGetState", "ReceiveSV", "ReceivePrepareMsg", "ReceiveGetState", "ReceiveNewState"}


def _logs_and_acked(P, rec):
    wpr = P.wpr()
    logs = tuple(int(rec[1 + r * wpr + 1]) & 0xFFFFFF for r in range(P.R))        # x0 of every replica block = its log (three entry bytes)
    acked = (int(rec[0]) >> 11) & ((1 << (2 * P.n)) - 1)                           # two bits per value: absent / FALSE / TRUE
    return logs, acked


@pytest.mark.parametrize("cfg,depth,mask", [((3, 1, 1, 1), 40, 3), ((3, 1, 2, 2), 10, 3), ((3, 1, 3, 3), 8, 3), ((5, 1, 2, 2), 6, 3), ((3, 2, 2, 1), 7, 3)])
def test_actions_outside_the_footprint_keep_the_verdict(cfg, depth, mask):
    from oracle import orc
    R, C_, n, L = cfg
    P = orc.Params(R, C_, n, L, invariant_mask=mask, assume_commit_number=(C_ > 1))
    b = orc.Bfs(P)
    seen_in, changed_in, outside = set(), set(), 0
    for _ in range(depth):
        w, off = b.frontier()
        step = max(1, (len(off) - 1) // 1500)
        for i in range(0, len(off) - 1, step):
            rec = w[int(off[i]):int(off[i + 1])]
            parent = _logs_and_acked(P, rec)
            pinv = orc.invariants(P, rec)
            for s in orc.successors(P, rec):
                name = orc.ACTIONS[s["action"]]
                child = _logs_and_acked(P, s["words"])
                if name in PROBE_ACTIONS:
                    seen_in.add(name)
                    if child != parent:
                        changed_in.add(name)
                else:
                    outside += 1
                    assert child == parent, (name, i)
                    assert s["inv"] == pinv, (name, i)
        try:
            if b.step() <= 0:
                break
        except orc.OracleError:
            break                                                                     # a TLC evaluation error ends the search (two clients, VSR.tla:421)
    assert outside > 1000
    # the writers of rep_log / aux_client_acked that occur this early do write them
    assert {"ReceiveClientRequest", "ReceivePrepareMsg", "ExecuteOp"} <= changed_in, changed_in


def test_the_set_in_the_kernel_source_is_this_one():
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vsr_tlaplus_amd", "csrc", "vsr_kernels.hpp")).read()
    body = src[src.index("static VSR_HD u32 probe_actions()"):]
    body = body[:body.index("}")]
    assert set(re.findall(r"A_(\w+)", body)) == PROBE_ACTIONS


@pytest.mark.gpu
def test_a_probe_pass_beside_a_representation_limit_is_run_again_with_every_action(tmp_path):
    """Round-5 review: a probe level applies only the footprint actions, so a representation limit (bag capacity, delivery count) hit by a successor
    of ANOTHER action went unreported.  Now the kernel counts the instances it did not apply in tiles that hold a record at such a limit
    (LevelCtl::limit_unchecked) and the host runs the pass again with every action applied; vsrmc_level_info.limit_rechecked says so.  No shipped
    configuration comes near a limit, so the hooks library lowers the bag capacity (VSRMC_TEST_MAX_BAG) to the largest bag the space holds: the
    probe of the level after the one that reaches it must be re-run, and must report what the unconstrained probe reports."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "limit_probe.py"
    script.write_text("""
import os, sys
sys.path.insert(0, %r)
import vsr_tlaplus_amd as vt
kw = dict(device=0, table_log2=16, frontier_words=1 << 20, frontier_states=1 << 15, pending_entries=1 << 15)
m = vt.Model.from_constants(R=2, C_=1, n=2, L=2)
mc = vt.ModelChecker(m, **kw)
bags = [0]
while True:
    d = mc.step()
    if not d["n_new"]:
        break
    bags.append(d["max_bag"])
B = max(bags)
L = bags.index(B) + 1                       # the first level that holds a record with the largest bag (Init = level 1)
assert 4 <= B < 40 and L < len(bags)
def probe_at(model):
    c = vt.ModelChecker(model, **kw)
    while c.level < L:
        c.step()
    p = c.probe()
    c.close()
    return p
p0 = probe_at(m)
os.environ["VSRMC_TEST_MAX_BAG"] = str(B)
m2 = vt.Model.from_constants(R=2, C_=1, n=2, L=2)
assert m2.layout.max_bag == B, (m2.layout.max_bag, B)
p1 = probe_at(m2)
assert p0["limit_rechecked"] == 0 and p1["limit_rechecked"] > 0, (p0["limit_rechecked"], p1["limit_rechecked"])
for k in ("level", "generated", "deadlocks", "viol_mask", "viol_fp", "pending"):
    assert p0[k] == p1[k], (k, p0[k], p1[k])
print("OK", B, L, p1["limit_rechecked"])
""" % root)
    hooks = os.path.join(root, "vsr_tlaplus_amd", "libvsrmc_hooks.so")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=dict(os.environ, VSRMC_LIB=hooks))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
