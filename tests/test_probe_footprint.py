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
