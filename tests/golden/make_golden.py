#!/usr/bin/env python3
"""tests/golden/make_golden.py — regenerates the golden fixtures in this directory.

Run in the build container (needs /root/reference, which does NOT exist on the GPU box):
    python tests/golden/make_golden.py [--deep]

Outputs
  state_transfer_trace.json   the reference's only golden vector (/root/reference/state_transfer_violation_trace.txt,
                              24 states, README defect config R=3 C=1 Values={v1,v2,v3} L=3) converted to packed
                              records.  Every step is first checked to be a legal step of the current VSR.tla by the
                              independent Python restatement (oracle/pyoracle.py).  For each state we also store the
                              SHA-256 of every printed `var |-> value` line of the reference file, so the product's
                              TLC-format printer can be pinned to the reference text without copying it.
  bfs_counts.json             per-level (new, generated, xor and sum of the canonical fingerprints) of the C++ oracle's
                              BFS for the BASELINE configs, to the depth the CPU finishes in about a minute each.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc, pycodec, pyoracle as po, tlcvalue  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = "/root/reference/state_transfer_violation_trace.txt"
M64 = (1 << 64) - 1


def make_trace():
    text = open(TRACE).read()
    tr = tlcvalue.parse_trace(text)
    M = po.Model(3, 1, ("v1", "v2", "v3"), 3)
    P = orc.Params(3, 1, 3, 3)
    vars17 = sorted(tr[0][2].keys())

    def proj(s):
        return po.canon({k: s[k] for k in vars17})

    # the printed lines per state, in file order
    blocks, cur = [], None
    for line in text.splitlines():
        if line.startswith(" _TEAction"):
            cur = {}
            blocks.append(cur)
        elif cur is not None and " |-> " in line and not line.startswith(" "):
            var = line.split(" |-> ", 1)[0]
            cur[var] = hashlib.sha256(line.rstrip().rstrip(",").encode()).hexdigest()
    assert len(blocks) == len(tr) == 24

    states = []
    cur_state = po.Init(M)
    assert proj(cur_state) == po.canon(tr[0][2])
    for i, (name, pos, st) in enumerate(tr):
        if i > 0:
            hits = [(n, t) for n, t in po.successors(M, cur_state) if proj(t) == po.canon(st)]
            assert len(hits) == 1 and hits[0][0] == name, (i, name, [h[0] for h in hits])
            cur_state = hits[0][1]
        words = pycodec.pack(M, cur_state)
        fp, ak = orc.fingerprint(P, words)
        states.append(dict(position=pos, action=name, words=["%016x" % w for w in words], fp="%016x" % fp, auxkey=ak,
                           inv_mask=orc.invariants(P, words),
                           holds=po.AcknowledgedWriteNotLost(M, cur_state), line_sha256=blocks[i]))
    out = dict(source="reference state_transfer_violation_trace.txt (README defect config); derived, not a copy",
               params=dict(R=3, C=1, values=["v1", "v2", "v3"], L=3), variables=vars17, states=states)
    with open(os.path.join(HERE, "state_transfer_trace.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("trace fixture: %d states, invariant holds: %s" % (len(states), [s["holds"] for s in states].count(True)))


def bfs_counts(P, max_depth, label):
    b = orc.Bfs(P)
    levels = []
    fps = b.level_fps(1)
    levels.append(dict(level=1, new=1, generated=0, fp_xor="%016x" % int(fps[0]), fp_sum="%016x" % int(fps[0]), ties=0,
                       deadlocks=0))
    err = None
    while b.info["depth"] < max_depth:
        try:
            nn = b.step()
        except orc.OracleError as e:
            err = dict(code=e.code, message=str(e), at_level=b.info["depth"] + 1)
            break
        if nn == 0:
            break
        fps = b.level_fps(b.info["depth"])
        x, s = 0, 0
        for v in fps.tolist():
            x ^= v
            s = (s + v) & M64
        levels.append(dict(level=b.info["depth"], new=int(nn), generated=b.info["generated"], fp_xor="%016x" % x,
                           fp_sum="%016x" % s, ties=b.info["ties"], deadlocks=b.info["deadlocks"]))
        print(label, levels[-1], flush=True)
    res = dict(label=label, params=dict(R=P.R, C=P.C, n=P.n, L=P.L, symmetry=bool(P.arr[6])),
               exhausted=bool(err is None and b.info["n_new"] == 0), distinct=b.info["distinct"] if levels else 0,
               depth=b.info["depth"], max_bag=b.info["max_bag"], viol_mask=b.info["viol_mask"], error=err, levels=levels)
    if err is not None:
        res["distinct"] = sum(l["new"] for l in levels)
    b.close()
    return res


def make_counts(deep):
    out = []
    out.append(bfs_counts(orc.Params(2, 1, 1, 1), 10 ** 6, "config1 (2,1,{v1},1)"))
    out.append(bfs_counts(orc.Params(2, 1, 2, 2), 10 ** 6, "(2,1,{v1,v2},2)"))
    out.append(bfs_counts(orc.Params(2, 1, 2, 2, symmetry=False), 10 ** 6, "(2,1,{v1,v2},2) no symmetry"))
    out.append(bfs_counts(orc.Params(2, 1, 2, 3, symmetry=False), 10 ** 6, "(2,1,{v1,v2},3) no symmetry"))
    out.append(bfs_counts(orc.Params(3, 1, 2, 2), 17 if deep else 14, "config2 (3,1,{v1,v2},2)"))
    out.append(bfs_counts(orc.Params(3, 1, 3, 3), 12 if deep else 10, "config3 (3,1,{v1,v2,v3},3)"))
    out.append(bfs_counts(orc.Params(3, 2, 3, 3), 6, "config4 (3,2,{v1,v2,v3},3) strict"))
    out.append(bfs_counts(orc.Params(3, 2, 3, 3, assume_commit_number=True), 8, "config4 assume-commit-number"))
    out.append(bfs_counts(orc.Params(5, 1, 2, 2), 8 if deep else 7, "config5 (5,1,{v1,v2},2)"))
    with open(os.path.join(HERE, "bfs_counts.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    make_trace()
    make_counts("--deep" in sys.argv)
