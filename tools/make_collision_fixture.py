#!/usr/bin/env python3
"""tests/golden/model2_fp_collision.json from a collision hunt of the memory-lean CPU oracle (oracle/vsr_oracle_lean.cpp, --hunt-seed):
the two reachable states of VR_STATE_TRANSFER (3, {v1,v2}, 2) that the repository's default fingerprint function (version 2, seed 0)
maps to one 64-bit value.  Everything in the fixture is re-derived here with the oracle's Python binding — the hunt only says where to look.

    VSR_ORACLE_FP_SEED=5eed5eed5eed5eed oracle/build/vrst_oracle_lean 3 1 2 2 --base-level 19 --slots 2000000000 --max-depth 27 \\
        --threads 8 --inv-mask 14 --no-symmetry --hunt-seed 0 --hunt-slots 2000000000 > hunt.jsonl
    python tools/make_collision_fixture.py hunt.jsonl "<that command line>"
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OTHER_SEED = 0x5EED5EED5EED5EED


def main():
    from oracle import orc2
    path, cmd = sys.argv[1], sys.argv[2]
    rows = [json.loads(l) for l in open(path) if l.startswith('{"fp_collision')]
    first = [r for r in rows if r.get("fp_collision")]
    assert first, "the hunt reported no collision"
    audit = first[0]["audit_fp"]
    members = {}
    for r in rows:
        if r["audit_fp"] == audit:
            members.setdefault(tuple(r["words"]), r)
    assert len(members) >= 2, "only one member of the pair was met again: rerun with --dump-audit-fp %s from the start" % audit
    P = orc2.Params(3, 2, 2, invariant_mask=14)
    states = []
    for words, r in sorted(members.items(), key=lambda kv: (kv[1]["level"], kv[0])):
        rec = np.array([int(w, 16) for w in words], dtype=np.uint64)
        assert tuple(int(x) for x in orc2.normalise(P, rec)) == tuple(int(x) for x in rec)      # a well-formed record in the codec's normal form
        orc2.set_fp_seed(0)
        f0 = orc2.fingerprint(P, rec)[0]
        orc2.set_fp_seed(OTHER_SEED)
        f1 = orc2.fingerprint(P, rec)[0]
        orc2.set_fp_seed(0)
        states.append(dict(level=r["level"], words=list(words), fp_seed0="%016x" % f0, fp_other_seed="%016x" % f1, invariants=int(orc2.invariants(P, rec))))
    assert len({s["fp_seed0"] for s in states}) == 1, "the members differ under seed 0 (only the audit set's 63 compared bits agree)"
    assert len({s["fp_other_seed"] for s in states}) == len(states)
    assert len({tuple(s["words"]) for s in states}) == len(states)
    out = dict(source="oracle/vsr_oracle_lean.cpp collision hunt (CPU oracle; nothing from the GPU path): `%s`; re-derived by tools/make_collision_fixture.py" % cmd,
               model="VR_STATE_TRANSFER", params=dict(R=3, n=2, L=2, inv_mask=14), fp_version=2, other_seed="%016x" % OTHER_SEED,
               fp_seed0=states[0]["fp_seed0"], states=states)
    dst = os.path.join(ROOT, "tests", "golden", "model2_fp_collision.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote %s: %d states share %s under seed 0 (levels %s)" % (dst, len(states), states[0]["fp_seed0"], [s["level"] for s in states]))


if __name__ == "__main__":
    main()
