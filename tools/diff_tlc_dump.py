#!/usr/bin/env python3
"""Pin this checker against TLC itself, for whoever has a JVM (this image has none: DESIGN.md §1, "parity unpinned").

    java -cp tla2tools.jar tlc2.TLC -deadlock -workers 1 -dump states.dump -config VSR.cfg VSR.tla     # elsewhere
    python tools/diff_tlc_dump.py -config VSR.cfg [-tla VSR.tla] states.dump [--max-depth N] [--subset-ok]

Reads TLC's -dump file (State k: + /\\ var = value conjuncts) with the product's own TLC reader (vsrmc_model_parse_states), takes the
canonical VIEW + SYMMETRY fingerprint of every dumped state on the GPU (vsrmc_fingerprint_batch) — TLC and this checker may keep
different value-permuted representatives of a state, the fingerprint does not care —, runs the GPU BFS on the same configuration and
compares the two SETS of states.  Exit code 0: the sets are equal (or, with --subset-ok, every TLC state was found here: a TLC run
that stopped at a violation dumps only part of its last level).  Needs a GPU; any of the three modules the build lowers."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-config", required=True)
    ap.add_argument("-tla", default=None)
    ap.add_argument("dump")
    ap.add_argument("--max-depth", type=int, default=0, help="stop the GPU BFS after this many levels (0 = to exhaustion / first violation)")
    ap.add_argument("--subset-ok", action="store_true", help="succeed when every TLC state was found, whatever else the GPU BFS found")
    ap.add_argument("--table-log2", type=int, default=24)
    ap.add_argument("--frontier-gib", type=float, default=1.0)
    ap.add_argument("--show", type=int, default=3, help="print up to this many states of each difference")
    a = ap.parse_args(argv)
    import vsr_tlaplus_amd as vt
    m = vt.Model.load(a.config, a.tla)
    with open(a.dump) as f:
        states = m.parse_states(f.read())
    if not states:
        print("no state found in %s" % a.dump)
        return 2
    words = np.concatenate([rec for _, rec in states])
    off = np.cumsum([0] + [len(rec) for _, rec in states]).astype(np.uint64)
    fps, _ = m.fingerprints(words, off)
    tlc = {}
    for k, fp in enumerate(fps):
        tlc.setdefault(int(fp), k)                               # first dumped state with this identity
    print("%s: %d states, %d distinct under VIEW%s" % (a.dump, len(states), len(tlc), " + SYMMETRY" if m.layout.symmetry else ""))
    fw = int(a.frontier_gib * (1 << 30) / 8)
    mc = vt.ModelChecker(m, table_log2=a.table_log2, frontier_words=fw, frontier_states=max(1 << 12, fw // 24))
    ours = {int(x): 1 for x in mc.level_fps()}
    while mc.n_frontier and mc.violation is None and (a.max_depth == 0 or mc.level < a.max_depth):
        d = mc.step()
        if d["n_new"]:
            for x in mc.level_fps():
                ours[int(x)] = d["level"]
    stop = "violation at depth %d" % mc.level if mc.violation else ("depth bound" if mc.n_frontier else "state space exhausted")
    print("GPU BFS: %d distinct states, depth %d (%s)" % (len(ours), mc.level, stop))
    only_tlc = [fp for fp in tlc if fp not in ours]
    only_ours = [fp for fp in ours if fp not in tlc]
    print("in both: %d; only in the TLC dump: %d; only in the GPU BFS: %d" % (len(tlc) - len(only_tlc), len(only_tlc), len(only_ours)))
    for fp in only_tlc[: a.show]:
        k = tlc[fp]
        print("\nonly in the TLC dump (its State %d):\n%s" % (k + 1, m.format_state(states[k][1])))
    for fp in sorted(only_ours, key=lambda f: ours[f])[: a.show]:
        print("\nonly in the GPU BFS: level %d, fingerprint %016x" % (ours[fp], fp))
        if ours[fp] <= mc.level:
            try:
                tr = mc.trace_fp(ours[fp], fp)
                print("(reached by: %s)\n%s" % (" -> ".join(act for act, _ in tr[1:]), m.format_state(tr[-1][1])))
            except vt.VsrmcError as e:
                print("(no path: %s)" % e)
    mc.close()
    ok = not only_tlc and (a.subset_ok or not only_ours)
    print("\n%s" % ("The two sets of states are equal." if not only_tlc and not only_ours else
                    "Every TLC state was found by the GPU BFS." if ok else "The sets DIFFER."))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
