"""Rate of the TLC-style fingerprint mode (FP64 over TLC's serialisation of the view; csrc/vsr_tlcfp.hpp) against the checker's own incremental
fingerprint: the shipped configuration is run to a level of a few million states, then k_tlc_fingerprints is timed over that level's frontier in HBM.
Prints one JSON line.  usage: python tools/tlc_fp_rate.py [level]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
    mc = vt.ModelChecker(m, table_log2=27, frontier_words=1 << 30, frontier_states=1 << 25, pending_entries=1 << 26, keep_trace=False)
    info = None
    while mc.level < level:
        info = mc.step()
    w, off = mc.frontier()
    sample = w[int(off[0]):int(off[1])]
    nbytes = len(m.tlc_view_bytes(sample, 0))
    times = []
    for _ in range(4):
        _, ms = mc.tlc_level_fps(fetch=False)
        times.append(ms)
    fps, _ = mc.tlc_level_fps()
    out = dict(level=mc.level, states=int(info["n_new"]), distinct_fp64=int(len(np.unique(fps))), kernel_ms=times, view_bytes_of_one_state=nbytes,
               states_per_s=info["n_new"] / (min(times) * 1e-3), expand_ms_of_the_level=info.get("expand_ms"),
               note="k_tlc_fingerprints: one lane per record, representative by compareTo + one table-driven FP64 pass over ~%d bytes" % nbytes)
    print(json.dumps(out))
    mc.close()


if __name__ == "__main__":
    main()
