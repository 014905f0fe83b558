"""Where a probe level's time goes: the README configuration is run to level 21 (stored), then level 22 is probed from it (vsrmc_checker_probe).
Prints the shader-clock phase split (wave 0 of every block: stage, enumerate, sort, apply, tail) of the last stored level's expansion and of the
probe pass, with their kernel times.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt


def split(d):
    pc = [int(x) for x in d["phase_cycles"][:5]]
    tot = float(sum(pc)) or 1.0
    out = dict(zip(("stage", "enumerate", "sort", "apply", "tail"), [round(x / tot, 3) for x in pc]))
    # thread 0's clock inside the apply loop: gen, hash, seen-set probe + claim, successor write (the last one arrives in act_generated[0])
    inner = [int(d["phase_cycles"][5]), int(d["phase_cycles"][6]), int(d["phase_cycles"][7]), int(d["act_generated"][0])]
    it = float(sum(inner)) or 1.0
    out["apply_split"] = dict(zip(("gen", "hash", "probe_claim", "write"), [round(x / it, 3) for x in inner]))
    return out


def main():
    m = vt.Model.from_constants(R=3, C_=1, n=3, L=3)
    mc = vt.ModelChecker(m, table_log2=32, frontier_words=int(12.8e9), frontier_words_b=int(7.0e9), frontier_states=int(2.85e8), pending_entries=1 << 16, keep_trace=False)
    last = None
    while mc.level < 21:
        last = mc.step()
    p = mc.probe()
    out = dict(stored_level=dict(level=last["level"], parents=int(last["frontier"]), generated=int(last["generated"]), expand_ms=round(last["expand_ms"], 2), phases=split(last)),
               probed_level=dict(level=p["level"], parents=int(p["frontier"]), generated=int(p["generated"]), expand_ms=round(p["expand_ms"], 2), phases=split(p),
                                 probes=int(p["probes"]), viol_mask=int(p["viol_mask"])))
    out["per_parent_ns"] = dict(stored=round(1e6 * last["expand_ms"] / last["frontier"], 3), probed=round(1e6 * p["expand_ms"] / p["frontier"], 3))
    print(json.dumps(out))
    mc.close()


if __name__ == "__main__":
    main()
