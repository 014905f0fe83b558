#!/usr/bin/env python3
"""Write a TLC configuration for VSR.tla in the layout of the reference's VSR.cfg (constants, model values bound to themselves,
INIT / NEXT / VIEW / SYMMETRY / INVARIANT).    python tools/make_cfg.py OUT R C "v1, v2, v3" L"""
import sys

MODEL_VALUES = ["Normal", "ViewChange", "Recovering", "RequestMsg", "ReplyMsg", "PrepareMsg", "PrepareOkMsg", "CommitMsg",
                "StartViewChangeMsg", "DoViewChangeMsg", "StartViewMsg", "GetStateMsg", "NewStateMsg", "RecoveryMsg",
                "RecoveryResponseMsg", "Nil"]


def text(R, C, values, L):
    lines = ["CONSTANTS", "    ReplicaCount = %d" % R, "    ClientCount = %d" % C, "    Values = {%s}" % values,
             "    StartViewOnTimerLimit = %d" % L, "    RestartEmptyLimit = 0"]
    lines += ["    %s = %s" % (v, v) for v in MODEL_VALUES]
    lines += ["", "INIT Init", "NEXT Next", "", "VIEW view"]
    if "," in values:
        lines.append("SYMMETRY symmValues")
    lines += ["", "INVARIANT", "AcknowledgedWriteNotLost", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    out, R, C, values, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    with open(out, "w") as f:
        f.write(text(R, C, values, L))
