"""ctypes binding of oracle/build/liborc3.so — the CPU ORACLE of the THIRD model (VR_APP_STATE.tla; oracle/vras_oracle.cpp
behind the same driver and C API as oracle/orc.py).  TEST INFRASTRUCTURE, NOT PRODUCT CODE."""
from . import orc as _orc
from .orc_analysis import install

install(globals(), "liborc3.so", [("PrimaryExecuteOp" if a == "ExecuteOp" else a) for a in _orc.ACTIONS],   # Next order, VR_APP_STATE.tla:811-831
        default_mask=30, words_per_replica=2, params_doc=(
    "VR_APP_STATE.cfg:4-7 constants; invariant_mask bits: 1 AcknowledgedWriteNotLost, 2 AcknowledgedWritesExistOnMajority, "
    "4 NoLogDivergence, 8 CommitNumberNeverHigherThanOpNumber, 16 NoAppStateDivergence (the shipped cfg checks 2 + 4 + 8 + 16 = 30)"))
