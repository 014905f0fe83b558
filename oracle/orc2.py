"""ctypes binding of oracle/build/liborc2.so — the CPU ORACLE of the SECOND model (VR_STATE_TRANSFER.tla; oracle/vrst_oracle.cpp
behind the same driver and C API as oracle/orc.py).  TEST INFRASTRUCTURE, NOT PRODUCT CODE."""
from . import orc as _orc
from .orc_analysis import install

install(globals(), "liborc2.so", _orc.ACTIONS, default_mask=14, words_per_replica=1, params_doc=(
    "VR_STATE_TRANSFER.cfg:4-7 constants; invariant_mask bits: 1 AcknowledgedWriteNotLost, 2 AcknowledgedWritesExistOnMajority, "
    "4 NoLogDivergence, 8 CommitNumberNeverHigherThanOpNumber (the shipped cfg checks 2 + 4 + 8 = 14)"))
