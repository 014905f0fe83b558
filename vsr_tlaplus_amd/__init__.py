"""vsr-tlaplus_amd — MI355X-native explicit-state model checker for Vanlightly/vsr-tlaplus's VSR.tla.

Package contents: csrc/ (HIP kernels + the C ABI of include/vsrmc.h), capi.py (ctypes binding), checker.py (host-side
mirror of the TLC interfaces the path replaces), sharded.py (multi-GPU level loop over torch.distributed), build.py.
`vsr-tlaplus_amd` at the repo root is a symbolic link to this directory (the name the task layout uses; a dash is not importable).
"""
from .capi import VsrmcError, load  # noqa: F401
from .checker import ACTION_NAMES, FPSet, Model, ModelChecker, StateQueue  # noqa: F401
