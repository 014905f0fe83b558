"""vsr-tlaplus_amd — MI355X-native explicit-state model checker for Vanlightly/vsr-tlaplus's VSR.tla.

Package contents: csrc/ (HIP kernels + the C ABI of include/vsrmc.h), capi.py (ctypes binding), checker.py (host-side
mirror of the TLC interfaces the path replaces), sharded.py (multi-GPU level loop over torch.distributed), build.py.
Importable as `vsr_tlaplus_amd` through the shim at the repo root.
"""
from .capi import VsrmcError, load  # noqa: F401
from .checker import ACTION_NAMES, FPSet, Model, ModelChecker, StateQueue  # noqa: F401
