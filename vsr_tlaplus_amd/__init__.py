"""Import shim: the package directory is `vsr-tlaplus_amd/` (not a valid Python identifier); this makes it importable
as `vsr_tlaplus_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vsr-tlaplus_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
