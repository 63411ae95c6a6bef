"""Developer helper: point the package's ctypes binding at another build of the library (tools/build_dev.py) inside ONE
process, so two builds can be timed in interleaved rounds on one device (guide rule 24).  Not used by the package."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noise_robust_vit_amd import _lib

_cache = {}


def path_of(name):
    return _lib._DEFAULT_LIB if name in (None, "", "product") else os.path.join(ROOT, "tools", "_build", f"libnrv_hip_{name}.so")


def use_library(name=None):
    """name = None / "product": the shipped library; otherwise tools/_build/libnrv_hip_<name>.so."""
    p = path_of(name)
    if p not in _cache:
        _cache[p] = _lib.bind(p)
    _lib._lib = _cache[p]
    return _cache[p]
