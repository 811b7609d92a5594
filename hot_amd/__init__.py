"""hot_amd — MI355X-native (gfx950 HIP) implementation of the HOT per-timestep hot path.

The product is the C-ABI shared library hot_amd/csrc/libhotmi355x.so (include/hot_mi355x.h); this package
is only its Python host-side mirror (ctypes).  There is NO CPU fallback: `load()` raises if the HIP
extension has not been built, and hot_create() fails on a box without a GPU."""
import os

from .binding import Context, HotError, HotLib, hot_config, hot_stats, ABI_SYMBOLS  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhotmi355x.so")
_lib = None


def build(jobs=8):
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-s", f"-j{jobs}", "-C", os.path.join(_HERE, "csrc")])
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HotError(f"{LIB_PATH} is missing: build it with `python -c 'import hot_amd; hot_amd.build()'` "
                           "(there is no CPU fallback)")
        _lib = HotLib(LIB_PATH, prefix="hot_")
    return _lib
