"""hot_amd — MI355X-native (gfx950 HIP) implementation of the HOT per-timestep hot path.

The product is the C-ABI shared library hot_amd/csrc/libhotmi355x.so (include/hot_mi355x.h); this package
is only its Python host-side mirror (ctypes).  There is NO CPU fallback: `load()` raises if the HIP
extension has not been built, and hot_create() fails on a box without a GPU."""
import os

from .binding import Context, HotError, HotLib, hot_config, hot_stats, ABI_SYMBOLS, PRODUCT_ONLY_SYMBOLS  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhotmi355x.so")
AB_LIB_PATH = os.path.join(_HERE, "csrc", "libhotmi355x_ab.so")  # -DHOT_AB_KERNELS build, tests/test_gpu_variants.py only
_lib = None


def build(jobs=8):
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-s", f"-j{jobs}", "-C", os.path.join(_HERE, "csrc"), "all"])
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        # HOT_AMD_AB=1 (set by tests/test_gpu_variants.py for its subprocesses) selects the A/B build of the same sources
        path = AB_LIB_PATH if os.environ.get("HOT_AMD_AB") else LIB_PATH
        if not os.path.exists(path):
            raise HotError(f"{path} is missing: build it with `python -c 'import hot_amd; hot_amd.build()'` "
                           "(there is no CPU fallback)")
        _lib = HotLib(path, prefix="hot_")
    return _lib
