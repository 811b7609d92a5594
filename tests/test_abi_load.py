"""CPU-side checks of the drop-in boundary: the HIP library builds in-tree, loads, and exports every symbol
include/hot_mi355x.h declares; without a GPU hot_create fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import hot_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "hot_mi355x.h")).read()
    declared = set(re.findall(r"\b(hot_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hot_ctx"}
    assert {"hot_" + s for s in hot_amd.ABI_SYMBOLS} == declared
    if not os.path.exists(hot_amd.LIB_PATH):
        hot_amd.build()
    lib = hot_amd.load()
    for s in declared:
        assert hasattr(lib.lib, s), s
    assert "gfx950" in lib.version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        return
    lib = hot_amd.load()
    cfg = lib.default_config()
    h = C.c_void_p()
    rc = lib.fn["create"](C.byref(cfg), C.byref(h))
    assert rc != 0 and not h


def test_product_never_touches_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hot_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and "oracle" not in txt.replace("no CPU fallback", "").lower(), (dirpath, f)
