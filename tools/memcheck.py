"""Check (not a test): device memory in use after 2, 4, 8, 16 time steps (Galerkin and --baseline hierarchies) — buffers are
grow-only and recycled, so the figure must level off."""
import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd, bench
from hot_amd import parallel, synth
hip = ctypes.CDLL("libamdhip64.so")
def used():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return (t.value - f.value) / 2**20
lib = hot_amd.load()
for name, kw in (("C2", {}), ("C2-baseline", dict(useBaselineMultigrid=1))):
    cfg = dict(synth.CONFIGS["C2"])
    cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
    ctx = bench.make_ctx(lib, cloud, cfg, **kw)
    m = []
    for s in range(16):
        ctx.advance(cfg["dt"])
        if s in (1, 3, 7, 15): m.append(round(used()))
    print(name, "MiB used after steps 2,4,8,16:", m)
    del ctx
print("after del:", round(used()))
