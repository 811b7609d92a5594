"""Hessian assembly time (product library, HIP events from the context profile) at C2 / C3 size, and a checksum of A x for a fixed random x."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from hot_amd import parallel, synth

for which in (sys.argv[1:] or ["C2", "C3"]):
    cfg = dict(synth.CONFIGS[which])
    cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
    lib = hot_amd.HotLib(os.environ["HOT_LIB"]) if os.environ.get("HOT_LIB") else hot_amd.load()
    ctx = lib.context(dtype=1 if cfg["dtype"] == np.float64 else 0, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=1, profile=1)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    ctx.sort(), ctx.p2g(), ctx.begin_step(cfg["dt"])
    ctx.update_state(ctx.get_dv())
    x = np.random.default_rng(1).standard_normal((ctx.Nn, 3))
    ctx.build_hessian()
    ctx.profile_reset()
    for _ in range(3):
        ctx.build_hessian()
    t = ctx.profile()
    y = ctx.spmv(0, x).astype(np.float64)
    print(which, {k: round(v["total_ms"] / v["calls"], 3) for k, v in t.items() if k.startswith("hessian")}, "|Ax| %.12e" % np.linalg.norm(y))
    del ctx
