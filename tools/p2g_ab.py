"""P2G / G2P timing at C2 and C3 size (HIP events on the launch stream): product kernel vs the A/B build's alternatives."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from hot_amd import parallel, synth

which = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = dict(synth.CONFIGS[which])
cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
for label, path, env in (("product", hot_amd.LIB_PATH, {}), ("cells1(ab)", hot_amd.AB_LIB_PATH, {"HOT_P2G_CELLS1": "1"})):
    for k in ("HOT_P2G_CELLS1",):
        os.environ.pop(k, None)
    os.environ.update(env)
    lib = hot_amd.HotLib(path)
    ctx = lib.context(dtype=1 if cfg["dtype"] == np.float64 else 0, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=3, profile=1)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    ctx.sort(), ctx.p2g(), ctx.begin_step(cfg["dt"])
    ctx.profile_reset()
    for _ in range(10):
        ctx.p2g()
    ctx.begin_step(cfg["dt"])
    for _ in range(5):
        ctx.g2p(0.0)
    t = ctx.profile()
    print(which, label, {k: round(v["total_ms"] / v["calls"], 4) for k, v in t.items() if k in ("p2g", "p2g_reduce", "g2p")})
    del ctx
