import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, hot_amd, bench
from hot_amd import parallel, synth
lib = hot_amd.load()
cfg = dict(synth.CONFIGS["C2"])
cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
for dxs in (1,):
    ctx = lib.context(dtype=1, dx=dxs*cloud["dx"], gravity=(0,-9.8,0), levelCnt=3, profile=1)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    ctx.sort(); 
    try:
        ctx.p2g()
    except Exception as e:
        print("err", e)
    ctx.profile_reset()
    for _ in range(5):
        try: ctx.p2g()
        except Exception as e: pass
    t = ctx.profile()
    print("dx sign", dxs, {k: round(v["total_ms"]/v["calls"],4) for k,v in t.items() if k.startswith("p2g")})
