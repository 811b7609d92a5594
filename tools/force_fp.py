"""Timing aid: state / force / P2G kernels on the C3 scene in fp32 and fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd, bench
from hot_amd import parallel, synth
lib = hot_amd.load()
cfg = dict(synth.CONFIGS["C3"])
for dt_ in (np.float32, np.float64):
    cfg["dtype"] = dt_
    cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
    ctx = bench.make_ctx(lib, cloud, cfg, profile=1)
    ctx.sort(); ctx.p2g(); ctx.begin_step(cfg["dt"])
    ctx.profile_reset()
    for _ in range(5):
        ctx.sort(); ctx.p2g(); ctx.begin_step(cfg["dt"]); ctx.update_state(ctx.get_dv())
    t = ctx.profile()
    print(dt_.__name__, "groups", ctx.counts(), {k: round(v["total_ms"] / v["calls"], 3) for k, v in t.items() if k in ("p2g", "force_scatter", "state_update", "p2g_reduce", "force_reduce", "radix_sort_pairs")})
    del ctx
