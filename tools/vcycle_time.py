"""Timing aid: V-cycles alone on one built C2 (or argv[1]) hierarchy with per-record HIP-event times; tolerant of kernels that are switched
off for a timing experiment (the iterate is never looked at).  HOT_SOAK_CFG carries hot_config overrides."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd, bench
from hot_amd import parallel, synth
which = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = dict(synth.CONFIGS[which])
cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
lib = hot_amd.HotLib(os.environ["HOT_LIB"]) if os.environ.get("HOT_LIB") else hot_amd.load()  # HOT_LIB: another build of the library
over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("HOT_SOAK_CFG", "").split(",") if kv)}
ctx = bench.make_ctx(lib, cloud, cfg, profile=1, **over)
ctx.sort(), ctx.p2g(), ctx.begin_step(cfg["dt"])
ctx.update_state(ctx.get_dv())
ctx.build_hessian(), ctx.build_mg()
x = ctx.project(np.random.default_rng(1).standard_normal((ctx.Nn, 3)))
ctx.vcycle(x)
ctx.profile_reset()
n = 10
for _ in range(n):
    ctx.vcycle(x)
t = ctx.profile()
for k, v in sorted(t.items(), key=lambda kv: -kv[1]["total_ms"])[:int(os.environ.get("HOT_PROF_TOP", "8"))]:
    print("%-28s calls/vcycle %6.1f  ms/vcycle %8.3f  avg us %8.2f" % (k, v["calls"] / n, v["total_ms"] / n, 1e3 * v["total_ms"] / max(v["calls"], 1)))
