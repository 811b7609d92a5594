"""Timing aid (not a test): full per-record kernel profile of two C2 time steps."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd, bench
from hot_amd import parallel, synth
lib = hot_amd.load()
cfg = dict(synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"])
cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
ctx = bench.make_ctx(lib, cloud, cfg, profile=1)
ctx.advance(cfg["dt"])
ctx.profile_reset()
st = [ctx.advance(cfg["dt"]) for _ in range(2)]
t = ctx.profile()
tot = sum(v["total_ms"] for v in t.values())
print("iters", [s["iterations"] for s in st], "ms_total", [round(s["ms_total"], 1) for s in st], "kernel ms/step", round(tot / 2, 1),
      {k: round(st[-1][k], 2) for k in ("ms_sort", "ms_p2g", "ms_begin", "ms_hessian", "ms_mg_build", "ms_solve", "ms_g2p")})
for k, v in sorted(t.items(), key=lambda kv: -kv[1]["total_ms"])[:45]:
    print("%-24s calls/step %6.1f  ms/step %8.3f  avg %8.4f ms" % (k, v["calls"] / 2, v["total_ms"] / 2, v["total_ms"] / v["calls"]))
