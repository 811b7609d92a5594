"""Per-record HIP-event table of whole C2 time steps (all records, calls and ms per step) + host wall clock per step."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd, bench
from hot_amd import parallel, synth
which = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = dict(synth.CONFIGS[which])
cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
lib = hot_amd.HotLib(os.environ["HOT_LIB"]) if os.environ.get("HOT_LIB") else hot_amd.load()  # HOT_LIB: another build of the library
over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("HOT_SOAK_CFG", "").split(",") if kv)}  # e.g. HOT_SOAK_CFG=gs_chain=2
ctx = bench.make_ctx(lib, cloud, cfg, profile=1, **over)
for _ in range(3):
    ctx.advance(cfg["dt"])
ctx.profile_reset()
n = 3
t0 = time.perf_counter()
sts = [ctx.advance(cfg["dt"]) for _ in range(n)]
ctx.sync()
wall = (time.perf_counter() - t0) * 1e3 / n
t = ctx.profile()
tot = sum(v["total_ms"] for v in t.values()) / n
print(which, "wall ms/step %.2f  sum of kernel ms/step %.2f  iterations/step %.1f" % (wall, tot, sum(s["iterations"] for s in sts) / n))
top = int(os.environ.get("HOT_PROF_TOP", "1000"))
for k, v in sorted(t.items(), key=lambda kv: -kv[1]["total_ms"])[:top]:
    print("%-28s calls/step %7.1f  ms/step %8.3f  avg us %8.2f" % (k, v["calls"] / n, v["total_ms"] / n, 1e3 * v["total_ms"] / max(v["calls"], 1)))
