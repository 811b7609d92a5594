"""Soak run (not a test): many consecutive time steps of a BASELINE config; every step must converge and stay finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd, bench
from hot_amd import parallel, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lib = hot_amd.load()
cfg = dict(synth.CONFIGS[name])
ncells = int(os.environ.get("HOT_SOAK_CELLS", cfg["n"]))  # e.g. a per-GPU share of C4 / C5
cloud = parallel.shard_cloud(cfg, 0, 1, n=ncells)
over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("HOT_SOAK_CFG", "").split(",") if kv)}  # e.g. HOT_SOAK_CFG=gs_chain=1
ctx = bench.make_ctx(lib, cloud, cfg, **over)
its, ms = [], []
for s in range(steps):
    st = ctx.advance(cfg["dt"])
    print("step", s, "iterations", st["iterations"], "trials", st["linesearch_trials"], "dropped", st["dropped_pairs"], "E %.6g" % st["energy"], "ms %.1f" % st["ms_total"], flush=True)
    assert st["converged"] == 1, (s, st)
    its.append(st["iterations"]), ms.append(st["ms_total"])
p = ctx.get_particles()
assert np.isfinite(p["X"]).all() and np.isfinite(p["F"]).all()
print(name, "steps", steps, "iterations", its, "ms/step mean %.1f" % (sum(ms) / len(ms)), "min y %.4f" % p["X"][:, 1].min())
