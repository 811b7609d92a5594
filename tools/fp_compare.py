"""Experiment (not a test): the C3 scene (8 M particles) in fp32 and in fp64 — iterations per step and time, to see how much of
the fp32 iteration count is rounding noise at cneps = 1e-7."""
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
    ctx = bench.make_ctx(lib, cloud, cfg)
    out = []
    for s in range(3):
        st = ctx.advance(cfg["dt"])
        out.append((st["iterations"], st["linesearch_trials"], round(st["ms_total"])))
    print(dt_.__name__, out)
    del ctx
