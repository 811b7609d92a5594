"""One C2 time step with the Galerkin hierarchy and with the baseline geometric hierarchy (--baseline): iterations and timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd
from hot_amd import parallel, synth

lib = hot_amd.load()
cfg = synth.CONFIGS["C2"]
for base in (0, 1):
    cloud = parallel.shard_cloud(cfg, 0, 1, n=63)
    ctx = lib.context(dtype=1, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=cfg["levelCnt"], useBaselineMultigrid=base)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    o, nrm = synth.sticky_floor(cloud["corner"][1], cloud["dx"])
    ctx.set_sticky_halfspaces(o, nrm)
    for step in range(2):
        st = ctx.advance(cfg["dt"])
        print("baseline" if base else "galerkin", "step", step, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items() if k in ("iterations", "converged", "vcycles", "linear_iterations", "ms_hessian", "ms_mg_build", "ms_solve", "ms_total")},
              [ctx.level(l, coords=False)["nrows"] for l in range(cfg["levelCnt"])])
    del ctx
