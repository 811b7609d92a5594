"""Timing experiment (not a test): GS smoother / SpMV / Hessian kernels in isolation on a C2-size system."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from hot_amd import synth
lib = hot_amd.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 63
c = synth.cube_cloud(n, ppc=8)
ctx = lib.context(dtype=1, dx=c["dx"], gravity=(0, -9.8, 0), levelCnt=3, profile=1)
ctx.set_particles(c["X"], c["V"], c["mass"], c["vol"], c["mu"], c["lam"])
o, nn = synth.sticky_floor(5.0, c["dx"]); ctx.set_sticky_halfspaces(o, nn)
ctx.sort(); ctx.p2g(); ctx.begin_step(1 / 24); ctx.update_state(None)
ctx.build_hessian(); ctx.build_mg()
Nn = ctx.Nn
print("Np", c["X"].shape[0], "Nn", Nn, "nnzb", [ctx.level_nnzb(l) for l in range(3)], [ctx.level(l, coords=False)["nrows"] for l in range(3)])
x = np.random.default_rng(0).standard_normal((Nn, 3))
ctx.profile_reset()
for _ in range(3):
    ctx.vcycle(x)
    ctx.spmv(0, x)
ctx.build_hessian()
p = ctx.profile()
for k, v in sorted(p.items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
    print("%-22s calls %5d  total %9.3f ms  avg %8.4f ms" % (k, v["calls"], v["total_ms"], v["total_ms"] / v["calls"]))
