import numpy as np, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import hot_amd
from hot_amd import synth
import pipeline_checks as pc
import oracle_lib
hotlib = hot_amd.load(); oracle = oracle_lib.load_oracle()
T = np.float32
c = synth.cube_cloud(8, ppc=8, dtype=T)
state = dict(X=c["X"], V=c["V"], C_=None, F=None)
o, nrm = synth.sticky_floor(5.0, c["dx"])
def ctx_for(lib, **kw):
    ctx = lib.context(dtype=0, dx=c["dx"], gravity=(0, -9.8, 0), levelCnt=2, **kw)
    ctx.set_particles(state["X"], state["V"], c["mass"], c["vol"], c["mu"], c["lam"], C_=state["C_"], F=state["F"])
    ctx.set_sticky_halfspaces(o, nrm)
    return ctx
DT = 0.03
for step in range(3):
    for its in (1, 2, 3, 4):
        res = {}
        for name, lib in (("gpu", hotlib), ("cpu", oracle)):
            ctx = ctx_for(lib, max_iterations=its, cneps=1e-7)
            pc.prepare(ctx, DT)
            st = ctx.solve()
            res[name] = (ctx.get_dv().astype(np.float64), st)
        (dg, sg), (dc, sc) = res["gpu"], res["cpu"]
        err = np.abs(dg - dc).max() / np.abs(dc).max()
        keys = ("iterations", "linesearch_trials", "dropped_pairs", "vcycles", "linear_iterations", "energy")
        print("step", step, "its", its, "err %.3g" % err, [sg[k] for k in keys], [sc[k] for k in keys], flush=True)
    full = ctx_for(hotlib, max_iterations=300, cneps=1e-4)
    stf = full.advance(DT)
    print("full", stf["iterations"], stf["converged"], stf["dropped_pairs"], stf["energy"])
    p = full.get_particles()
    state = dict(X=p["X"], V=p["V"], C_=p["C"], F=p["F"])
