"""Converged solves (tests/test_gpu_solver.py::test_solve_to_convergence_against_oracle) of three configurations against the oracle: mass-weighted
|ddv| / |dv| and the counters, to see how far amplified round-off carries a long solve apart.  HOT_LIB selects another build of the library."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from tests import pipeline_checks as pc
from tests.oracle_lib import load_oracle
lib = hot_amd.HotLib(os.environ["HOT_LIB"]) if os.environ.get("HOT_LIB") else hot_amd.load()
ora = load_oracle()
for kw in (dict(lsolver=3, levelCnt=1), dict(lsolver=3, levelCnt=3), dict(lsolver=2, levelCnt=2)):
    out = {}
    for name, L in (("gpu", lib), ("cpu", ora)):
        ctx, c = pc.make_ctx(L, n=8, cneps=1e-7, **kw)
        pc.prepare(ctx)
        st = ctx.solve()
        out[name] = (ctx.get_dv(), st, ctx.grid()["mass"])
    sg, sc = out["gpu"][1], out["cpu"][1]
    m = out["cpu"][2][:, None]
    a, b = out["gpu"][0], out["cpu"][0]
    err = np.sqrt((m * (a - b) ** 2).sum()) / np.sqrt((m * b ** 2).sum())
    print(kw, "err %.3g" % err, sg["iterations"], sc["iterations"], sg["linesearch_trials"], sc["linesearch_trials"])
