"""How fast the HIP and the oracle MINRES iterates separate behind the lumped-mass preconditioner: relative difference of the Newton step
after a fixed number of Lanczos steps (hot_config.linear_iteration_cap)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from tests.oracle_lib import load_oracle
from tests import pipeline_checks as pc
lib, ora = hot_amd.load(), load_oracle()
for ainv in (2, 1):
    for cap in (1, 2, 3, 5, 10, 20, 40, 80, 150):
        out = {}
        for name, L in (("gpu", lib), ("cpu", ora)):
            ctx, c = pc.make_ctx(L, n=8, cneps=1e-7, max_iterations=1, lsolver=1, levelCnt=1, Ainv=ainv, linear_iteration_cap=cap, linesearch=0)
            pc.prepare(ctx)
            st = ctx.solve()
            out[name] = (ctx.get_dv(), st)
        a, b = out["gpu"][0], out["cpu"][0]
        print("Ainv", ainv, "cap", cap, "linear its", out["gpu"][1]["linear_iterations"], out["cpu"][1]["linear_iterations"], "rel diff %.3e" % (np.abs(a - b).max() / np.abs(b).max()), flush=True)
