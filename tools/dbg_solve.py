import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from tests.oracle_lib import load_oracle
from tests import pipeline_checks as pc
lib, ora = hot_amd.load(), load_oracle()
res = {}
for name, L in (("cpu", ora), ("gpu", lib)):
    ctx, c = pc.make_ctx(L, n=8, cneps=1e-7, lsolver=3, levelCnt=3, max_iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    pc.prepare(ctx)
    st = ctx.solve()
    print(name, st, flush=True)
    res[name] = ctx.get_dv()
print("dv diff", np.abs(res["cpu"] - res["gpu"]).max(), np.abs(res["cpu"]).max())
