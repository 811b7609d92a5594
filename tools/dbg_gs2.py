"""Debug aid: level-0 GS smoother with half-block kernels against the oracle on a small body."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd
from tests import pipeline_checks as pc
from tests.oracle_lib import load_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = hot_amd.HotLib(hot_amd.AB_LIB_PATH) if os.environ.get("HOT_AMD_AB") else hot_amd.load()
ora = load_oracle()
def built(L, **kw):
    ctx, c = pc.make_ctx(L, n=n, dtype=dtype, **kw)
    pc.prepare(ctx)
    ctx.update_state(ctx.get_dv())
    ctx.build_hessian()
    ctx.build_mg()
    return ctx
g = built(lib, levelCnt=2, gs_chain=1, gs_sub_block=32)
c = built(ora, levelCnt=2)
nn = g.level(0, coords=False)["nrows"]
b = c.project(np.random.default_rng(3).standard_normal((nn, 3)))
print("built, rows", nn, flush=True)
t = time.time()
ug, rg = g.smooth(0, 5, 2, np.zeros_like(b), b, tolerance=0.0)
print("gpu smooth done %.2fs" % (time.time() - t), flush=True)
uc, rc = c.smooth(0, 5, 2, np.zeros_like(b), b, tolerance=0.0)
rel = lambda a, bb: np.abs(a - bb).max() / np.abs(bb).max()
print("rel u %.3e  rel r %.3e" % (rel(ug, uc), rel(rg, rc)), flush=True)
