"""Stress aid (not a test): repeat the fp32 three-step scenario of tests/test_gpu_solver.py and report failures."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from tests import pipeline_checks as pc
lib = hot_amd.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fails = 0
for r in range(reps):
    ctx, c = pc.make_ctx(lib, n=8, dtype=0, levelCnt=2, cneps=1e-4)
    its = []
    try:
        import time
        for s in range(3):
            t0 = time.time()
            st = ctx.advance(1.0 / 24)
            its.append((st["iterations"], st["converged"], st["linesearch_trials"], round(time.time() - t0, 2)))
        if max(i[0] for i in its) > 60 or min(i[1] for i in its) == 0:
            print("rep", r, "SLOW", its, flush=True)
    except Exception as e:
        fails += 1
        print("rep", r, "step", len(its), "its", its, "ERR", e)
        try:
            for lv in range(2):
                cols, vals = ctx.matrix(lv)
                n = cols.shape[0]
                vals = vals.reshape(n, 125, 3, 3)
                fin = np.isfinite(vals).all()
                diag = np.zeros((n, 3, 3))
                for k in range(125):
                    hit = (cols[:, k] == np.arange(n)) & (np.abs(vals[:, k]).sum(axis=(1, 2)) > 0)
                    diag[hit] = vals[hit, k].astype(np.float64)
                det = np.linalg.det(diag)
                cond = np.linalg.cond(diag)
                print("  level", lv, "n", n, "finite", fin, "min det", det.min(), "max cond", cond.max(), "neg det", (det <= 0).sum())
            x = np.random.default_rng(0).standard_normal((ctx.Nn, 3)).astype(np.float32)
            y = ctx.vcycle(x)
            print("  vcycle(random) finite:", np.isfinite(y).all())
            r = ctx.residual()
            print("  residual finite:", np.isfinite(r).all(), "vcycle(residual) finite:", np.isfinite(ctx.vcycle(r)).all())
        except Exception as e2:
            print("  diag failed:", e2)
    del ctx
print("fails", fails, "of", reps)
