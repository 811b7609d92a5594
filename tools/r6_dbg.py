"""round 6 debugging: determinism of the GS launch structures after a hierarchy rebuild; P2G stream kernel against the rounds 2 - 5 kernel"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from hot_amd import synth
from tests.test_gpu_fullsize import make
ablib = hot_amd.HotLib(hot_amd.AB_LIB_PATH)
for cname, n in (("C2", int(os.environ.get("N", "63"))), ("C3", 40)):
    cfg = synth.CONFIGS[cname]
    ctx, cloud = make(ablib, cfg, n)
    ctx.sort()
    os.environ["HOT_P2G_CELLS2"] = "1"
    ctx.p2g()
    g0 = {k: np.array(v, copy=True) for k, v in ctx.grid().items()}
    os.environ.pop("HOT_P2G_CELLS2")
    ctx.p2g()
    g1 = ctx.grid()
    for k in g0:
        if g0[k].dtype.kind == "f":
            print(cname, "p2g stream vs cells2", k, "max rel", float(np.abs(g1[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-300)))
        else:
            print(cname, "p2g stream vs cells2", k, "equal", np.array_equal(g0[k], g1[k]))
    for sw in ("HOT_P2G_CELLS2", ""):
        if sw:
            os.environ[sw] = "1"
        ctx.profile_reset() if hasattr(ctx, "profile_reset") else None
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.p2g()
        ctx.sync()
        print(cname, "p2g x20 wall ms each", sw or "stream", (time.perf_counter() - t0) * 1e3 / 20)
        os.environ.pop(sw, None)
    del ctx
cfg = synth.CONFIGS["C2"]
ctx, cloud = make(ablib, cfg, 63)
ctx.sort(), ctx.p2g(), ctx.begin_step(cfg["dt"])
ctx.update_state(ctx.get_dv())
ctx.build_hessian(), ctx.build_mg()
x = ctx.project(np.random.default_rng(1).standard_normal((ctx.Nn, 3)))
f = [ctx.vcycle(x) for _ in range(3)]
print("fused deterministic", [np.array_equal(f[0], y) for y in f[1:]])
ctx.build_mg()
f2 = [ctx.vcycle(x) for _ in range(3)]
print("fused after rebuild: deterministic", [np.array_equal(f2[0], y) for y in f2[1:]], "equal to before", np.array_equal(f[0], f2[0]), np.abs(f[0] - f2[0]).max() / np.abs(f[0]).max())
os.environ["HOT_GS_PAIR"] = "1"
ctx.build_mg()
p = [ctx.vcycle(x) for _ in range(4)]
print("pair after rebuild: deterministic", [np.array_equal(p[0], y) for y in p[1:]], [float(np.abs(p[0] - y).max() / np.abs(p[0]).max()) for y in p[1:]], "vs fused", np.abs(f[0] - p[0]).max() / np.abs(f[0]).max())
