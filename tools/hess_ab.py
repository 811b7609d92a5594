"""Hessian assembly at C2 / C3 size: k_hessian_rows (production) against the LDS-staged tile kernel of rounds 2 - 4 (A/B build, HOT_HESSIAN_TILES)
and its MFMA pair phase (HOT_HESSIAN_MFMA); times from HIP events, matrices compared through SpMV with a random vector."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from hot_amd import parallel, synth

which = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = dict(synth.CONFIGS[which])
cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
lib = hot_amd.HotLib(hot_amd.AB_LIB_PATH)
ctx = lib.context(dtype=1 if cfg["dtype"] == np.float64 else 0, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=1, profile=1)
ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
ctx.sort(), ctx.p2g(), ctx.begin_step(cfg["dt"])
ctx.update_state(ctx.get_dv())
x = np.random.default_rng(1).standard_normal((ctx.Nn, 3))
out = {}
for label, env in (("mfma", "HOT_HESSIAN_MFMA"), ("tiles", "HOT_HESSIAN_TILES"), ("rows", None)):
    os.environ.pop("HOT_HESSIAN_MFMA", None), os.environ.pop("HOT_HESSIAN_TILES", None)
    if env:
        os.environ[env] = "1"
    ctx.build_hessian()
    ctx.profile_reset()
    for _ in range(3):
        ctx.build_hessian()
    t = ctx.profile()
    out[label] = ctx.spmv(0, x).astype(np.float64)
    print(which, label, {k: round(v["total_ms"] / v["calls"], 3) for k, v in t.items() if k.startswith("hessian")})
for k in ("mfma", "tiles"):
    print("rel diff of A x, %s vs rows:" % k, np.abs(out[k] - out["rows"]).max() / np.abs(out["rows"]).max())
