"""P2G / G2P / force-pass time (HIP events from the context profile) at C2 / C3 size; HOT_LIB selects another build of the library
(e.g. the per-phase clock build of tools/hess_phases.sh).  HOT_COLD=1 overwrites 2 GB of device memory before every call (the 256 MB
last-level cache then holds none of the particle arrays, as in a real step where the solve ran in between); HOT_P2G_ONLY=1 times hot_p2g alone; HOT_PRESTEPS=n advances n
time steps first (the bench measures the transfers inside steps 2..: 0.153 + 0.029 / 0.165 ms at C2 where the untouched lattice gives 0.118 +
0.029 / 0.150)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import hot_amd
from hot_amd import parallel, synth

cold = bool(os.environ.get("HOT_COLD"))
if cold:
    import torch
    junk = torch.empty(1 << 29, dtype=torch.float32, device="cuda")


def flush():
    if cold:
        junk.add_(1.0)
        torch.cuda.synchronize()


for which in (sys.argv[1:] or ["C2", "C3"]):
    cfg = dict(synth.CONFIGS[which])
    cloud = parallel.shard_cloud(cfg, 0, 1, n=cfg["n"])
    lib = hot_amd.HotLib(os.environ["HOT_LIB"]) if os.environ.get("HOT_LIB") else hot_amd.load()
    ctx = lib.context(dtype=1 if cfg["dtype"] == np.float64 else 0, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=3, profile=1)
    ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
    for _ in range(int(os.environ.get("HOT_PRESTEPS", "0"))):  # a moved body: cell populations no longer the lattice's 8 per cell
        ctx.advance(cfg["dt"])
    if os.environ.get("HOT_P2G_ONLY"):  # experiment builds whose P2G leaves no grid behind (tools/variant.sh ... -DHOT_P2G_NO_ITEMS=1)
        ctx.sort(), ctx.p2g()
        ctx.profile_reset()
        for _ in range(6):
            flush()
            ctx.p2g()
        print(which, {k: round(v["total_ms"] / v["calls"], 4) for k, v in ctx.profile().items() if "p2g" in k})
        continue
    ctx.sort(), ctx.p2g(), ctx.begin_step(cfg["dt"])
    ctx.profile_reset()
    for _ in range(6):
        flush()
        ctx.p2g()
    ctx.begin_step(cfg["dt"])
    dv = ctx.get_dv()
    for _ in range(6):
        flush()
        ctx.update_state(dv)
        flush()
        ctx.residual()
    for _ in range(6):
        flush()
        ctx.g2p(0.0)
    t = ctx.profile()
    print(which, {k: round(v["total_ms"] / v["calls"], 4) for k, v in t.items() if any(x in k for x in ("p2g", "g2p", "force", "state", "reduce"))})
    del ctx
