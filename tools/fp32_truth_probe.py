"""fp32 accuracy probe: HIP fp32 and the oracle's fp32 (reference arithmetic) against the oracle's fp64 on the SAME float inputs.
Nodes are matched by grid coordinate (the fp32 / fp64 SPGrid blocks differ, so the node ids do)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd
from hot_amd import synth, parallel
from tests.oracle_lib import load_oracle

lib, ora = hot_amd.load(), load_oracle()


def key(c):
    c = c.astype(np.int64)
    return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]


def run(L, dtype, cloud, cfg, its, dt, what):
    T = np.float64 if dtype == 1 else np.float32
    kw = dict(dtype=dtype, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=cfg["levelCnt"], max_iterations=its, debug_store=1)
    kw.update(cfg.get("kw", {}))
    ctx = L.context(**kw)
    ctx.set_particles(*(cloud[k].astype(T) for k in ("X", "V", "mass", "vol", "mu", "lam")))
    o, nrm = synth.sticky_floor(cloud["corner"][1], cloud["dx"])
    ctx.set_sticky_halfspaces(o, nrm)
    ctx.sort(), ctx.p2g(), ctx.begin_step(dt)
    g = ctx.grid()
    out = dict(k=key(g["id2coord"]), mass=g["mass"].astype(np.float64), v=g["v"].astype(np.float64))
    if what == "pieces":
        dv0 = ctx.get_dv()
        rng = np.random.default_rng(7)
        # the perturbation is a function of the node coordinate, so that all three runs see the same field
        h = (out["k"] * 2654435761 % 1000003) / 1000003.0
        dv = dv0 + (0.02 * (np.stack([h, (h * 7) % 1, (h * 13) % 1], 1) - 0.5)).astype(dv0.dtype)
        out["e"] = np.array([ctx.update_state(dv)])
        out["r"] = ctx.residual().astype(np.float64)
        ps = ctx.particle_state()
        for f in ("F", "stress", "gradV"):
            out["p_" + f] = ps[f].astype(np.float64)
    else:
        st = ctx.solve()
        out["dv"] = ctx.get_dv().astype(np.float64)
        out["st"] = st
    return out


def match(a, b, f):
    ka, kb = a["k"], b["k"]
    common, ia, ib = np.intersect1d(ka, kb, return_indices=True)
    x, y = a[f][ia], b[f][ib]
    return np.abs(x - y).max() / max(np.abs(y).max(), 1e-300), len(common), len(ka), len(kb)


for cname, n, dt in (("C3", 24, 1 / 24), ("C3", 40, 1 / 24), ("C5", 32, 1 / 24)):
    cfg = dict(synth.CONFIGS[cname])
    cloud = parallel.shard_cloud(cfg, 0, 1, n=n)
    cloud = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v) for k, v in cloud.items()}
    t = run(ora, 1, cloud, cfg, 0, dt, "pieces")
    o = run(ora, 0, cloud, cfg, 0, dt, "pieces")
    h = run(lib, 0, cloud, cfg, 0, dt, "pieces")
    print(cname, n, "pieces  oracle32-vs-64 | hip32-vs-64 | hip32-vs-oracle32:", " ".join("%s %.2e|%.2e|%.2e" % (f, match(o, t, f)[0], match(h, t, f)[0], match(h, o, f)[0]) for f in ("mass", "v", "r")),
          "energy %.3e|%.3e" % (abs(o["e"][0] - t["e"][0]) / abs(t["e"][0]), abs(h["e"][0] - t["e"][0]) / abs(t["e"][0])), "nodes", match(h, t, "mass")[1:], flush=True)
    prel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    print(cname, n, "particles oracle32-vs-64 | hip32-vs-64:", " ".join("%s %.2e|%.2e" % (f, prel(o["p_" + f], t["p_" + f]), prel(h["p_" + f], t["p_" + f])) for f in ("F", "gradV", "stress")), flush=True)
    for its in (1, 3, 6):
        t = run(ora, 1, cloud, cfg, its, dt, "solve")
        o = run(ora, 0, cloud, cfg, its, dt, "solve")
        h = run(lib, 0, cloud, cfg, its, dt, "solve")
        cnt = ("iterations", "linesearch_trials", "linear_iterations", "vcycles")
        print(cname, n, "its", its, "dv: oracle32-vs-64 %.2e | hip32-vs-64 %.2e | hip32-vs-oracle32 %.2e" % (match(o, t, "dv")[0], match(h, t, "dv")[0], match(h, o, "dv")[0]),
              "counters 64/o32/h32", [t["st"][k] for k in cnt], [o["st"][k] for k in cnt], [h["st"][k] for k in cnt], flush=True)
