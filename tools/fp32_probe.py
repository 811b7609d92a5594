"""fp32 parity probe: HIP fp32 against the oracle's narrow (float sums, the reference's) and wide (double sums) variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hot_amd
from hot_amd import synth
from tests import pipeline_checks as pc
from tests.oracle_lib import load_oracle, wide_sums

lib, ora = hot_amd.load(), load_oracle()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def pieces(L, n, E):
    ctx, c = pc.make_ctx(L, n=n, dtype=0, bc=True, E=E)
    pc.prepare(ctx)
    g = ctx.grid()
    dv0 = ctx.get_dv()
    rng = np.random.default_rng(7)
    dv = dv0 + 0.02 * rng.standard_normal(dv0.shape).astype(dv0.dtype)
    e = ctx.update_state(dv)
    st = ctx.particle_state()
    r = ctx.residual()
    tol_n = ctx.cn_tolerance()
    x = rng.standard_normal(dv0.shape)
    hx = ctx.matfree_multiply(x)
    ctx.build_hessian()
    ax = ctx.spmv(0, x)
    return dict(mass=g["mass"], v=g["v"], dv0=dv0, e=np.array([e]), r=r, tol=tol_n, hx=hx, ax=ax, stress=st["stress"], gradV=st["gradV"])


for n, E in ((8, 5e4), (12, 1e9)):
    g = pieces(lib, n, E)
    for wide in (False, True):
        with wide_sums(wide):
            c = pieces(ora, n, E)
        print("pieces n=%d E=%g %s:" % (n, E, "wide" if wide else "narrow"), " ".join("%s=%.2e" % (k, rel(g[k], c[k])) for k in g), flush=True)


def iters(cname, n, its, dt, levelCnt=None):
    cfg = synth.CONFIGS[cname]
    from hot_amd import parallel
    cloud = parallel.shard_cloud(cfg, 0, 1, n=n)
    out = {}
    for name in ("gpu", "narrow", "wide"):
        L = lib if name == "gpu" else ora
        with wide_sums(name == "wide"):
            ctx = L.context(dtype=0, dx=cloud["dx"], gravity=(0, -9.8, 0), levelCnt=levelCnt or cfg["levelCnt"], max_iterations=its)
            ctx.set_particles(cloud["X"], cloud["V"], cloud["mass"], cloud["vol"], cloud["mu"], cloud["lam"])
            o, nrm = synth.sticky_floor(cloud["corner"][1], cloud["dx"])
            ctx.set_sticky_halfspaces(o, nrm)
            ctx.sort(), ctx.p2g(), ctx.begin_step(dt)
            st = ctx.solve()
            out[name] = (ctx.get_dv().astype(np.float64), st)
            del ctx
    for name in ("narrow", "wide"):
        same = all(out["gpu"][1][k] == out[name][1][k] for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles"))
        print("%s n=%d its=%d vs %s: rel dv %.3e energies %.9g %.9g counters %s" % (cname, n, its, name, rel(out["gpu"][0], out[name][0]), out["gpu"][1]["energy"], out[name][1]["energy"], "equal" if same else "differ"), flush=True)


nC3 = int(os.environ.get("PROBE_C3", "40"))
for its in (1, 2, 3, 5):
    iters("C3", nC3, its, 1.0 / 24)
for its in (1, 3, 5):
    iters("C5", 32, its, 1.0 / 24)
