"""Parity fuzzing aid (not a test): random bodies (carved cubes, random particles per cell, corners anywhere in the 4096^3 grid,
random materials per particle), random solver knobs, a few nonlinear iterations on the HIP library and on the CPU oracle; reports every
case whose counters or dv differ.  python tools/fuzz_parity.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def main():
    import hot_amd
    from hot_amd import synth
    from tests.oracle_lib import load_oracle
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mode = sys.argv[3] if len(sys.argv) > 3 else ""  # "minres_fixed": lsolver 1 behind a fixed (linear) preconditioner; "fp32_one": fp32, one iteration
    rng = np.random.default_rng(seed)
    hip, cpu = hot_amd.load(), load_oracle()
    bad = 0
    for case in range(cases):
        n = int(rng.integers(4, 13))
        ppc = int(rng.choice([1, 2, 4, 8, 12]))
        corner = tuple(float(v) for v in rng.choice([0.05, 1.0, 5.0, 17.3, 38.0], 3))  # 38.0 / 0.01 = node 3800 of 4096
        dtype = int(rng.random() < 0.8)
        T = np.float64 if dtype else np.float32
        c = synth.cube_cloud(n, ppc=ppc, corner=corner, dtype=T, seed=int(rng.integers(1, 10 ** 6)), cells=tuple(int(v) for v in rng.integers(3, n + 1, 3)))
        X = c["X"].astype(np.float64)
        ctr = X.mean(0)
        shape = int(rng.integers(0, 4))
        r = np.linalg.norm(X - ctr, axis=1)
        ext = X.max(0) - X.min(0)
        if shape == 1:
            keep = r < 0.5 * ext.min() + 0.004
        elif shape == 2:
            keep = (r > 0.25 * ext.min()) | (np.abs(X[:, 0] - ctr[0]) < 0.011)
        elif shape == 3:
            keep = rng.random(len(X)) < 0.6  # random holes: cells with few or no particles
        else:
            keep = np.ones(len(X), bool)
        if keep.sum() < 8:
            keep[:] = True
        mu = (c["mu"].astype(np.float64) * rng.uniform(0.5, 2.0, len(X))).astype(T)
        lam = (c["lam"].astype(np.float64) * rng.uniform(0.5, 2.0, len(X))).astype(T)
        lsolver = int(rng.choice([3, 3, 3, 2, 1]))
        kw = dict(lsolver=lsolver, levelCnt=int(rng.integers(1, 4)), max_iterations=int(rng.integers(1, 5)), cneps=1e-7, linesearch=int(rng.random() < 0.8),
                  boundaryType=int(rng.integers(0, 2)), useCN=int(rng.random() < 0.8), Ainv=int(rng.choice([0, 1])))
        if rng.random() < 0.3:
            kw.update(smoother=int(rng.choice([0, 1, 5])), coarseSolver=int(rng.choice([2, 5, 1])))
        if rng.random() < 0.2:
            kw.update(times=2)
        dt = float(rng.choice([1 / 24, 0.02, 0.01]))
        if mode == "minres_fixed":
            dtype, T = 1, np.float64
            kw.update(lsolver=1, levelCnt=int(rng.integers(1, 3)), smoother=0, coarseSolver=0)
        if mode == "fp32_one":
            dtype, T = 0, np.float32
            kw.update(max_iterations=1)
        c = {k: (v.astype(T) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
        mu, lam = mu.astype(T), lam.astype(T)
        out = {}
        err = None
        for name, lib in (("gpu", hip), ("cpu", cpu)):
            try:
                over = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("HOT_FUZZ_CFG", "").split(",") if kv)} if name == "gpu" else {}  # e.g. HOT_FUZZ_CFG=ls_energy_only=1: HIP-only knobs
                ctx = lib.context(dtype=dtype, dx=c["dx"], gravity=(0, -9.8, 0), **dict(kw, **over))
                ctx.set_particles(c["X"][keep], c["V"][keep], c["mass"][keep], c["vol"][keep], mu[keep], lam[keep])
                o, nrm = synth.sticky_floor(float(X[keep][:, 1].min()) - 0.002, c["dx"])
                ctx.set_sticky_halfspaces(o, nrm)
                ctx.sort(), ctx.p2g(), ctx.begin_step(dt)
                st = ctx.solve()
                ctx.g2p(dt)
                out[name] = (ctx.get_dv().astype(np.float64), st, ctx.get_particles()["X"].astype(np.float64))
            except Exception as e:  # both sides must refuse the same inputs
                out[name] = ("error", str(e)[:120], None)
        g, cc = out["gpu"], out["cpu"]
        tag = "case %d: n=%d ppc=%d corner=%s dtype=%d shape=%d Np=%d dt=%.4g %s" % (case, n, ppc, corner, dtype, shape, int(keep.sum()), dt, kw)
        if isinstance(g[0], str) or isinstance(cc[0], str):
            if isinstance(g[0], str) != isinstance(cc[0], str):
                bad += 1
                print("MISMATCH (one side refused)", tag, g[1] if isinstance(g[0], str) else "", cc[1] if isinstance(cc[0], str) else "", flush=True)
            else:
                print("both refused:", tag, "|", g[1], flush=True)
            continue
        tol = 1e-8 if dtype else 5e-3
        keys = ("iterations", "linesearch_trials", "vcycles", "dropped_pairs", "num_nodes", "num_levels") + (("linear_iterations",) if dtype else ())
        same = all(g[1][k] == cc[1][k] for k in keys)
        e_dv, e_x = rel(g[0], cc[0]), np.abs(g[2] - cc[2]).max() / c["dx"]
        # MINRES behind a weak preconditioner (no coarse level) reproduces only to kappa^2 eps (Sleijpen, van der Vorst, Modersitzki 2000):
        # two runs of the SAME library differ by 1e-7 there (LDS-atomic order in the assembly, 4e-16 in the matrix); informational only
        soft = kw["lsolver"] == 1 and kw["levelCnt"] == 1
        # a linear solve that ran into its iteration cap on the oracle (10000, the reference's max_iterations) has not converged on either side: what it
        # returns after 10^4 fp32 steps is round-off; informational as well
        capped = kw["lsolver"] in (1, 2) and cc[1]["linear_iterations"] >= 10000 * max(cc[1]["iterations"], 1)
        soft = soft or capped
        ok = (e_dv < tol and e_x < tol and (same or not dtype)) or (soft and e_dv < 1e-1)
        if abs(g[1]["energy"] - cc[1]["energy"]) > (1e-9 if dtype else 2e-2) * max(abs(cc[1]["energy"]), 1e-6) and not soft and np.isfinite(cc[1]["energy"]):
            ok = False
        note = ""
        if capped:
            ok = True
        elif not ok and not dtype and np.isfinite(g[2]).all() and np.isfinite(cc[2]).all():
            # an fp32 case: who is closer to the oracle's fp64 run on the same float inputs?  (fp32 line searches decide on energy differences below fp32's
            # resolution; the device evaluates trial energies in the invariant form, the oracle in the reference's)
            try:
                c64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
                ctx = cpu.context(dtype=1, dx=c["dx"], gravity=(0, -9.8, 0), **kw)
                ctx.set_particles(c64["X"][keep], c64["V"][keep], c64["mass"][keep], c64["vol"][keep], mu[keep].astype(np.float64), lam[keep].astype(np.float64))
                o, nrm = synth.sticky_floor(float(X[keep][:, 1].min()) - 0.002, c["dx"])
                ctx.set_sticky_halfspaces(o, nrm)
                ctx.sort(), ctx.p2g(), ctx.begin_step(dt)
                ctx.solve()
                ctx.g2p(dt)
                x64 = ctx.get_particles()["X"].astype(np.float64)
                d_g, d_c = np.abs(g[2] - x64).max() / c["dx"], np.abs(cc[2] - x64).max() / c["dx"]
                note = " [fp32: |x - x_fp64| / dx  HIP %.2e  oracle-fp32 %.2e]" % (d_g, d_c)
                if d_g <= max(2 * d_c, 1e-4):
                    ok, note = True, note + " fp32 round-off, informational"
            except Exception as e:
                note = " [fp64 run failed: %s]" % str(e)[:80]
        if not ok:
            bad += 1
        extra = "lin %d/%d its %d/%d trials %d/%d nan gpu=%s cpu=%s E %.6g/%.6g" % (g[1]["linear_iterations"], cc[1]["linear_iterations"], g[1]["iterations"], cc[1]["iterations"],
                                                                                    g[1]["linesearch_trials"], cc[1]["linesearch_trials"], not np.isfinite(g[0]).all(), not np.isfinite(cc[0]).all(), g[1]["energy"], cc[1]["energy"])
        print("%s dv %.2e x %.2e counters %s %s | %s" % (("ok (linear solves capped, informational)" if capped and ok else "ok      ") if ok else "MISMATCH", e_dv, e_x, "equal" if same else "differ", extra, tag + note), flush=True)
    print("cases", cases, "mismatches", bad)


if __name__ == "__main__":
    main()
