"""Comparison of a library exporting the C ABI of include/hot_mi355x.h with the numpy-generated vectors of
tests/golden/fp_golden.npz (generator: tests/golden/make_fp_golden.py, numpy only).  The library is driven through a
small ctypes mirror written here, independent of hot_amd/binding.py, so a mistake in the shared binding cannot cancel."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "fp_golden.npz")


class Config(C.Structure):  # include/hot_mi355x.h: hot_config
    _fields_ = [("dtype", C.c_int32), ("device", C.c_int32), ("dx", C.c_double), ("gravity", C.c_double * 3), ("apic_rpic_ratio", C.c_double), ("cfl", C.c_double),
                ("lsolver", C.c_int32), ("Ainv", C.c_int32), ("smoother", C.c_int32), ("coarseSolver", C.c_int32), ("levelCnt", C.c_int32), ("times", C.c_int32),
                ("levelscale", C.c_int32), ("omega", C.c_double), ("topomega", C.c_double), ("cneps", C.c_double), ("useCN", C.c_int32), ("project", C.c_int32),
                ("systemBCProject", C.c_int32), ("linesearch", C.c_int32), ("matrixFree", C.c_int32), ("boundaryType", C.c_int32), ("useAdaptiveHessian", C.c_int32),
                ("topDownMGS", C.c_int32), ("max_iterations", C.c_int32), ("plasticity", C.c_int32), ("yield_stress", C.c_double), ("snow", C.c_double * 5),
                ("profile", C.c_int32), ("debug_store", C.c_int32), ("useBaselineMultigrid", C.c_int32), ("gs_chain", C.c_int32), ("gs_sub_block", C.c_int32), ("shard_gs", C.c_int32), ("shard_replicated", C.c_int32), ("ls_energy_only", C.c_int32), ("linear_iteration_cap", C.c_int32), ("shard_owner", C.c_int32), ("reserved", C.c_int32 * 5)]


class Raw:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.p = prefix

    def f(self, name, restype=C.c_int):
        fn = getattr(self.lib, self.p + name)
        fn.restype = restype
        return fn

    def create(self, dtype, **kw):
        cfg = Config()
        self.f("default_config", None)(C.byref(cfg))
        cfg.dtype = dtype
        for k, v in kw.items():
            if k == "snow":
                for i in range(5):
                    cfg.snow[i] = float(v[i])
            else:
                setattr(cfg, k, v)
        h = C.c_void_p()
        rc = self.f("create")(C.byref(cfg), C.byref(h))
        assert rc == 0 and h, rc
        return h

    def call(self, name, h, *args):
        rc = self.f(name)(h, *args)
        if rc != 0:
            msg = self.f("last_error", C.c_char_p)(h)
            raise RuntimeError(f"{name} -> {rc}: {msg}")


def vp(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def relerr(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-300)


def check_constitutive(path, prefix, dtype):
    g = np.load(GOLDEN)
    T = np.float64 if dtype == 1 else np.float32
    r = Raw(path, prefix)
    h = r.create(dtype)
    n = g["F"].shape[0]
    F = np.ascontiguousarray(g["F"], T)
    mu, lam = np.full(n, float(g["mu"]), T), np.full(n, float(g["lam"]), T)
    out = {}
    for project in (0, 1):
        psi, P, D = np.empty(n, T), np.empty((n, 9), T), np.empty((n, 81), T)
        r.call("constitutive_eval", h, C.c_int32(n), vp(F), vp(mu), vp(lam), C.c_int32(project), vp(psi), vp(P), vp(D))
        out[project] = (psi, P, D)
    r.f("destroy", None)(h)
    # scale: stresses / moduli are O(mu |F - R|) / O(mu + lambda J^2 ...): compare per sample against the sample's own magnitude
    tol = 1e-10 if dtype == 1 else 3e-4
    psi, P, D = out[0]
    scaleP = np.abs(g["P"]).max(1) + float(g["mu"]) * 1e-3
    assert (np.abs(P.astype(np.float64) - g["P"]).max(1) / scaleP).max() < tol
    assert (np.abs(psi.astype(np.float64) - g["psi"]) / (np.abs(g["psi"]) + float(g["mu"]) * 1e-3)).max() < tol
    gen = slice(9, None)  # members 0..8 sit at kinks of the singular-value frame (repeated / vanishing / clamped singular values)
    scaleD = np.abs(g["dPdF"]).max(1)
    errD = np.abs(D.astype(np.float64) - g["dPdF"]).max(1) / scaleD
    assert errD[gen].max() < tol * 10, errD[gen].max()
    errDp = np.abs(out[1][2].astype(np.float64) - g["dPdF_projected"]).max(1) / scaleD
    assert errDp[gen].max() < tol * 10, errDp[gen].max()
    # the special members: stress and energy are continuous there, and are pinned above; of the derivative the identity
    # and pure-rotation members have a well-defined limit (A, B blocks of a triple singular value)
    assert errD[:2].max() < tol * 10 and errDp[:2].max() < tol * 10
    return dict(P=(np.abs(P.astype(np.float64) - g["P"]).max(1) / scaleP).max(), D=errD[gen].max(), Dproj=errDp[gen].max())


def check_plasticity(path, prefix, dtype):
    g = np.load(GOLDEN)
    T = np.float64 if dtype == 1 else np.float32
    r = Raw(path, prefix)
    n = g["pl_F"].shape[0]
    tol = 1e-11 if dtype == 1 else 2e-5
    h = r.create(dtype, plasticity=1, yield_stress=float(g["vm_yield"]))
    F = np.ascontiguousarray(g["pl_F"], T)
    mu, lam, Jp = np.full(n, float(g["mu"]), T), np.full(n, float(g["lam"]), T), np.ones(n, T)
    r.call("plasticity_eval", h, C.c_int32(1), C.c_int32(n), vp(F), vp(mu), vp(lam), vp(Jp))
    r.f("destroy", None)(h)
    assert relerr(F, g["vm_F"]) < tol, relerr(F, g["vm_F"])
    stay = ~g["vm_hit"]
    assert np.array_equal(F[stay], np.ascontiguousarray(g["pl_F"], T)[stay])  # inside the yield surface nothing moves
    h = r.create(dtype, plasticity=2, snow=tuple(g["snow_params"]))
    F = np.ascontiguousarray(g["pl_F"], T)
    mu, lam, Jp = np.full(n, float(g["mu"]), T), np.full(n, float(g["lam"]), T), np.ascontiguousarray(g["snow_Jp0"], T)
    r.call("plasticity_eval", h, C.c_int32(2), C.c_int32(n), vp(F), vp(mu), vp(lam), vp(Jp))
    r.f("destroy", None)(h)
    assert relerr(F, g["snow_F"]) < tol and relerr(Jp, g["snow_Jp"]) < tol * 10
    assert relerr(mu, g["snow_mu"]) < tol * 50 and relerr(lam, g["snow_lam"]) < tol * 50  # exp(psi (Jp - Jp')) amplifies the error of Jp by psi


def check_p2g(path, prefix, dtype):
    g = np.load(GOLDEN)
    T = np.float64 if dtype == 1 else np.float32
    r = Raw(path, prefix)
    h = r.create(dtype, dx=float(g["p2g_dx"]))
    n = g["p2g_X"].shape[0]
    X, V, Cm, m = (np.ascontiguousarray(g[k], T) for k in ("p2g_X", "p2g_V", "p2g_C", "p2g_mass"))
    vol, mu, lam = np.full(n, 1e-6 / 8, T), np.full(n, float(g["mu"]), T), np.full(n, float(g["lam"]), T)
    r.call("set_particles", h, C.c_int64(n), vp(X), vp(V), vp(m), vp(Cm), None, vp(vol), vp(mu), vp(lam), None)
    r.call("sort", h)
    r.call("p2g", h)
    np_, ng, nb, nn = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
    r.call("get_counts", h, C.byref(np_), C.byref(ng), C.byref(nb), C.byref(nn))
    ic, gm, gv = np.empty((nn.value, 3), np.int32), np.empty(nn.value, T), np.empty((nn.value, 3), T)
    r.call("get_grid", h, vp(ic), vp(gm), vp(gv))
    r.f("destroy", None)(h)
    # fp32: positions near 5.0 with dx = 0.01 leave ~1e-5 of the cell in the weights, and a weight that is tiny in double can
    # round to exactly 0 in float (such a node is then not a DOF): compare on the coordinates both sides have
    key = lambda a: [tuple(x) for x in a]
    have = {k: i for i, k in enumerate(key(ic))}
    ref_idx = [i for i, k in enumerate(key(g["p2g_nodes"])) if k in have]
    got_idx = [have[k] for i, k in enumerate(key(g["p2g_nodes"])) if k in have]
    if dtype == 1:
        assert nn.value == g["p2g_nodes"].shape[0] and len(ref_idx) == nn.value
    else:
        assert len(ref_idx) >= 0.98 * g["p2g_nodes"].shape[0] and nn.value <= g["p2g_nodes"].shape[0]
    tol = 1e-11 if dtype == 1 else 5e-3
    rm, rv = g["p2g_node_mass"][ref_idx], g["p2g_node_v"][ref_idx]
    assert np.abs(gm[got_idx] - rm).max() < tol * rm.max()
    heavy = rm > 1e-3 * rm.max()  # v = mv / m of almost massless nodes is ill-conditioned in any precision
    assert np.abs(gv[got_idx] - rv)[heavy].max() < tol * np.abs(rv[heavy]).max()
    assert np.abs((gm[got_idx, None] * gv[got_idx]) - (rm[:, None] * rv)).max() < tol * np.abs(rm[:, None] * rv).max()


def check_step(path, prefix, dtype=1):
    """One whole tiny time step against tests/golden/np_step.py (numpy only): node masses and velocities, energy, residual, the assembled
    Hessian with its boundary projection, prolongation and Galerkin coarse matrix, one symmetric coloured GS sweep, the two-level V-cycle
    and two L-BFGS iterations with their line searches — in the library's own node numbering (an input of the numpy restatement)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import np_step as ns
    T = np.float64 if dtype == 1 else np.float32
    c = ns.tiny_cloud()
    r = Raw(path, prefix)
    kw = dict(dx=c.dx, levelCnt=2, lsolver=3, smoother=5, coarseSolver=2, max_iterations=2)
    h = r.create(dtype, **kw)
    n = c.X.shape[0]
    cmaj = lambda M: np.ascontiguousarray(np.transpose(M, (0, 2, 1)).reshape(-1, 9), T)
    X, V, m, vol = (np.ascontiguousarray(a, T) for a in (c.X, c.V, c.mass, c.vol))
    mu, lam = np.full(n, c.mu, T), np.full(n, c.lam, T)
    Cm, Fm = cmaj(c.C), cmaj(c.F)
    r.call("set_particles", h, C.c_int64(n), vp(X), vp(V), vp(m), vp(Cm), vp(Fm), vp(vol), vp(mu), vp(lam), None)
    org, nrm = np.array([0.0, c.floor_y, 0.0]), np.array([0.0, 1.0, 0.0])
    r.call("set_sticky_halfspaces", h, C.c_int32(1), vp(org), vp(nrm))
    r.call("sort", h)
    r.call("p2g", h)
    np_, ng, nb, nn = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
    r.call("get_counts", h, C.byref(np_), C.byref(ng), C.byref(nb), C.byref(nn))
    N = nn.value
    ic, gm, gv = np.empty((N, 3), np.int32), np.empty(N, T), np.empty((N, 3), T)
    r.call("get_grid", h, vp(ic), vp(gm), vp(gv))
    st = ns.Step(c, ic)
    out = {}
    out["mass"], out["v"] = relerr(gm, st.m), relerr(gv, st.vn)
    r.call("begin_step", h, C.c_double(c.dt))
    dv = np.empty((N, 3), T)
    r.call("get_dv", h, vp(dv))
    out["dv0"] = relerr(dv, st.dv0)
    e = C.c_double()
    r.call("update_state", h, None, C.byref(e))
    out["energy"] = abs(e.value - st.energy(st.dv0)) / abs(st.energy(st.dv0))
    res = np.empty((N, 3), T)
    r.call("residual", h, vp(res))
    out["residual"] = relerr(res, st.residual(st.dv0))
    r.call("build_hessian", h)
    r.call("build_mg", h)

    def dense(level, nrows):
        nr, cs = C.c_int32(), C.c_int32()
        r.call("get_level", h, C.c_int32(level), C.byref(nr), C.byref(cs), None)
        assert nr.value == nrows
        col, val = np.empty((nrows, cs.value), np.int32), np.empty((nrows, cs.value, 9), T)
        r.call("get_matrix", h, C.c_int32(level), vp(col), vp(val))
        A = np.zeros((3 * nrows, 3 * nrows))
        for i in range(nrows):
            for k in range(cs.value):
                A[3 * i:3 * i + 3, 3 * col[i, k]:3 * col[i, k] + 3] += val[i, k].reshape(3, 3).T  # 3x3 column-major
        return A
    H = st.hessian(st.dv0)
    A0 = dense(0, N)
    out["hessian"] = np.abs(A0 - H).max() / np.abs(H).max()
    coord1, P = ns.coarsen(st.coord)
    nr1 = C.c_int32()
    r.call("get_level", h, C.c_int32(1), C.byref(nr1), C.byref(C.c_int32()), None)
    assert nr1.value == len(coord1), (nr1.value, len(coord1))
    ic1 = np.empty((nr1.value, 3), np.int32)
    r.call("get_level", h, C.c_int32(1), C.byref(nr1), C.byref(C.c_int32()), vp(ic1))
    assert [tuple(int(v) for v in q) for q in ic1] == coord1  # first-touch coarse numbering, bit for bit
    pc, pw = np.empty((N, 8), np.int32), np.empty((N, 8), T)
    r.call("get_prolongation", h, C.c_int32(0), vp(pc), vp(pw))
    Pd = np.zeros_like(P)
    for i in range(N):
        for k in range(8):
            Pd[i, pc[i, k]] += pw[i, k]
    out["prolongation"] = np.abs(Pd - P).max()
    P3 = ns.expand3(P)
    A1 = P3.T @ H @ P3
    dense1 = dense(1, len(coord1))
    out["coarse_matrix"] = np.abs(dense1 - A1).max() / np.abs(A1).max()
    # one symmetric coloured GS sweep on level 0 (iterations = 2 in the reference's counting)
    b = st.project(np.random.default_rng(7).standard_normal((N, 3)))
    u, rr = np.zeros((N, 3), T), np.array(b, T)  # (a copy: the call overwrites r)
    r.call("smooth", h, C.c_int32(0), C.c_int32(5), C.c_int32(2), C.c_double(0.0), vp(u), vp(rr), None)
    un, rn = ns.gs_smooth(H, st.coord, np.zeros((N, 3)), b, 2)
    out["gs_u"], out["gs_r"] = relerr(u, un), relerr(rr, rn)
    mg = ns.Hierarchy(H, st.coord)
    xin, xout = np.ascontiguousarray(b, T), np.empty((N, 3), T)
    r.call("vcycle", h, vp(xin), vp(xout))
    out["vcycle"] = relerr(xout, mg.vcycle(b))
    # two L-BFGS iterations with their line searches
    stats = (C.c_byte * 512)()
    r.call("solve", h, C.byref(stats))
    r.call("get_dv", h, vp(dv))
    xn, trials, _ = ns.lbfgs(st, 2)
    out["lbfgs_dv"] = relerr(dv, xn)
    out["linesearch_trials"] = (int(np.frombuffer(stats, np.int32, 3)[2]), trials)
    r.f("destroy", None)(h)
    # ---- the stored vectors (tests/golden/step_golden.npz, keyed by coordinates in lexicographic order): the library's numbering mapped onto them
    g = np.load(os.path.join(ROOT, "tests", "golden", "step_golden.npz"))
    lex = {tuple(int(v) for v in k): i for i, k in enumerate(g["coord"])}
    assert len(lex) == N and all(k in lex for k in st.coord)
    perm = np.array([lex[k] for k in st.coord])  # library id -> lexicographic index
    p3 = (3 * perm[:, None] + np.arange(3)[None, :]).reshape(-1)
    out["stored_mass"], out["stored_v"], out["stored_dv0"] = relerr(gm, g["mass"][perm]), relerr(gv, g["v"][perm]), relerr(dv * 0 + st.dv0, g["dv0"][perm])
    out["stored_energy"] = abs(e.value - float(g["energy"])) / abs(float(g["energy"]))
    out["stored_residual"] = relerr(res, g["residual"][perm])
    out["stored_hessian"] = np.abs(A0 - g["hessian"][np.ix_(p3, p3)]).max() / np.abs(g["hessian"]).max()
    lex1 = {tuple(int(v) for v in k): i for i, k in enumerate(g["coord1"])}
    perm1 = np.array([lex1[k] for k in coord1])
    q3 = (3 * perm1[:, None] + np.arange(3)[None, :]).reshape(-1)
    out["stored_prolongation"] = np.abs(Pd - g["P"][np.ix_(perm, perm1)]).max()
    out["stored_coarse_matrix"] = np.abs(dense1 - g["A1"][np.ix_(q3, q3)]).max() / np.abs(g["A1"]).max()
    return out


def check_step_numpy_regression():
    """the numpy restatement in lexicographic numbering reproduces the stored numbering-dependent vectors (pins np_step.py itself)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import np_step as ns
    g = np.load(os.path.join(ROOT, "tests", "golden", "step_golden.npz"))
    st = ns.Step(ns.tiny_cloud(), g["coord"])
    H = st.hessian(st.dv0)
    assert np.abs(H - g["hessian"]).max() < 1e-12 * np.abs(H).max()
    u, rr = ns.gs_smooth(H, st.coord, np.zeros((st.n, 3)), g["lex_rhs"], 2)
    assert relerr(u, g["lex_gs_u"]) < 1e-11 and relerr(rr, g["lex_gs_r"]) < 1e-11
    assert relerr(ns.Hierarchy(H, st.coord).vcycle(g["lex_rhs"]), g["lex_vcycle"]) < 1e-10
    x2, trials, _ = ns.lbfgs(st, 2)
    assert relerr(x2, g["lex_lbfgs_dv"]) < 1e-9 and trials == int(g["lex_linesearch_trials"])


def load_tie(kind):
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return {b: json.load(open(os.path.join(here, "golden", f"spgrid_tie_{kind}_{b}.json"))) for b in ("O2", "native")}


def check_tie_points(lib, kind, dtype):
    """Base nodes at cell faces (tests/golden/spgrid_tie_*.json, oracle/spgrid_ref_driver.cpp run_tie): particles whose X / dx - 0.5 lies within
    an ulp of an integer, where floor(RN(X / dx) - 0.5) — the product rounded first — and floor(RN(X / dx - 0.5)) — one rounding, an fma —
    can differ (they do only where the integer is a power of two: 256, 512, 1024, 2048).  What the vectors establish:
      * the driver built with the reference's own Release flags (-O3 -march=native -fno-math-errno, CMakeLists.txt:28) on an FMA machine evaluates
        the reference's statement shape (product into a temporary, int_floor(x - 0.5) in another function) with ONE rounding: as_compiled == fma;
      * built for the x86-64 baseline (-O2, no FMA instruction) the same source gives the rounded product;
      * the library under test (`lib`: the CPU oracle or the HIP library) takes the fma: its sort keys at these points are the `base_fma` ones,
        i.e. what the reference yields when built the way its CMakeLists builds it."""
    t = load_tie(kind)
    o2, nat = t["O2"], t["native"]
    assert o2["X"] == nat["X"] and o2["base_fma"] == nat["base_fma"] and o2["base_rounded_product"] == nat["base_rounded_product"]
    n = len(o2["X"])
    differ = [i for i in range(n) if o2["base_fma"][i] != o2["base_rounded_product"][i]]
    assert len(differ) >= 8  # real tie points, not only near misses
    assert all(o2["base_fma"][i][0] + 1 == o2["base_rounded_product"][i][0] and o2["base_rounded_product"][i][0] in (256, 512, 1024, 2048) for i in differ)
    assert nat["base_as_compiled"] == nat["base_fma"] and o2["base_as_compiled"] == o2["base_rounded_product"]
    T = np.float32 if dtype == 0 else np.float64
    X, dxs = np.array(o2["X"], T), np.array(o2["dx"], np.float64)
    for dx in np.unique(dxs):
        sel = np.nonzero(dxs == dx)[0]
        m = len(sel)
        one = np.ones(m, T)
        ctx = lib.context(dtype=dtype, dx=float(dx))
        ctx.set_particles(X[sel], np.zeros((m, 3), T), one, one, one, one)
        ctx.sort()
        got = ctx.indexing()["particle_base_offset"].tolist()
        assert got == [o2["base_fma"][i][3] for i in sel], (float(dx), got, [o2["base_fma"][i][3] for i in sel])
    return len(differ)


class Stats(C.Structure):  # include/hot_mi355x.h: hot_stats
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("linesearch_trials", C.c_int32), ("linear_iterations", C.c_int32), ("vcycles", C.c_int32), ("dropped_pairs", C.c_int32),
                ("num_nodes", C.c_int32), ("num_levels", C.c_int32), ("final_scaled_residual", C.c_double), ("energy", C.c_double), ("ms", C.c_double * 8),
                ("comm_calls", C.c_int64), ("comm_bytes_index", C.c_int64), ("comm_bytes_data", C.c_int64), ("comm_calls_index", C.c_int64)]


def fixed_iterations_raw(path, prefix, n=8, iterations=5, **kw):
    """Five L-BFGS iterations of one time step of an n^3-cell cube over a sticky floor, driven through the ctypes mirror of THIS file (no
    hot_amd/binding.py): returns (dv, counters, energy).  The fixed-iteration parity test runs it on the HIP library and on the oracle, so that
    an argument the shared binding passes wrongly to both cannot cancel."""
    rng = np.random.default_rng(11)
    dx, ppc = 0.01, 8
    cells = np.stack(np.meshgrid(*(np.arange(n),) * 3, indexing="ij"), -1).reshape(-1, 3)
    X = ((500 + np.repeat(cells, ppc, 0)) + rng.random((len(cells) * ppc, 3))) * dx
    N = len(X)
    rho, E, nu = 1000.0, 5e4, 0.3
    V = np.tile(np.array([0.3, -1.0, 0.2]), (N, 1)) + 0.05 * rng.standard_normal((N, 3))
    vol = np.full(N, dx ** 3 / ppc)
    mass, mu, lam = rho * vol, np.full(N, E / (2 * (1 + nu))), np.full(N, E * nu / ((1 + nu) * (1 - 2 * nu)))
    r = Raw(path, prefix)
    h = r.create(1, dx=dx, max_iterations=iterations, cneps=1e-7, **kw)
    X, V, mass, vol, mu, lam = (np.ascontiguousarray(a, np.float64) for a in (X, V, mass, vol, mu, lam))
    r.call("set_particles", h, C.c_int64(N), vp(X), vp(V), vp(mass), None, None, vp(vol), vp(mu), vp(lam), None)
    org, nrm = np.array([0.0, 5.0, 0.0]), np.array([0.0, 1.0, 0.0])
    r.call("set_sticky_halfspaces", h, C.c_int32(1), vp(org), vp(nrm))
    r.call("sort", h)
    r.call("p2g", h)
    r.call("begin_step", h, C.c_double(1.0 / 24))
    st = Stats()
    r.call("solve", h, C.byref(st))
    dv = np.empty((st.num_nodes, 3), np.float64)
    r.call("get_dv", h, vp(dv))
    r.f("destroy", None)(h)
    return dv, {k: getattr(st, k) for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles", "dropped_pairs", "num_levels", "num_nodes")}, st.energy
