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
                ("profile", C.c_int32), ("debug_store", C.c_int32), ("useBaselineMultigrid", C.c_int32), ("gs_chain", C.c_int32), ("gs_sub_block", C.c_int32), ("shard_gs", C.c_int32), ("shard_replicated", C.c_int32), ("ls_energy_only", C.c_int32), ("reserved", C.c_int32 * 7)]


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
