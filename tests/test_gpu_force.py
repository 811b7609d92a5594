"""GPU parity: objective pieces (state update, energy, residual, projection, CN tolerances, matrix-free product)
against the CPU oracle on identical inputs."""
import numpy as np
import pytest

from tests import pipeline_checks as pc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(np.asarray(b, np.float64)).max(), 1e-300)


@pytest.mark.parametrize("dtype,tol", [(1, 1e-10), (0, 1e-6)])  # fp32 against the oracle's float arithmetic: measured 2e-7 (dv0), 6e-11 (energy), <= 4e-6 (stress; bound 20 tol = 2e-5); round 2: 1.5e-5 / 3e-4 before the B-spline fraction was evaluated with the exact product (hot_common.h bspline)
@pytest.mark.parametrize("n", [6, 12])
def test_objective_pieces_against_oracle(hotlib, oracle, dtype, tol, n):
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=n, dtype=dtype, bc=True)
        pc.prepare(ctx)
        dv0 = ctx.get_dv()
        rng = np.random.default_rng(7)
        dv = dv0 + 0.02 * rng.standard_normal(dv0.shape).astype(dv0.dtype)
        dv = ctx.project(dv) if False else dv
        e = ctx.update_state(dv)
        st = ctx.particle_state()
        r = ctx.residual()
        tol_n = ctx.cn_tolerance()
        x = rng.standard_normal(dv0.shape)
        hx = ctx.matfree_multiply(x)
        pv = ctx.project(x)
        out[name] = dict(dv0=dv0, e=e, r=r, tol=tol_n, hx=hx, pv=pv, **st)
    g, c = out["gpu"], out["cpu"]
    assert rel(g["dv0"], c["dv0"]) < tol
    assert abs(g["e"] - c["e"]) < tol * max(abs(c["e"]), 1e-6) * 10
    for k in ("F", "gradV", "stress", "r", "tol", "hx", "pv"):
        assert rel(g[k], c[k]) < tol * 20, (k, rel(g[k], c[k]))


def test_diff_test_on_gpu(hotlib):
    assert pc.check_diff_test(hotlib) < 1e-5


def test_explicit_bc_list_matches_halfspace(hotlib, oracle):
    """hot_set_bc with an explicit collision-node list (sticky + slip) against the oracle."""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=6, bc=False, boundaryType=1)
        ctx.sort()
        ctx.p2g()
        coords = ctx.grid()["id2coord"]
        ymin = coords[:, 1].min()
        sticky = np.where(coords[:, 1] <= ymin + 1)[0]
        slip = np.where(coords[:, 0] == coords[:, 0].min())[0]
        slip = np.setdiff1d(slip, sticky)
        ids = np.concatenate([sticky, slip]).astype(np.int32)
        nc = len(ids)
        P = np.zeros((nc, 9))
        R = np.tile(np.eye(3).reshape(1, 9), (nc, 1))
        # slip nodes: normal (1,0,0) -> P = I - n n^T, R = I (already aligned with x)
        P[len(sticky):] = np.diag([0.0, 1.0, 1.0]).reshape(9)
        flags = np.zeros(nc, np.uint8)
        flags[len(sticky):] = 1
        ctx.set_bc(ids, P, R, R, flags)
        ctx.begin_step(1.0 / 24)
        dv = ctx.get_dv()
        ctx.update_state(dv)
        out[name] = (dv, ctx.residual())
    assert rel(out["gpu"][0], out["cpu"][0]) < 1e-10
    assert rel(out["gpu"][1], out["cpu"][1]) < 1e-9


def _exact_half_fr2_and_j(F):
    """|F - R|_F^2 / 2 and J - 1 in 50-digit arithmetic (singular values with the reference's convention: an inverted element's negative one last)."""
    import mpmath as mp
    mp.mp.dps = 50
    Fm = mp.matrix(F.tolist())
    _, S, _ = mp.svd_r(Fm)
    s = [S[i] for i in range(3)]
    d = mp.det(Fm)
    if d < 0:
        s[2] = -s[2]
    return float(sum((x - 1) ** 2 for x in s) / 2), float(d - 1)


@pytest.mark.parametrize("dtype", [1, 0])
def test_trial_energy_without_svd(hotlib, oracle, dtype):
    """The line search's energy-only evaluation (hot_constitutive.h corotated_psi_invariants: the polar decomposition's trace from the
    invariants of F^T F, no SVD) against 50-digit arithmetic, against the full evaluation (one SVD, mu |F - R|^2: CorotatedIsotropic.h:151-155)
    of the same library and against the oracle's; F = rotation x (I + strain x random) at strains 1e-8 .. 1, plus compressed / inverted
    samples where the evaluation falls back on the singular values."""
    pytest.importorskip("mpmath")
    T = np.float64 if dtype == 1 else np.float32
    eps = np.finfo(T).eps
    mu, lam = 3.0e4, 7.0e4
    rng = np.random.default_rng(11)
    Fs, strain = [], []
    for scale in (1e-8, 1e-6, 1e-4, 1e-2, 0.1, 0.3, 1.0):
        k = 0
        while k < 40:
            Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
            if np.linalg.det(Q) < 0:
                Q[:, 0] *= -1
            F = (Q if k % 2 else np.eye(3)) @ (np.eye(3) + scale * rng.standard_normal((3, 3)))
            if np.linalg.det(F) < 0.15:
                continue
            Fs.append(F), strain.append(scale)
            k += 1
    n_regular = len(Fs)
    for s3 in (0.05, 1e-3, 0.0, -1e-3, -0.3, -0.9, -1.0):  # det F <= 0.1: the singular-value path
        Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        Q2, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        Fs.append(Q @ np.diag([1.2, 1.0, s3]) @ Q2.T * np.sign(np.linalg.det(Q) * np.linalg.det(Q2))), strain.append(1.0)
    F = np.stack(Fs).astype(T)
    Fcm = np.ascontiguousarray(F.transpose(0, 2, 1).reshape(-1, 9))  # column-major 3x3
    ctx = hotlib.context(dtype=dtype)
    trial = ctx.trial_energy(Fcm, mu, lam).astype(np.float64)
    full, _, _ = ctx.constitutive_eval(Fcm, mu, lam, project=0, derivative=False)
    octx = oracle.context(dtype=dtype)
    ofull, _, _ = octx.constitutive_eval(Fcm, mu, lam, project=0, derivative=False)
    exact = np.array([_exact_half_fr2_and_j(f.astype(np.float64)) for f in F])
    psi_exact = 2 * mu * exact[:, 0] + 0.5 * lam * exact[:, 1] ** 2
    strain = np.array(strain)
    # what any evaluation from F in T can deliver: F^T F - I (or sigma - 1) carries eps absolute, the energy is quadratic in a strain of size `strain`
    bound = (40 if dtype == 1 else 200) * eps * ((mu + lam) * strain * (1 + strain) + psi_exact) + 1e-300
    err_trial, err_full = np.abs(trial - psi_exact), np.abs(full.astype(np.float64) - psi_exact)
    assert (err_trial / bound).max() < 1, (err_trial / bound).max()
    assert (err_full / bound).max() < 1, (err_full / bound).max()
    assert (np.abs(trial - ofull.astype(np.float64)) / bound).max() < 2
    # and it is not the less accurate of the two where it replaces the SVD
    assert np.median(err_trial[:n_regular] / np.maximum(err_full[:n_regular], 1e-300)) < 1.5


@pytest.mark.parametrize("dtype,tol", [(1, 1e-10), (0, 2e-5)])
def test_state_pass_with_inverted_elements_against_oracle(hotlib, oracle, dtype, tol):
    """A state update far outside the elastic range (nodal velocity increments that turn some elements inside out: det F < 0.1, where the sign
    convention of the singular values matters and the energy-only form falls back on them), then a moderate one in the same time step: energy, trial F
    and stresses as the oracle's."""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, dtype=dtype, bc=True)
        pc.prepare(ctx)
        dv0 = ctx.get_dv()
        rng = np.random.default_rng(5)
        big = dv0 + (0.6 * c["dx"] * 24.0) * rng.standard_normal(dv0.shape).astype(dv0.dtype)  # displacements of ~0.6 dx over the step
        e1 = ctx.update_state(big)
        st1 = ctx.particle_state()
        small = dv0 + 0.02 * rng.standard_normal(dv0.shape).astype(dv0.dtype)
        e2 = ctx.update_state(small)
        st2 = ctx.particle_state()
        out[name] = (e1, st1, e2, st2)
    g, c_ = out["gpu"], out["cpu"]
    F = c_[1]["F"].astype(np.float64).reshape(-1, 3, 3)
    assert (np.linalg.det(F) < 0.1).any()  # the case is what it says
    for e_g, st_g, e_c, st_c in ((g[0], g[1], c_[0], c_[1]), (g[2], g[3], c_[2], c_[3])):
        assert abs(e_g - e_c) < tol * 10 * max(abs(e_c), 1e-6)
        for k in ("F", "stress"):
            assert rel(st_g[k], st_c[k]) < tol * 20, (k, rel(st_g[k], st_c[k]))
