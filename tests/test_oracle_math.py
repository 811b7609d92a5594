"""Pins for the oracle's floating-point kernels.  The reference has no tests for these, so the
pins are (i) cross-library goldens (numpy.linalg svd / eigh), (ii) analytic known answers and (iii) the
reference's own consistency idea (Lib/Ziran/Sim/DiffTest.h:19-138: energy <-> gradient <-> Hessian by centred
finite differences)."""
import numpy as np
import pytest

from tests import oracle_lib as ol


def rand_F(n, seed, spread=0.4):
    rng = np.random.default_rng(seed)
    F = np.eye(3)[None] + spread * rng.standard_normal((n, 3, 3))
    return F


def cm(F):  # (n,3,3) row-major numpy -> (n,9) column-major
    return np.ascontiguousarray(np.transpose(F, (0, 2, 1)).reshape(-1, 9))


def from_cm(a):
    return np.transpose(a.reshape(-1, 3, 3), (0, 2, 1))


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-13), (np.float32, 2e-5)])
def test_svd_rotation_variant(dt, tol):
    F = rand_F(500, 1).astype(dt)
    F[0] = np.eye(3)
    F[1] = np.diag([2.0, 2.0, 0.5])
    F[2] = np.diag([1.0, -1.0, 1.0]) @ F[2]  # reflection -> negative sigma_2
    F[3] = 0
    F[4] = np.outer([1, 2, 3], [3, 2, 1])  # rank one
    U, S, V = ol.svd3(cm(F))
    U, V = from_cm(U).astype(np.float64), from_cm(V).astype(np.float64)
    S = S.astype(np.float64)
    rec = np.einsum("nij,nj,nkj->nik", U, S, V)
    scale = 1 + np.abs(F).max(axis=(1, 2))
    assert np.max(np.abs(rec - F).max(axis=(1, 2)) / scale) < 40 * tol
    assert np.allclose(np.linalg.det(U), 1, atol=50 * tol) and np.allclose(np.linalg.det(V), 1, atol=50 * tol)
    assert np.allclose(np.einsum("nij,nkj->nik", U, U), np.eye(3), atol=50 * tol)
    # ordering: s0 >= s1 >= |s2|, sign only on s2 (reference ImplicitQRSVD.h:348-353)
    assert np.all(S[:, 0] >= S[:, 1] - 50 * tol) and np.all(S[:, 1] >= np.abs(S[:, 2]) - 50 * tol)
    sv = np.linalg.svd(F.astype(np.float64), compute_uv=False)
    assert np.allclose(np.abs(S), sv, atol=100 * tol * scale[:, None])
    assert np.allclose(np.sign(S[:, 2]) * (np.abs(S[:, 2]) > 1e-3), np.sign(np.linalg.det(F.astype(np.float64))) * (np.abs(S[:, 2]) > 1e-3))


def test_make_pd_against_eigh():
    rng = np.random.default_rng(3)
    A = rng.standard_normal((300, 3, 3))
    A = A + np.transpose(A, (0, 2, 1))
    out = from_cm(ol.make_pd3(cm(A)))
    w, Q = np.linalg.eigh(A)
    ref = np.einsum("nij,nj,nkj->nik", Q, np.maximum(w, 0), Q)
    assert np.abs(out - ref).max() < 1e-12
    # idempotent on PSD input
    assert np.abs(from_cm(ol.make_pd3(cm(ref))) - ref).max() < 1e-12
    abd = rng.standard_normal((300, 3))
    o2 = ol.make_pd2(abd)
    M = np.stack([np.stack([abd[:, 0], abd[:, 1]], 1), np.stack([abd[:, 1], abd[:, 2]], 1)], 1)
    w, Q = np.linalg.eigh(M)
    r2 = np.einsum("nij,nj,nkj->nik", Q, np.maximum(w, 0), Q)
    assert np.abs(o2[:, 0] - r2[:, 0, 0]).max() < 1e-13 and np.abs(o2[:, 1] - r2[:, 0, 1]).max() < 1e-13 and np.abs(o2[:, 2] - r2[:, 1, 1]).max() < 1e-13


MU, LAM = 19230.77, 28846.15


def test_corotated_known_answers_and_numpy():
    # rigid rotation: psi = 0, P = 0
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    psi, P, _ = ol.corotated(cm(R[None]), MU, LAM, False)
    assert abs(psi[0]) < 1e-9 and np.abs(P).max() < 1e-9
    F = rand_F(200, 5, 0.25)
    psi, P, _ = ol.corotated(cm(F), MU, LAM, True)
    U, s, Vt = np.linalg.svd(F)
    det = np.linalg.det(U @ Vt)
    U[:, :, 2] *= det[:, None]
    Rp = U @ Vt
    J = np.linalg.det(F)
    Pref = 2 * MU * (F - Rp) + LAM * (J - 1)[:, None, None] * J[:, None, None] * np.transpose(np.linalg.inv(F), (0, 2, 1))
    psiref = MU * ((F - Rp) ** 2).sum(axis=(1, 2)) + 0.5 * LAM * (J - 1) ** 2
    assert np.allclose(from_cm(P), Pref, rtol=1e-10, atol=1e-8)
    assert np.allclose(psi, psiref, rtol=1e-10, atol=1e-10)


def test_corotated_diff_test():
    """DiffTest.h idea: psi <-> P <-> dPdF by centred differences, h = 1e-5."""
    F = rand_F(20, 7, 0.2)
    psi, P, dPdF = ol.corotated(cm(F), MU, LAM, False)
    h = 1e-5
    rng = np.random.default_rng(8)
    for _ in range(3):
        dF = rng.standard_normal(F.shape)
        pp, Pp, _ = ol.corotated(cm(F + h * dF), MU, LAM, False)
        pm, Pm, _ = ol.corotated(cm(F - h * dF), MU, LAM, False)
        dpsi = (pp - pm) / (2 * h)
        assert np.allclose(dpsi, (P * cm(dF)).sum(1), rtol=1e-7, atol=1e-5)
        dP_fd = (Pp - Pm) / (2 * h)
        dP = np.einsum("nab,nb->na", dPdF.reshape(-1, 9, 9), cm(dF))  # symmetric 9x9, col-major either way
        assert np.allclose(dP, dP_fd, rtol=1e-6, atol=1e-3)
        dP2 = ol.corotated_differential(cm(F), cm(dF), MU, LAM, False)
        assert np.allclose(dP2, dP, rtol=1e-10, atol=1e-7)


def test_projected_derivative_is_psd_and_consistent():
    F = rand_F(100, 9, 0.5)
    _, _, d0 = ol.corotated(cm(F), MU, LAM, False)
    _, _, d1 = ol.corotated(cm(F), MU, LAM, True)
    d0 = d0.reshape(-1, 9, 9)
    d1 = d1.reshape(-1, 9, 9)
    assert np.abs(d1 - np.transpose(d1, (0, 2, 1))).max() < 1e-7
    w1 = np.linalg.eigvalsh(d1)
    assert w1.min() > -1e-6 * np.abs(w1).max()
    # the projection clamps the spectrum of the un-projected derivative block-wise: where d0 is already PSD they agree
    w0 = np.linalg.eigvalsh(d0)
    ok = w0.min(axis=1) > 1e-6
    assert ok.sum() > 0
    assert np.allclose(d0[ok], d1[ok], rtol=1e-9, atol=1e-6)
    dF = np.random.default_rng(1).standard_normal(F.shape)
    dP = ol.corotated_differential(cm(F), cm(dF), MU, LAM, True)
    assert np.allclose(dP, np.einsum("nab,nb->na", d1, cm(dF)), rtol=1e-10, atol=1e-6)


def test_plasticity_known_answers():
    F = rand_F(50, 11, 0.05)
    n = len(F)
    mu, lam = np.full(n, MU), np.full(n, LAM)
    # huge yield stress: nothing changes
    F1, *_ = ol.plasticity(1, cm(F), mu, lam, np.ones(n), yield_stress=1e30)
    assert np.array_equal(F1, cm(F))
    # tiny yield stress: deviatoric Kirchhoff stress lands on the yield surface
    ys = 50.0
    F2, *_ = ol.plasticity(1, cm(F), mu, lam, np.ones(n), yield_stress=ys)
    s = np.linalg.svd(from_cm(F2), compute_uv=False)
    J = s.prod(1)
    tau = 2 * MU * (s - 1) * s + (LAM * (J - 1) * J)[:, None]
    dev = tau - tau.mean(1, keepdims=True)
    assert np.all(np.linalg.norm(dev, axis=1) <= np.sqrt(2.0 / 3.0) * ys * 1.2 + 1e-6)
    # snow: singular values clamped to [1-theta_c, 1+theta_s], hardening follows Jp
    snow = (10.0, 2e-2, 7.5e-3, 0.6, 20.0)
    F3, mu3, lam3, Jp3 = ol.plasticity(2, cm(F), mu, lam, np.ones(n), snow=snow)
    s3 = np.linalg.svd(from_cm(F3), compute_uv=False)
    assert s3.max() <= 1 + snow[2] + 1e-12 and s3.min() >= 1 - snow[1] - 1e-12
    assert np.allclose(Jp3, np.clip(np.linalg.det(F) / np.linalg.det(from_cm(F3)), snow[3], snow[4]))
    assert np.allclose(mu3, MU * np.exp(snow[0] * (1 - Jp3)))


def test_oracle_incomplete_cholesky_top_solver(oracle):
    """-coarseSolver 7 in the oracle (block IC(0) in the smoother's order, Eigen's shift strategy: oracle/sim_matrix.hpp setup_ic): the V-cycle
    with the IC top solve is a symmetric positive operator on 1, 2 and 3 levels, and the L-BFGS solve behind it converges."""
    from tests import pipeline_checks as pc
    for levelCnt in (1, 2, 3):
        ctx, c = pc.make_ctx(oracle, n=6, levelCnt=levelCnt, coarseSolver=7, cneps=1e-7, max_iterations=300)
        pc.prepare(ctx)
        e0 = ctx.update_state(ctx.get_dv())
        st = ctx.solve()
        assert st["converged"] == 1 and st["energy"] < e0
        rng = np.random.default_rng(1)
        x, y = ctx.project(rng.standard_normal((ctx.Nn, 3))), ctx.project(rng.standard_normal((ctx.Nn, 3)))
        Mx, My = ctx.vcycle(x), ctx.vcycle(y)
        assert abs((y * Mx).sum() - (x * My).sum()) < 1e-10 * abs((y * Mx).sum())
        assert (x * Mx).sum() > 0 and (y * My).sum() > 0
