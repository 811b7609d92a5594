#!/usr/bin/env python3
"""Generator of tests/golden/fp_golden.npz: floating-point pins computed with numpy only (numpy.linalg.svd / eigh / det /
inv and plain array arithmetic) — no code of this repository is imported, neither the CPU oracle nor the ctypes binding.
The GPU tests compare the HIP library's outputs with these vectors directly (tests/test_gpu_golden.py), the CPU tests
do the same for the oracle (tests/test_oracle_golden.py), so this part of the floating-point path is pinned by a third,
independent restatement in another language and library.

What is restated (formulas only; reference files for the reader):
  * fixed-corotated energy density, first Piola stress and dP/dF in the SVD frame, with and without the PSD projection of
    the A / B blocks (Lib/Ziran/Physics/ConstitutiveModel/CorotatedIsotropic.h:110-230, SvdBasedIsotropicHelper.h:223-282,
    Lib/Ziran/Math/Linear/EigenDecomposition.h:126-135 makePD = clamp negative eigenvalues to zero);
  * von Mises and snow return mappings in singular-value space (Lib/Ziran/Physics/PlasticityApplier.cpp:96-131, :18-50);
  * APIC particle-to-grid transfer with quadratic B-splines (Lib/MPM/MpmSimulationBase.cpp:611-656, Lib/Ziran/Math/Splines/BSplines.h:55-81).
Self-checks run at generation time: the un-projected dP/dF is verified against centred finite differences of P, P against
centred differences of psi; P2G conserves mass and momentum.

Run:  python tests/golden/make_fp_golden.py      (deterministic; rewrites fp_golden.npz)
"""
import os

import numpy as np

MU, LAM = 19230.769230769230, 28846.153846153848  # E = 5e4, nu = 0.3


def cm(F):  # (n,3,3) -> (n,9) column-major
    return np.ascontiguousarray(np.transpose(F, (0, 2, 1)).reshape(-1, 9))


def rotation_svd(F):
    """F = U diag(s) V^T with U, V rotations, s0 >= s1 >= |s2|, the sign of det F carried by s2."""
    U, s, Vt = np.linalg.svd(F)
    V = np.transpose(Vt, (0, 2, 1)).copy()
    s = s.copy()
    du, dv = np.linalg.det(U), np.linalg.det(V)
    U[du < 0, :, 2] *= -1
    s[du < 0, 2] *= -1
    V[dv < 0, :, 2] *= -1
    s[dv < 0, 2] *= -1
    return U, s, V


def clamp_small_magnitude(x, eps):
    return np.where(x < -eps, x, np.where(x < 0, -eps, np.where(x < eps, eps, x)))


def make_pd(M):
    w, Q = np.linalg.eigh(M)
    return np.einsum("nij,nj,nkj->nik", Q, np.maximum(w, 0.0), Q)


def corotated(F, mu, lam, project):
    U, s, V = rotation_svd(F)
    R = U @ np.transpose(V, (0, 2, 1))
    J = s.prod(1)
    cof = np.empty_like(F)  # J F^-T as the cofactor matrix (defined for singular F as well)
    for i in range(3):
        for j in range(3):
            a, b = [k for k in range(3) if k != i], [k for k in range(3) if k != j]
            cof[:, i, j] = (-1) ** (i + j) * (F[:, a[0], b[0]] * F[:, a[1], b[1]] - F[:, a[0], b[1]] * F[:, a[1], b[0]])
    psi = mu * ((F - R) ** 2).sum((1, 2)) + 0.5 * lam * (J - 1) ** 2
    P = 2 * mu * (F - R) + (lam * (J - 1))[:, None, None] * cof
    # ---- derivative in the singular-value frame
    _2mu, _lam = 2 * mu, lam * (J - 1)
    sp = np.stack([s[:, 1] * s[:, 2], s[:, 0] * s[:, 2], s[:, 0] * s[:, 1]], 1)
    ps = _2mu * (s - 1) + _lam[:, None] * sp  # psi_i
    A = np.empty_like(F)
    for i in range(3):
        A[:, i, i] = _2mu + lam * sp[:, i] ** 2
    for i, j, k in ((0, 1, 2), (0, 2, 1), (1, 2, 0)):
        A[:, i, j] = A[:, j, i] = _lam * s[:, k] + lam * sp[:, i] * sp[:, j]
    eps = 1e-6
    B = {}
    for (i, j, k) in ((0, 1, 2), (1, 2, 0), (0, 2, 1)):
        m = _2mu - _lam * s[:, k]
        p = (ps[:, i] + ps[:, j]) / clamp_small_magnitude(s[:, i] + s[:, j], eps)
        B[(i, j)] = np.stack([np.stack([(m + p) / 2, (m - p) / 2], 1), np.stack([(m - p) / 2, (m + p) / 2], 1)], 1)
    if project:
        A = make_pd(A)
        B = {k: make_pd(v) for k, v in B.items()}
    # K[(a,b),(c,d)] in the frame: diagonal-diagonal through A, (i,j)/(j,i) pairs through the 2x2 blocks
    n = F.shape[0]
    K = np.zeros((n, 3, 3, 3, 3))
    for a in range(3):
        for c in range(3):
            K[:, a, a, c, c] = A[:, a, c]
    for (i, j), b in B.items():
        K[:, i, j, i, j] = b[:, 0, 0]
        K[:, i, j, j, i] = b[:, 0, 1]
        K[:, j, i, i, j] = b[:, 1, 0]
        K[:, j, i, j, i] = b[:, 1, 1]
    # dP_{ij}/dF_{rs} = U_ia V_jb K_{ab,cd} U_rc V_sd
    D = np.einsum("nia,njb,nabcd,nrc,nsd->nijrs", U, V, K, U, V)
    # 9x9, column-major vectorisation of both index pairs: row = i + 3 j, col = r + 3 s
    D9 = np.transpose(D, (0, 2, 1, 4, 3)).reshape(n, 9, 9)
    return psi, P, D9


def von_mises(F, mu, lam, yield_stress):
    U, s, V = rotation_svd(F)
    s = np.maximum(s, 1e-4)
    J = s.prod(1)
    tau = 2 * mu * (s - 1) * s + (lam * (J - 1) * J)[:, None]
    tr = tau.sum(1, keepdims=True)
    dev = tau - tr / 3
    nrm = np.sqrt((dev ** 2).sum(1))
    ty = np.sqrt(2.0 / 3.0) * yield_stress
    out = F.copy()
    hit = nrm - ty > 0
    alpha = ty / nrm[hit]
    tau_new = alpha[:, None] * dev[hit] + tr[hit] / 3
    disc = mu * mu - 2 * mu * ((lam * (J[hit] - 1) * J[hit])[:, None] - tau_new)
    sn = (mu + np.sqrt(disc)) / (2 * mu)
    out[hit] = np.einsum("nij,nj,nkj->nik", U[hit], sn, V[hit])
    return out, hit


def snow(F, mu, lam, Jp, psi, theta_c, theta_s, min_Jp, max_Jp):
    U, s, V = rotation_svd(F)
    sc = np.clip(s, 1 - theta_c, 1 + theta_s)
    out = np.einsum("nij,nj,nkj->nik", U, sc, V)
    Jp_new = np.clip(Jp * np.linalg.det(F) / sc.prod(1), min_Jp, max_Jp)
    hard = np.exp(psi * (Jp - Jp_new))
    return out, mu * hard, lam * hard, Jp_new


def bspline(x):
    """quadratic B-spline of index-space coordinate x: base node and the three weights"""
    base = np.floor(x - 0.5).astype(np.int64)
    d0 = x - base
    w = np.stack([0.5 * (1.5 - d0) ** 2, 0.75 - (d0 - 1) ** 2, 0.5 * (d0 - 0.5) ** 2], -1)
    return base, w


def p2g(X, V, Cm, mass, dx):
    """APIC particle-to-grid: node mass and velocity keyed by integer node coordinates"""
    base, w = bspline(X / dx)  # (n,3), (n,3,3)
    acc = {}
    for p in range(X.shape[0]):
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    node = (base[p, 0] + i, base[p, 1] + j, base[p, 2] + k)
                    wt = w[p, 0, i] * w[p, 1, j] * w[p, 2, k]
                    xi = np.array(node) * dx
                    mom = mass[p] * (V[p] + Cm[p] @ (xi - X[p]))
                    a = acc.setdefault(node, np.zeros(4))
                    a[0] += wt * mass[p]
                    a[1:] += wt * mom
    nodes = sorted(k for k, a in acc.items() if a[0] != 0)
    m = np.array([acc[k][0] for k in nodes])
    v = np.array([acc[k][1:] / acc[k][0] for k in nodes])
    return np.array(nodes, np.int32), m, v


def main():
    rng = np.random.default_rng(20260928)
    n = 160
    F = np.eye(3)[None] + 0.35 * rng.standard_normal((n, 3, 3))
    # special members: identity, pure rotation, inverted (det < 0), nearly equal singular values, one tiny singular value,
    # strong compression, strong stretch
    th = 0.7
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    F[0] = np.eye(3)
    F[1] = Rz
    F[2] = np.diag([1.0, -1.0, 1.0]) @ F[2]
    F[3] = np.diag([1.0, 1.0, -0.3]) @ Rz
    Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    F[4] = Q @ np.diag([1.2, 1.2 + 1e-7, 0.9]) @ Rz.T
    F[5] = Q @ np.diag([1.1, 0.8, 1e-5]) @ Rz
    F[6] = Q @ np.diag([0.2, 0.15, 0.1]) @ Rz
    F[7] = Q @ np.diag([3.0, 2.5, 2.0]) @ Rz.T
    F[8] = Q @ np.diag([1.3, 0.7, -0.7 + 1e-4]) @ Rz  # sigma1 + sigma2 close to zero: the clamped division
    psi, P, D = corotated(F, MU, LAM, False)
    _, _, Dp = corotated(F, MU, LAM, True)
    # ---- self-checks by centred differences on the generic members (the special ones sit at kinks of the SVD frame)
    h = 1e-6
    gen = np.arange(9, n)
    for _ in range(3):
        dF = rng.standard_normal(F.shape)
        pp, Pp, _ = corotated(F + h * dF, MU, LAM, False)
        pm, Pm, _ = corotated(F - h * dF, MU, LAM, False)
        assert np.allclose(((pp - pm) / (2 * h))[gen], (P * dF).sum((1, 2))[gen], rtol=1e-6, atol=1e-4)
        dP = np.einsum("nab,nb->na", D, cm(dF))
        assert np.allclose(cm((Pp - Pm) / (2 * h))[gen], dP[gen], rtol=2e-5, atol=2e-2), np.abs(cm((Pp - Pm) / (2 * h)) - dP)[gen].max()
    assert np.abs(D - np.transpose(D, (0, 2, 1))).max() < 1e-6 and np.linalg.eigvalsh(Dp).min() > -1e-6
    # ---- plasticity
    Fpl = np.eye(3)[None] + 0.04 * rng.standard_normal((n, 3, 3))
    ys = 2000.0
    vm, hit = von_mises(Fpl, MU, LAM, ys)
    assert 0.3 < hit.mean() < 1.0, hit.mean()
    sn = (5.0, 2e-2, 7.5e-3, 0.6, 20.0)
    Jp0 = 1 + 0.05 * rng.standard_normal(n)
    sF, smu, slam, sJp = snow(Fpl, MU, LAM, Jp0, *sn)
    # ---- APIC P2G of a small cloud near (5,5,5), dx = 0.01
    dx = 0.01
    npart = 150
    X = 5.0 + dx * (1.0 + 3.0 * rng.random((npart, 3)))
    Vp = rng.standard_normal((npart, 3))
    Cm = 5.0 * rng.standard_normal((npart, 3, 3))
    mass = 2000.0 * dx ** 3 / 8 * (0.5 + rng.random(npart))
    nodes, gm, gv = p2g(X, Vp, Cm, mass, dx)
    assert abs(gm.sum() - mass.sum()) < 1e-12 * mass.sum()
    momp = (mass[:, None] * Vp).sum(0)  # sum_i w_i (x_i - x_p) = 0: the affine part carries no net momentum
    assert np.abs((gm[:, None] * gv).sum(0) - momp).max() < 1e-10 * np.abs(mass[:, None] * Vp).sum()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp_golden.npz")
    np.savez_compressed(out, mu=MU, lam=LAM, F=cm(F), psi=psi, P=cm(P), dPdF=D.reshape(n, 81), dPdF_projected=Dp.reshape(n, 81),
                        pl_F=cm(Fpl), vm_yield=ys, vm_F=cm(vm), vm_hit=hit, snow_params=np.array(sn), snow_Jp0=Jp0, snow_F=cm(sF), snow_mu=smu, snow_lam=slam,
                        snow_Jp=sJp, p2g_dx=dx, p2g_X=X, p2g_V=Vp, p2g_C=cm(Cm), p2g_mass=mass, p2g_nodes=nodes, p2g_node_mass=gm, p2g_node_v=gv)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
