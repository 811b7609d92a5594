"""Library-agnostic consistency checks of the hot path (run against the CPU oracle on CPU and against the HIP
library on the GPU).  They re-create the reference's opt-in runtime self-checks:
runDiffTest (Lib/Ziran/Sim/DiffTest.h:19-138), matrixSanityCheck (Projects/multigrid/ImplicitSolver.h:698-739),
symmetricSanityCheck / PDSanityCheck (Projects/multigrid/SquareMatrix.h:84-194), checkPreconditioningMatrix
(Lib/Ziran/Math/Nonlinear/LBFGS.h:95-175), plus analytic known answers."""
import os

import numpy as np

from hot_amd import synth


def make_ctx(lib, n=6, dtype=1, bc=True, seed=123, noise=0.1, E=5e4, ppc=8, cells=None, **kw):
    T = np.float64 if dtype == 1 else np.float32
    for item in os.environ.get("HOT_TEST_CFG", "").split(","):  # hot_config overrides from tests/test_gpu_variants.py
        if item:
            k, v = item.split("=")
            kw.setdefault(k, int(v))
    c = synth.cube_cloud(n, ppc=ppc, dtype=T, seed=seed, noise=noise, E=E, cells=cells)
    kw.setdefault("debug_store", 1)  # tests read back per-particle grad v
    ctx = lib.context(dtype=dtype, dx=c["dx"], gravity=(0, -9.8, 0), **kw)
    ctx.set_particles(c["X"], c["V"], c["mass"], c["vol"], c["mu"], c["lam"])
    if bc:
        o, nrm = synth.sticky_floor(5.0, c["dx"])
        ctx.set_sticky_halfspaces(o, nrm)
    return ctx, c


def prepare(ctx, dt=1.0 / 24):
    ctx.sort()
    ctx.p2g()
    ctx.begin_step(dt)


def check_transfer_conservation(lib, dtype, tol):
    ctx, c = make_ctx(lib, dtype=dtype, bc=False)
    ctx.sort()
    ctx.p2g()
    g = ctx.grid()
    m, v = g["mass"].astype(np.float64), g["v"].astype(np.float64)
    mp = c["mass"].astype(np.float64)
    assert abs(m.sum() - mp.sum()) < tol * mp.sum()
    mom = (m[:, None] * v).sum(0)
    momp = (mp[:, None] * c["V"].astype(np.float64)).sum(0)
    assert np.abs(mom - momp).max() < tol * np.abs(mp[:, None] * c["V"]).sum()


def check_apic_affine_reproduction(lib, dtype, tol):
    """P2G then G2P with dv = 0 reproduces an affine velocity field exactly under APIC (C = A)."""
    T = np.float64 if dtype == 1 else np.float32
    c = synth.cube_cloud(6, ppc=8, dtype=T)
    A = np.array([[0.1, -0.3, 0.2], [0.05, 0.2, -0.1], [0.3, 0.1, -0.25]])
    b = np.array([0.3, -0.2, 0.1])
    X = c["X"].astype(np.float64)
    V = X @ A.T + b
    C = np.tile(A.T.reshape(1, 9), (len(X), 1))  # column-major A
    ctx = lib.context(dtype=dtype, dx=c["dx"], gravity=(0, 0, 0))
    ctx.set_particles(c["X"], V, c["mass"], c["vol"], c["mu"], c["lam"], C_=C)
    ctx.sort()
    ctx.p2g()
    g = ctx.grid()
    xi = g["id2coord"].astype(np.float64) * c["dx"]
    assert np.abs(g["v"] - (xi @ A.T + b)).max() < tol
    ctx.begin_step(1e-3)
    ctx.set_dv(np.zeros((ctx.Nn, 3)))
    ctx.g2p(0.0)
    p = ctx.get_particles()
    assert np.abs(p["V"] - V).max() < tol
    assert np.abs(p["C"] - C).max() < tol * 100  # D^{-1} = 4/dx^2 amplifies round-off


def check_diff_test(lib, dtype=1):
    """energy <-> residual: E(dv+h d) - E(dv-h d) = -2h <r, d> + O(h^3)   (r = -dE/d dv)."""
    ctx, c = make_ctx(lib, dtype=dtype, bc=False, project=0)  # un-projected dP/dF is the true Hessian
    prepare(ctx)
    dv = ctx.get_dv()
    rng = np.random.default_rng(123)
    ctx.update_state(dv)
    r = ctx.residual().astype(np.float64)
    d = rng.standard_normal(dv.shape) * 0.05
    errs = []
    for h in (2.0 ** -6, 2.0 ** -8, 2.0 ** -10):
        ep = ctx.update_state(dv + h * d)
        em = ctx.update_state(dv - h * d)
        fd = (ep - em) / (2 * h)
        errs.append(abs(fd + (r * d).sum()) / abs((r * d).sum()))
    assert errs[-1] < 1e-3 and errs[-1] < errs[0]
    # residual <-> matrix-free Hessian product
    h = 1e-4
    ctx.update_state(dv + h * d)
    rp = ctx.residual().astype(np.float64)
    ctx.update_state(dv - h * d)
    rm = ctx.residual().astype(np.float64)
    ctx.update_state(dv)
    Hd = ctx.matfree_multiply(d).astype(np.float64)
    fd = -(rp - rm) / (2 * h)
    return np.abs(fd - Hd).max() / np.abs(Hd).max()


def check_matrix_vs_matfree(lib, dtype=1, project=1):
    ctx, c = make_ctx(lib, dtype=dtype, bc=False, project=project)
    prepare(ctx)
    ctx.update_state(ctx.get_dv())
    ctx.build_hessian()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((ctx.Nn, 3))
    y1 = ctx.spmv(0, x).astype(np.float64)
    y2 = ctx.matfree_multiply(x).astype(np.float64)
    z = rng.standard_normal((ctx.Nn, 3))
    y3 = ctx.spmv(0, z).astype(np.float64)
    sym = abs((z * y1).sum() - (x * y3).sum()) / abs((z * y1).sum())
    pd = (x * y1).sum()
    return np.abs(y1 - y2).max() / np.abs(y1).max(), sym, pd


def ell_to_scipy(col, val, ncols):
    import scipy.sparse as sp
    n, k = col.shape
    rows = np.repeat(np.arange(n), k * 9)
    blocks = val.reshape(n, k, 3, 3).transpose(0, 1, 3, 2)  # column-major 3x3 -> [r][c]
    r = (rows.reshape(n, k, 3, 3) * 0 + (3 * np.arange(n))[:, None, None, None] + np.arange(3)[None, None, :, None])
    cidx = 3 * col[:, :, None, None] + np.arange(3)[None, None, None, :]
    cidx = np.broadcast_to(cidx, blocks.shape)
    r = np.broadcast_to(r, blocks.shape)
    return sp.coo_matrix((blocks.ravel().astype(np.float64), (r.ravel(), cidx.ravel())), shape=(3 * n, 3 * ncols)).tocsr()


def check_galerkin(lib, dtype=1, levelCnt=3, tol=1e-10):
    import scipy.sparse as sp
    ctx, c = make_ctx(lib, dtype=dtype, bc=True, levelCnt=levelCnt, n=8)
    prepare(ctx)
    ctx.update_state(ctx.get_dv())
    ctx.build_hessian()
    ctx.build_mg()
    mats = []
    for l in range(levelCnt):
        col, val = ctx.matrix(l)
        mats.append(ell_to_scipy(col, val, col.shape[0]))
    for l in range(levelCnt - 1):
        pc, pw = ctx.prolongation(l)
        nf, ncoarse = pc.shape[0], mats[l + 1].shape[0] // 3
        P1 = sp.coo_matrix((pw.ravel().astype(np.float64), (np.repeat(np.arange(nf), 8), pc.ravel())), shape=(nf, ncoarse)).tocsr()
        assert np.allclose(np.asarray(P1.sum(1)).ravel(), 1.0)  # rows of P sum to one
        P3 = sp.kron(P1, sp.identity(3)).tocsr()
        RAP = (P3.T @ mats[l] @ P3).tocsr()
        diff = abs(RAP - mats[l + 1]).max()
        assert diff < tol * abs(mats[l + 1]).max(), (l, diff)
        # coarse coordinates: every fine node's parents exist, numbering is first-touch
        fine = ctx.level(l)["id2coord"]
        coarse = ctx.level(l + 1)["id2coord"]
        seen = {}
        for i in range(len(fine)):
            x, y, z = fine[i]
            for nx in (x // 2, x // 2 + 1):
                for ny in (y // 2, y // 2 + 1):
                    for nz in (z // 2, z // 2 + 1):
                        if (nx > x // 2 and x % 2 == 0) or (ny > y // 2 and y % 2 == 0) or (nz > z // 2 and z % 2 == 0):
                            continue
                        if (nx, ny, nz) not in seen:
                            seen[(nx, ny, nz)] = len(seen)
        assert len(seen) == len(coarse)
        assert all(seen[tuple(cc)] == j for j, cc in enumerate(coarse.tolist()))
        # operators agree with the exported matrices
        x = np.random.default_rng(1).standard_normal((nf, 3))
        assert np.allclose(ctx.restrict(l, x).astype(np.float64).ravel(), P3.T @ x.ravel(), rtol=1e-6 if dtype == 0 else 1e-11, atol=1e-6 if dtype == 0 else 1e-11)
        xc = np.random.default_rng(2).standard_normal((ncoarse, 3))
        assert np.allclose(ctx.prolong(l, xc).astype(np.float64).ravel(), P3 @ xc.ravel(), rtol=1e-6 if dtype == 0 else 1e-11, atol=1e-6 if dtype == 0 else 1e-11)
    return ctx, mats


def check_vcycle_spd(lib, dtype=1):
    """With linear smoothers on every level (symmetric GS everywhere) the V-cycle is a symmetric PD operator."""
    ctx, c = make_ctx(lib, dtype=dtype, bc=True, levelCnt=3, coarseSolver=5, n=8)
    prepare(ctx)
    ctx.update_state(ctx.get_dv())
    ctx.build_hessian()
    ctx.build_mg()
    rng = np.random.default_rng(9)
    x = ctx.project(rng.standard_normal((ctx.Nn, 3)))
    y = ctx.project(rng.standard_normal((ctx.Nn, 3)))
    Mx = ctx.vcycle(x).astype(np.float64)
    My = ctx.vcycle(y).astype(np.float64)
    sym = abs((y * Mx).sum() - (x * My).sum()) / abs((y * Mx).sum())
    assert (x * Mx).sum() > 0 and (y * My).sum() > 0
    # A-norm contraction: stationary iteration u += M (b - A u) monotonically decreases 1/2 u'Au - b'u
    # (the matrix has cond ~ 1e8 from low-mass boundary nodes, so the residual 2-norm is not monotone)
    b = x
    u = np.zeros_like(b, dtype=np.float64)
    energies = [0.0]
    for _ in range(3):
        r = b - ctx.spmv(0, u).astype(np.float64)
        u = u + ctx.vcycle(r).astype(np.float64)
        energies.append(0.5 * (u * ctx.spmv(0, u).astype(np.float64)).sum() - (b * u).sum())
    return sym, energies
