#!/usr/bin/env python3
"""One whole tiny backward-Euler step of the HOT hot path restated with numpy only: no code of this repository is imported (neither the
CPU oracle nor the ctypes binding), dense matrices and plain loops throughout.  It is the third, independent restatement that pins the
rows the constitutive goldens (make_fp_golden.py) do not reach: assembled Hessian with its boundary projection, trilinear prolongation
and Galerkin coarse matrix, the coloured symmetric Gauss-Seidel sweep in the reference's (colour, first-touch 4^3 block, node) order, the
V-cycle and the L-BFGS recursion with its energy line search.

What is restated (formulas only; reference files for the reader, paths relative to the reference tree):
  * APIC P2G, v = mv / m                                        Lib/MPM/MpmSimulationBase.cpp:611-656
  * dv0 = g dt (collision nodes: -v), vn = v                    Projects/multigrid/MultigridSimulation.h:167-186
  * trial F = (I + dt grad(vn + dv)) Fn, energy, residual        Lib/MPM/Force/MpmForceBase.cpp:213-248,309-328, Projects/multigrid/ImplicitSolver.h:128-155,254-275
  * Hessian M + dt^2 sum_p V_p (Fn^T grad w_i)^T dP/dF (Fn^T grad w_j), sticky rows / columns -> identity   ImplicitSolver.h:470-603
  * P (trilinear, first-touch coarse numbering), A_1 = P^T A_0 P  Projects/multigrid/MultigridPreconditioner.h:445-466,554-703
  * markColors, gs_smooth, cg_smooth, the V-cycle                MultigridPreconditioner.h:582-605,266-318,190-226,362-421
  * LBFGS::solve, lineSearch, shouldExitByCN                     Lib/Ziran/Math/Nonlinear/LBFGS.h:300-437, ImplicitSolver.h:312-333,174-211,667-696

Node NUMBERING is an input (`id2coord`, the reference's g.idx order, which the SPGrid golden vectors pin separately): everything here is
computed in whatever numbering the library under test reports, so that numbering-dependent results (the GS order, hence the V-cycle and
the L-BFGS iterates) can be compared entry by entry.  make_step_golden.py stores the numbering-free part as vectors keyed by coordinates."""
import numpy as np

from make_fp_golden import corotated, bspline  # numpy-only siblings (psi, P, dP/dF; quadratic B-spline weights)


def dbspline(x):
    """derivatives of the three quadratic B-spline weights with respect to the index-space coordinate"""
    base = np.floor(x - 0.5)
    d0 = x - base
    return np.stack([-(1.5 - d0), -2 * (d0 - 1), d0 - 0.5], -1)


class Cloud:
    def __init__(self, X, V, C, mass, vol, F, mu, lam, dx, dt, gravity, floor_y):
        self.X, self.V, self.C, self.mass, self.vol, self.F, self.mu, self.lam = X, V, C, mass, vol, F, mu, lam
        self.dx, self.dt, self.g, self.floor_y = dx, dt, np.asarray(gravity, float), floor_y
        self.base, self.w = bspline(X / dx)
        self.dw = dbspline(X / dx) / dx


def touched_nodes(c):
    """every node a particle's 3^3 kernel reaches, as a set of integer coordinates"""
    s = set()
    for p in range(len(c.X)):
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    s.add((int(c.base[p, 0]) + i, int(c.base[p, 1]) + j, int(c.base[p, 2]) + k))
    return s


class Step:
    """all level-0 quantities in the numbering `id2coord` ((n,3) integer node coordinates of the DOFs, i.e. the nodes of non-zero mass)"""

    def __init__(self, c, id2coord):
        self.c = c
        self.coord = [tuple(int(v) for v in r) for r in np.asarray(id2coord)]
        self.n = n = len(self.coord)
        self.idx = {k: i for i, k in enumerate(self.coord)}
        m, mv = np.zeros(n), np.zeros((n, 3))
        self.kern = []  # per particle: [(dof, w, grad w)]
        for p in range(len(c.X)):
            lst = []
            for i in range(3):
                for j in range(3):
                    for k in range(3):
                        node = (int(c.base[p, 0]) + i, int(c.base[p, 1]) + j, int(c.base[p, 2]) + k)
                        w = c.w[p, 0, i] * c.w[p, 1, j] * c.w[p, 2, k]
                        gw = np.array([c.dw[p, 0, i] * c.w[p, 1, j] * c.w[p, 2, k], c.w[p, 0, i] * c.dw[p, 1, j] * c.w[p, 2, k], c.w[p, 0, i] * c.w[p, 1, j] * c.dw[p, 2, k]])
                        if node not in self.idx:
                            assert w * c.mass[p] == 0.0, "a node with mass is missing from id2coord"
                            continue
                        d = self.idx[node]
                        xi = np.array(node) * c.dx
                        m[d] += w * c.mass[p]
                        mv[d] += w * c.mass[p] * (c.V[p] + c.C[p] @ (xi - c.X[p]))
                        lst.append((d, w, gw))
            self.kern.append(lst)
        self.m, self.vn = m, mv / m[:, None]
        # sticky half space {x : y <= floor_y}: collision nodes keep dv = -v (static collider), everything else starts from g dt
        self.bc = np.array([k[1] * c.dx - c.floor_y <= 0 for k in self.coord])
        self.dv0 = np.where(self.bc[:, None], -self.vn, c.g[None, :] * c.dt)

    # ---- objective
    def trial_F(self, dv):
        v = self.vn + dv
        out = np.empty_like(self.c.F)
        for p, lst in enumerate(self.kern):
            grad = np.zeros((3, 3))
            for d, w, gw in lst:
                grad += np.outer(v[d], gw)
            out[p] = (np.eye(3) + self.c.dt * grad) @ self.c.F[p]
        return out

    def energy(self, dv):
        c = self.c
        psi, _, _ = corotated(self.trial_F(dv), c.mu, c.lam, False)
        return float((c.vol * psi).sum() + 0.5 * (self.m * (dv ** 2).sum(1)).sum() - c.dt * (self.m * (dv @ c.g)).sum())

    def project(self, v):
        v = v.copy()
        v[self.bc] = 0.0
        return v

    def residual(self, dv):
        c = self.c
        _, P, _ = corotated(self.trial_F(dv), c.mu, c.lam, False)
        r = c.g[None, :] * c.dt * self.m[:, None] - dv * self.m[:, None]
        for p, lst in enumerate(self.kern):
            stress = c.vol[p] * P[p] @ c.F[p].T
            for d, w, gw in lst:
                r[d] -= c.dt * stress @ gw
        return self.project(r)

    def hessian(self, dv, project_psd=True):
        """dense 3n x 3n, xyz interleaved per node; sticky rows / columns replaced by the identity"""
        c = self.c
        _, _, D9 = corotated(self.trial_F(dv), c.mu, c.lam, project_psd)  # D9[p, i + 3 j, r + 3 s] = dP_ij / dF_rs
        n = self.n
        H = np.zeros((3 * n, 3 * n))
        for d in range(n):
            H[3 * d:3 * d + 3, 3 * d:3 * d + 3] = self.m[d] * np.eye(3)
        for p, lst in enumerate(self.kern):
            D = D9[p].reshape(3, 3, 3, 3)  # [j, i, s, r]
            FnT = c.F[p].T
            ws = [(d, FnT @ gw) for d, w, gw in lst]
            for da, wa in ws:
                for db, wb in ws:
                    # block(a, b)_{i r} = dt^2 V_p sum_{j s} wa_j dP_ij/dF_rs wb_s
                    blk = np.einsum("j,jisr,s->ir", wa, D, wb)
                    H[3 * da:3 * da + 3, 3 * db:3 * db + 3] += c.dt ** 2 * c.vol[p] * blk
        for d in np.nonzero(self.bc)[0]:
            H[3 * d:3 * d + 3, :] = 0.0
            H[:, 3 * d:3 * d + 3] = 0.0
            H[3 * d:3 * d + 3, 3 * d:3 * d + 3] = np.eye(3)
        return H

    def cn_tolerance(self, cneps):
        """per-node tolerance of the characteristic-norm exit test: |dP/dF(I)|_F mass-weighted onto the nodes"""
        c = self.c
        tol = np.zeros(self.n)
        for p, lst in enumerate(self.kern):
            _, _, D9 = corotated(np.eye(3)[None], c.mu, c.lam, True)
            nrm = np.sqrt((D9[0] ** 2).sum())
            for d, w, gw in lst:
                tol[d] += w * c.mass[p] * nrm
        return tol * (cneps * 24 * c.dx * c.dx * c.dt) / self.m


# ---------------------------------------------------------------------------------------------------- hierarchy
def coarsen(coord):
    """trilinear prolongation to the next level: coarse nodes numbered by first touch while the fine nodes are walked in id order and the
    2^3 parents in (x, y, z) order; returns (coarse coords, dense P (n x nc))"""
    w1d = [[0.0, 1.0, 0.0], [0.0, 0.5, 0.5]]
    new, ids = [], {}
    rows = []
    for (x, y, z) in coord:
        row = {}
        for nx in (x // 2, x // 2 + 1):
            for ny in (y // 2, y // 2 + 1):
                for nz in (z // 2, z // 2 + 1):
                    wt = w1d[x & 1][nx - x // 2 + 1] * w1d[y & 1][ny - y // 2 + 1] * w1d[z & 1][nz - z // 2 + 1]
                    if wt == 0:
                        continue
                    k = (nx, ny, nz)
                    if k not in ids:
                        ids[k] = len(new)
                        new.append(k)
                    row[ids[k]] = wt
        rows.append(row)
    P = np.zeros((len(coord), len(new)))
    for i, row in enumerate(rows):
        for j, wt in row.items():
            P[i, j] = wt
    return new, P


def expand3(P):
    return np.kron(P, np.eye(3))


def gs_order(coord):
    """(colour, block, index) key of every node: 4^3 blocks, colour = parity bits of the block coordinates, blocks numbered per colour in
    first-touch order, nodes inside a block in id order"""
    cnt = [0] * 8
    bid = [dict() for _ in range(8)]
    fill = {}
    key = []
    for (x, y, z) in coord:
        b = (x >> 2, y >> 2, z >> 2)
        col = ((b[0] & 1) << 2) | ((b[1] & 1) << 1) | (b[2] & 1)
        if b not in bid[col]:
            bid[col][b] = cnt[col]
            cnt[col] += 1
        k = (col, bid[col][b])
        fill[k] = fill.get(k, 0) + 1
        key.append((col, bid[col][b], fill[k]))
    return key


def blk(A, i, j):
    return A[3 * i:3 * i + 3, 3 * j:3 * j + 3]


def gs_smooth(A, coord, u, r, iterations):
    """symmetric coloured block Gauss-Seidel, `iterations` counted like the reference (two per symmetric sweep); returns (u, r)"""
    n = len(coord)
    key = gs_order(coord)
    order = sorted(range(n), key=lambda i: key[i])
    Dinv = [np.linalg.inv(blk(A, i, i)) for i in range(n)]
    u, r = u.copy(), r.copy()
    for _ in range((iterations + 1) >> 1):
        h = np.zeros((n, 3))
        for i in order:
            s = np.zeros(3)
            for j in range(n):
                if key[j] < key[i]:
                    s += blk(A, i, j) @ h[j]
            h[i] = Dinv[i] @ (r[i] - s)
        hD = np.array([blk(A, i, i) @ h[i] for i in range(n)])
        du = np.zeros((n, 3))
        for i in reversed(order):
            s = np.zeros(3)
            for j in range(n):
                if key[j] > key[i]:
                    s += blk(A, i, j) @ du[j]
            du[i] = Dinv[i] @ (hD[i] - s)
        u += du
        r -= (A @ du.reshape(-1)).reshape(n, 3)
    return u, r


def cg_smooth(A, u, r, r_init, iterations):
    """preconditioned CG with the 3x3 block-diagonal inverse, stopping at z.r < (0.5)^2 z0.r0 of the reference residual"""
    n = A.shape[0] // 3
    Dinv = [np.linalg.inv(blk(A, i, i)) for i in range(n)]
    scale = lambda v: np.array([Dinv[i] @ v[i] for i in range(n)])
    u, r = u.copy(), r.copy()
    z = scale(r_init)
    tol = (z * r_init).sum() * 0.25
    z = scale(r)
    du = z.copy()
    ztr = (z * r).sum()
    its = 0
    while iterations > 0 and not ztr < tol:
        iterations -= 1
        dAu = (A @ du.reshape(-1)).reshape(n, 3)
        om = ztr / (dAu * du).sum()
        u += om * du
        r -= om * dAu
        z = scale(r)
        pre, ztr = ztr, (z * r).sum()
        du = z + du * (ztr / pre)
        its += 1
    return u, r, its


class Hierarchy:
    """two levels: symmetric GS on level 0 (one symmetric sweep down, one up), PCG on top (the tog.sh command set with -mg_level 2)"""

    def __init__(self, A0, coord0):
        self.A0, self.coord0 = A0, coord0
        self.coord1, self.P = coarsen(coord0)
        P3 = expand3(self.P)
        self.A1 = P3.T @ A0 @ P3
        self.linear_iterations = 0

    def vcycle(self, x):
        n0, n1 = len(self.coord0), len(self.coord1)
        r0 = x.copy()
        out = np.zeros((n0, 3))
        init1 = self.P.T @ r0
        out, r0 = gs_smooth(self.A0, self.coord0, out, r0, 1)
        r1 = self.P.T @ r0
        sol1, r1, its = cg_smooth(self.A1, np.zeros((n1, 3)), r1, init1, 10000)
        self.linear_iterations += its
        du = self.P @ sol1
        out = out + du
        r0 = r0 - (self.A0 @ du.reshape(-1)).reshape(n0, 3)
        out, r0 = gs_smooth(self.A0, self.coord0, out, r0, 1)
        return out


# ---------------------------------------------------------------------------------------------------- L-BFGS
def lbfgs(step, iterations, cneps=1e-7, history=8):
    """LBFGS::solve with the V-cycle of the Hessian at the first iterate as initial inverse Hessian, energy line search; returns
    (dv after `iterations` iterations, line-search trials, hierarchy).  With --linesearch the reference adds the accepted step to an x
    that the line search has already moved: reproduced."""
    n = step.n
    tol = step.cn_tolerance(cneps)
    x = step.dv0.copy()
    Ek = step.energy(x)
    residual = step.residual(x)
    hist = [dict()]
    trials = 0
    mg = None
    dv0 = step.dv0.copy()
    for it in range(iterations):
        if ((residual ** 2).sum(1) / tol ** 2).sum() < n:
            break
        if it == 0:
            mg = Hierarchy(step.hessian(x), step.coord)
            hist = [dict()]
        hist[-1]["dg"] = residual.copy()
        ksi = {}
        q = residual.copy()
        for i in range(len(hist) - 2, -1, -1):
            ksi[i] = (hist[i]["dx"] * q).sum() * hist[i]["rho"]
            q -= hist[i]["dg"] * ksi[i]
        d = step.project(mg.vcycle(q))
        for i in range(len(hist) - 1):
            cc = ksi[i] - (hist[i]["dg"] * d).sum() * hist[i]["rho"]
            d += hist[i]["dx"] * cc
        # line search from alpha = 1 on the energy
        alpha, Ek0 = 1.0, Ek
        while True:
            dvnew = dv0 + alpha * d
            Ek = step.energy(dvnew)
            trials += 1
            alpha *= 0.5
            if Ek <= Ek0:
                break
        alpha *= 2
        d = d * alpha
        residual = step.residual(dvnew)
        dv0 = dvnew
        x = dvnew + d  # x was moved to the accepted point by the line search, then `x += step` (the reference's aliasing)
        hist[-1]["dx"] = d
        hist[-1]["dg"] = hist[-1]["dg"] - residual
        rho = 1.0 / (hist[-1]["dg"] * d).sum()
        if rho <= 0:
            hist.pop()
        else:
            hist[-1]["rho"] = rho
        hist.append(dict())
        if len(hist) > history + 1:
            hist.pop(0)
    return x, trials, mg


def tiny_cloud(seed=20260929):
    """~40 particles around the corner (500, 500, 500) dx of eight 4^3 colour blocks, so that all eight colours and a coarse level exist;
    deformed Fn, per-particle velocities and APIC matrices; a sticky floor under the lowest node layers"""
    rng = np.random.default_rng(seed)
    dx, dt = 0.01, 1.0 / 24
    n = 40
    X = 5.0 + dx * (-1.2 + 2.4 * rng.random((n, 3)))
    V = 0.5 * rng.standard_normal((n, 3))
    C = 2.0 * rng.standard_normal((n, 3, 3))
    F = np.eye(3)[None] + 0.05 * rng.standard_normal((n, 3, 3))
    mass = 2000.0 * dx ** 3 / 8 * (0.5 + rng.random(n))
    vol = np.full(n, dx ** 3 / 8)
    E, nu = 5e4, 0.3
    mu, lam = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    return Cloud(X, V, C, mass, vol, F, mu, lam, dx, dt, (0.0, -9.8, 0.0), 5.0 - 1.5 * dx)
