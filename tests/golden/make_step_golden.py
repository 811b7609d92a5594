#!/usr/bin/env python3
"""Generator of tests/golden/step_golden.npz: the numbering-free part of one whole tiny time step (tests/golden/np_step.py, numpy only),
stored as vectors keyed by integer node coordinates in lexicographic order: node masses and velocities, the start iterate dv0, the
incremental potential and the projected residual there, the assembled Hessian with its boundary projection (dense, 3 x 3 blocks by
coordinate pair), the trilinear prolongation and the Galerkin coarse matrix in the coarse nodes' lexicographic order.  A library under
test reports its own node numbering; tests/golden_checks.check_step maps it onto these coordinates and compares.  The numbering-dependent
results (coloured Gauss-Seidel sweep, V-cycle, L-BFGS iterates) are computed by np_step.py at test time in the library's numbering, and
their values IN LEXICOGRAPHIC NUMBERING are stored here as well, as a regression pin of the numpy code itself.

Run:  python tests/golden/make_step_golden.py      (deterministic; rewrites step_golden.npz)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import np_step as ns  # noqa: E402


def main():
    c = ns.tiny_cloud()
    # the degrees of freedom: touched nodes of non-zero mass, lexicographic
    cand = sorted(ns.touched_nodes(c))
    probe = ns.Step.__new__(ns.Step)
    mass = {}
    for p in range(len(c.X)):
        for i in range(3):
            for j in range(3):
                for k in range(3):
                    node = (int(c.base[p, 0]) + i, int(c.base[p, 1]) + j, int(c.base[p, 2]) + k)
                    mass[node] = mass.get(node, 0.0) + c.w[p, 0, i] * c.w[p, 1, j] * c.w[p, 2, k] * c.mass[p]
    coord = np.array([k for k in cand if mass[k] != 0.0], np.int32)
    st = ns.Step(c, coord)
    H = st.hessian(st.dv0)
    assert np.abs(H - H.T).max() < 1e-12 * np.abs(H).max() and np.linalg.eigvalsh(H).min() > 0
    # the residual is minus the gradient of the energy, the Hessian its derivative (centred differences, free nodes)
    rng = np.random.default_rng(3)
    d = st.project(rng.standard_normal((st.n, 3)))
    h = 1e-6
    g = (st.energy(st.dv0 + h * d) - st.energy(st.dv0 - h * d)) / (2 * h)
    r0 = st.residual(st.dv0)
    assert abs(g + (r0 * d).sum()) < 1e-6 * abs(g), (g, (r0 * d).sum())
    Hun = st.hessian(st.dv0, project_psd=False)
    dr = (st.residual(st.dv0 + h * d) - st.residual(st.dv0 - h * d)) / (2 * h)
    free = ~st.bc
    assert np.abs(-dr[free] - (Hun @ d.reshape(-1)).reshape(-1, 3)[free]).max() < 1e-5 * np.abs(dr).max()
    coord1, P = ns.coarsen(st.coord)
    order1 = sorted(range(len(coord1)), key=lambda i: coord1[i])
    P3 = ns.expand3(P)
    A1 = P3.T @ H @ P3
    perm3 = np.array([3 * i + k for i in order1 for k in range(3)])
    b = st.project(np.random.default_rng(7).standard_normal((st.n, 3)))
    gu, gr = ns.gs_smooth(H, st.coord, np.zeros((st.n, 3)), b, 2)
    mg = ns.Hierarchy(H, st.coord)
    vc = mg.vcycle(b)
    x2, trials, _ = ns.lbfgs(st, 2)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_golden.npz")
    np.savez_compressed(out, coord=coord, mass=st.m, v=st.vn, dv0=st.dv0, bc=st.bc, energy=st.energy(st.dv0), residual=r0, hessian=H.astype(np.float64),
                        coord1=np.array([coord1[i] for i in order1], np.int32), P=P[:, order1], A1=A1[np.ix_(perm3, perm3)],
                        lex_rhs=b, lex_gs_u=gu, lex_gs_r=gr, lex_vcycle=vc, lex_lbfgs_dv=x2, lex_linesearch_trials=trials)
    print("wrote", out, os.path.getsize(out), "bytes;", st.n, "nodes,", len(coord1), "coarse nodes,", int(st.bc.sum()), "collision nodes, colours",
          sorted({k[0] for k in ns.gs_order(st.coord)}))


if __name__ == "__main__":
    main()
