"""CPU: the oracle passes the reference's own self-check ideas (so it can be trusted as the checker)."""
import numpy as np
import pytest

from tests import pipeline_checks as pc


def test_transfer_conservation(oracle):
    pc.check_transfer_conservation(oracle, 1, 1e-12)
    pc.check_transfer_conservation(oracle, 0, 2e-5)


def test_apic_affine(oracle):
    pc.check_apic_affine_reproduction(oracle, 1, 1e-10)


def test_diff_test(oracle):
    err = pc.check_diff_test(oracle)
    assert err < 1e-5


@pytest.mark.parametrize("project", [0, 1])
def test_matrix_equals_matrix_free(oracle, project):
    err, sym, pd = pc.check_matrix_vs_matfree(oracle, project=project)
    assert err < 1e-10 and sym < 1e-10 and pd > 0


def test_galerkin_hierarchy(oracle):
    pc.check_galerkin(oracle)


def test_vcycle_symmetric_pd(oracle):
    sym, energies = pc.check_vcycle_spd(oracle)
    assert sym < 1e-8
    assert all(b < a for a, b in zip(energies, energies[1:]))


def test_solvers_converge_and_agree(oracle):
    res = {}
    for name, kw in {"lbfgs": dict(lsolver=3, levelCnt=2), "pn": dict(lsolver=2, levelCnt=2), "pn_jacobi": dict(lsolver=2, levelCnt=1, coarseSolver=0, smoother=0),
                     "pn_minres": dict(lsolver=1, levelCnt=2, coarseSolver=5), "pn_minres_mass": dict(lsolver=1, levelCnt=1, Ainv=2),
                     "pn_matfree": dict(lsolver=2, levelCnt=1, matrixFree=1, systemBCProject=0)}.items():
        ctx, c = pc.make_ctx(oracle, n=5, cneps=1e-8, **kw)
        pc.prepare(ctx)
        st = ctx.solve()
        assert st["converged"] == 1, (name, st)
        assert st["final_scaled_residual"] < 1.0
        # the state the solver stopped at (dv0) has a small residual: re-evaluate there
        res[name] = (ctx.get_dv(), st, ctx.grid()["mass"])
    # all three minimise the same energy.  The returned dv is only loosely comparable: low-mass corner nodes
    # are poorly determined at the CN tolerance and the reference applies the last accepted step twice
    # (LBFGS.h:412-413 after lineSearch's moveNodes aliasing, see DESIGN.md "reference quirks").
    for other in ("pn", "pn_jacobi", "pn_minres", "pn_minres_mass", "pn_matfree"):
        a, b = res["lbfgs"][0], res[other][0]
        m = res["lbfgs"][2][:, None]
        assert np.sqrt((m * (a - b) ** 2).sum()) < 2e-2 * np.sqrt((m * a ** 2).sum())  # kinetic-energy norm
        ea, eb = res["lbfgs"][1]["energy"], res[other][1]["energy"]
        assert abs(ea - eb) < 1e-6 * max(abs(ea), 1e-3)
