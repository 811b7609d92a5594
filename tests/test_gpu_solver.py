"""GPU parity: assembled Hessian, Galerkin hierarchy, smoothers, V-cycle, L-BFGS / PN solves and whole time steps
through the C ABI against the CPU oracle."""
import os

import numpy as np
import pytest

from tests import pipeline_checks as pc

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def built(lib, n=8, dtype=1, **kw):
    ctx, c = pc.make_ctx(lib, n=n, dtype=dtype, **kw)
    pc.prepare(ctx)
    ctx.update_state(ctx.get_dv())
    ctx.build_hessian()
    ctx.build_mg()
    return ctx


def level_matrices(ctx, nlev):
    out = []
    for l in range(nlev):
        col, val = ctx.matrix(l)
        out.append(pc.ell_to_scipy(col, val, col.shape[0]))
    return out


@pytest.mark.parametrize("dtype,tol", [(1, 1e-10), (0, 5e-4)])
def test_hessian_and_hierarchy_against_oracle(hotlib, oracle, dtype, tol):
    g = built(hotlib, dtype=dtype, levelCnt=3)
    c = built(oracle, dtype=dtype, levelCnt=3)
    mg, mc = level_matrices(g, 3), level_matrices(c, 3)
    for l in range(3):
        assert np.array_equal(g.level(l)["id2coord"], c.level(l)["id2coord"])  # coarse numbering is bit-exact
        d = abs(mg[l] - mc[l]).max()
        assert d < tol * abs(mc[l]).max(), (l, d)
        assert abs(mg[l] - mg[l].T).max() < tol * abs(mc[l]).max()
    for l in range(2):
        gp, cp = g.prolongation(l), c.prolongation(l)
        assert np.array_equal(gp[0], cp[0]) and np.array_equal(gp[1], cp[1])


def test_matrix_vs_matfree_and_galerkin_on_gpu(hotlib):
    err, sym, pd = pc.check_matrix_vs_matfree(hotlib, project=1)
    assert err < 1e-10 and sym < 1e-10 and pd > 0
    err, sym, pd = pc.check_matrix_vs_matfree(hotlib, project=0)
    assert err < 1e-10
    pc.check_galerkin(hotlib)
    sym, energies = pc.check_vcycle_spd(hotlib)
    assert sym < 1e-8 and all(b < a for a, b in zip(energies, energies[1:]))


@pytest.mark.parametrize("kind", [0, 1, 2, 5])
@pytest.mark.parametrize("level", [0, 1])
def test_smoothers_against_oracle(hotlib, oracle, kind, level):
    g, c = built(hotlib, levelCnt=2), built(oracle, levelCnt=2)
    n = g.level(level, coords=False)["nrows"]
    rng = np.random.default_rng(3)
    b = rng.standard_normal((n, 3))
    if level == 0:
        b = c.project(b)
    its = 6 if kind != 5 else 4
    ug, rg = g.smooth(level, kind, its, np.zeros_like(b), b, tolerance=0.0)
    uc, rc = c.smooth(level, kind, its, np.zeros_like(b), b, tolerance=0.0)
    assert rel(ug, uc) < 1e-9, rel(ug, uc)
    assert rel(rg, rc) < 1e-8


@pytest.mark.parametrize("level", [0, 1])
def test_chebyshev_smoother_against_oracle(hotlib, oracle, level):
    """smoother 6: the spectrum bound comes from a power iteration stopped at a 1e-6 relative change (estimate2norm),
    so the two implementations share lMax to ~1e-6 and the smoothed iterate to about that level."""
    cfg = dict(levelCnt=3, smoother=6, coarseSolver=2)
    g, c = built(hotlib, **cfg), built(oracle, **cfg)
    n = g.level(level, coords=False)["nrows"]
    b = np.random.default_rng(3).standard_normal((n, 3))
    if level == 0:
        b = c.project(b)
    ug, rg = g.smooth(level, 6, 6, np.zeros_like(b), b, tolerance=0.0)
    uc, rc = c.smooth(level, 6, 6, np.zeros_like(b), b, tolerance=0.0)
    assert rel(ug, uc) < 1e-5, rel(ug, uc)
    assert rel(rg, rc) < 1e-5
    # (no descent check: the reference bounds the spectrum of A but iterates on D^-1 A, MultigridPreconditioner.h:231-239,
    # so whether this smoother contracts depends on the scene's units; only parity is asserted)
    x = c.project(np.random.default_rng(5).standard_normal((c.Nn, 3)))
    assert rel(g.vcycle(x), c.vcycle(x)) < 1e-5


@pytest.mark.parametrize("cfg", [dict(levelCnt=3, coarseSolver=2), dict(levelCnt=2, coarseSolver=5), dict(levelCnt=1, coarseSolver=2), dict(levelCnt=3, smoother=0, coarseSolver=2)])
def test_vcycle_against_oracle(hotlib, oracle, cfg):
    g, c = built(hotlib, **cfg), built(oracle, **cfg)
    x = c.project(np.random.default_rng(5).standard_normal((c.Nn, 3)))
    assert rel(g.vcycle(x), c.vcycle(x)) < 1e-8


SOLVER_CFGS = [dict(lsolver=3, levelCnt=3), dict(lsolver=3, levelCnt=1), dict(lsolver=2, levelCnt=2), dict(lsolver=2, levelCnt=1, smoother=0, coarseSolver=0),
               dict(lsolver=1, levelCnt=2, coarseSolver=5),  # projected Newton + MINRES, V-cycle preconditioner (GS on every level: MINRES needs a fixed SPD preconditioner, the PCG coarse solve is not one)
               dict(lsolver=1, levelCnt=1, Ainv=2),  # ... lumped-mass preconditioner (no hierarchy)
               dict(lsolver=2, levelCnt=1, matrixFree=1, systemBCProject=0),  # matrix-free PN with the block-diagonal preconditioner
               dict(lsolver=2, levelCnt=1, matrixFree=1, systemBCProject=0, Ainv=0)]


@pytest.mark.parametrize("kw", SOLVER_CFGS)
def test_solver_iterates_against_oracle(hotlib, oracle, kw):
    """Tight parity: after a fixed small number of nonlinear iterations the iterates agree to round-off."""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, cneps=1e-7, max_iterations=5, **kw)
        pc.prepare(ctx)
        st = ctx.solve()
        out[name] = (ctx.get_dv(), st)
    sg, sc = out["gpu"][1], out["cpu"][1]
    if kw.get("Ainv") == 2:
        # MINRES behind the lumped-mass preconditioner needs hundreds of Lanczos steps per Newton iteration; its
        # loss of orthogonality makes the stopping iteration (not the solution) sensitive to round-off
        assert sg["iterations"] == sc["iterations"] and sg["num_levels"] == sc["num_levels"]
        assert abs(sg["linear_iterations"] - sc["linear_iterations"]) <= 0.05 * sc["linear_iterations"], (sg, sc)
        assert rel(out["gpu"][0], out["cpu"][0]) < 1e-2
        assert abs(sg["energy"] - sc["energy"]) < 1e-3 * max(abs(sc["energy"]), 1e-6)
        return
    for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles", "dropped_pairs", "num_levels"):
        assert sg[k] == sc[k], (k, sg, sc)
    assert rel(out["gpu"][0], out["cpu"][0]) < 1e-9
    assert abs(sg["energy"] - sc["energy"]) < 1e-10 * max(abs(sc["energy"]), 1e-6)
    assert abs(sg["final_scaled_residual"] - sc["final_scaled_residual"]) < 1e-7 * sc["final_scaled_residual"]


@pytest.mark.parametrize("kw", [dict(lsolver=3, levelCnt=3), dict(lsolver=2, levelCnt=2)], ids=["lbfgs_mg3", "pn_mgpcg2"])
def test_solver_iterates_through_the_independent_mirror(kw):
    """The fixed-iteration parity once more with BOTH libraries driven through tests/golden_checks.py's own ctypes mirror of the C ABI (its own
    hot_config / hot_stats layouts, its own argument marshalling) instead of hot_amd/binding.py: a mistake of the shared binding — an argument
    in the wrong order, a field at the wrong offset — would reach the HIP library and the oracle alike and cancel in every other test here."""
    import hot_amd
    from tests import golden_checks, oracle_lib
    oracle_lib.load_oracle()  # (builds oracle/liboracle.so if needed)
    dg, cg, eg = golden_checks.fixed_iterations_raw(hot_amd.LIB_PATH, "hot_", **kw)
    dc, cc, ec = golden_checks.fixed_iterations_raw(os.path.join(oracle_lib.ORACLE_DIR, "liboracle.so"), "hoto_", **kw)
    assert cg == cc and cg["iterations"] == 5, (cg, cc)
    assert rel(dg, dc) < 1e-9, rel(dg, dc)
    assert abs(eg - ec) < 1e-10 * max(abs(ec), 1e-6)


@pytest.mark.parametrize("Ainv", [2, 1])
@pytest.mark.parametrize("cap,tol", [(5, 1e-11), (10, 1e-10), (20, 1e-9)])
def test_minres_at_fixed_lanczos_count(hotlib, oracle, Ainv, cap, tol):
    """Projected Newton + MINRES (Minres.h:69-178) behind the lumped-mass (Ainv = 2) and the block-diagonal preconditioner: a Newton iteration
    takes hundreds of Lanczos steps, and where the stopping test fires depends on round-off (the iterate test above allows 5 % on the count
    and 1e-2 on the result for that reason).  Stopped at the SAME Lanczos step (hot_config.linear_iteration_cap) the two implementations run
    the same three-term recurrences on the same operator, and the Newton step itself is compared.  Measured separation of the two steps
    (tools/minres_growth.py, n = 8, fp64): 2e-15 after 1 step, 2e-14 after 10, 4e-12 after 20 — then the Lanczos vectors lose orthogonality
    (the first Ritz value has converged) and the runs are two different Krylov processes: 5e-3 after 40 steps, 8e-4 after 80, 8e-5 after 150,
    both on their way to the same solution.  The tight comparison is therefore made where the recurrence is still the same computation."""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, cneps=1e-7, max_iterations=1, lsolver=1, levelCnt=1, Ainv=Ainv, linear_iteration_cap=cap)
        pc.prepare(ctx)
        st = ctx.solve()
        out[name] = (ctx.get_dv(), st)
    sg, sc = out["gpu"][1], out["cpu"][1]
    assert sg["linear_iterations"] == sc["linear_iterations"] == cap, (sg, sc)
    assert sg["linesearch_trials"] == sc["linesearch_trials"]
    assert rel(out["gpu"][0], out["cpu"][0]) < tol, rel(out["gpu"][0], out["cpu"][0])
    assert abs(sg["energy"] - sc["energy"]) < 1e-10 * max(abs(sc["energy"]), 1e-6)


@pytest.mark.parametrize("kw", SOLVER_CFGS)
def test_solve_to_convergence_against_oracle(hotlib, oracle, kw):
    """Converged solves: same minimum (energy), iteration counts within a few percent, dv within solver tolerance.
    (Round-off differences are amplified by line-search accept/reject decisions over many iterations, so the
    converged dv is only comparable at the level the termination test controls.)"""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, cneps=1e-7, **kw)
        pc.prepare(ctx)
        st = ctx.solve()
        out[name] = (ctx.get_dv(), st, ctx.grid()["mass"])
    sg, sc = out["gpu"][1], out["cpu"][1]
    assert sg["converged"] == 1 and sc["converged"] == 1
    assert abs(sg["iterations"] - sc["iterations"]) <= max(2, sc["iterations"] // 10), (sg, sc)
    m = out["cpu"][2][:, None]
    a, b = out["gpu"][0], out["cpu"][0]
    err = np.sqrt((m * (a - b) ** 2).sum()) / np.sqrt((m * b ** 2).sum())
    same = all(sg[k] == sc[k] for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles", "dropped_pairs"))
    print("converged solve %s: mass-weighted |ddv| / |dv| = %.3g, counters %s (%d / %d iterations)" % (kw, err, "equal" if same else "differ", sg["iterations"], sc["iterations"]))
    # equal counters: the two runs took the same discrete decisions and differ by amplified round-off only; different counters: both stopped
    # by the same test, at different points of the same convergence history
    # (beyond a hundred iterations the amplification outgrows 1e-3 whatever the counters do: the single-level L-BFGS solve, 208 iterations, gives
    # 0.9e-3 .. 1.8e-3 from run to run and from build to build — tools/conv_err.py, the round-4 library included —, with the counters equal or one apart)
    assert err < ((1e-3 if sc["iterations"] <= 100 else 3e-3) if same else 1e-2), (err, same)
    assert abs(sg["energy"] - sc["energy"]) < 1e-4 * max(abs(sc["energy"]), 1e-6)


@pytest.mark.parametrize("dtype,cneps,E", [(1, 1e-8, 5e4), (1, 1e-7, 2e6), (0, 1e-5, 5e4)], ids=["fp64", "fp64_stiff", "fp32_tight"])
def test_line_search_decisions_pinned_against_oracle(hotlib, oracle, dtype, cneps, E):
    """The accept / reject decisions `Ek <= Ek0` of lineSearch (ImplicitSolver.h:312-333), end to end.  Since round 5 the device evaluates the trial
    energies from the invariants of F^T F (hot_constitutive.h corotated_psi_invariants) where the reference sums mu |F - R|^2 + lambda / 2 (J - 1)^2
    through the SVD; a search that accepted one trial later would still converge in a similar number of iterations, so the iteration-count bound of
    the converged-solve tests does not see it.  Here: the same solve cut off after 1, 2, 3, 5, 8, 13, 21, ... iterations and at convergence, on the
    device with the trials as full state passes (ls_energy_only = 1), as energy-only passes (2) and adaptive (0), on the oracle in the reference's
    form, and on the oracle with EVERY energy in the product's form (tests/oracle_lib.py psi_invariants): wherever two runs made the same number of
    iterations they must have evaluated the same number of trials — at every cut, i.e. search by search up to the point where round-off first
    moves a decision, which must not happen before convergence in fp64."""
    from tests.oracle_lib import psi_invariants
    kw = dict(lsolver=3, levelCnt=2, cneps=cneps)
    cuts = [1, 2, 3, 5, 8, 13, 21, 34, 55, 10000]

    def history(lib, **over):
        out = []
        for k in cuts:
            ctx, c = pc.make_ctx(lib, n=8, dtype=dtype, E=E, max_iterations=k, **dict(kw, **over))
            pc.prepare(ctx)
            st = ctx.solve()
            out.append((st["iterations"], st["linesearch_trials"], st["converged"]))
            if st["converged"] == 1 and st["iterations"] < k:
                break
        return out

    runs = {"oracle": history(oracle)}
    with psi_invariants():
        runs["oracle, product form"] = history(oracle)
    for mode in (1, 2, 0):
        runs["device ls_energy_only=%d" % mode] = history(hotlib, ls_energy_only=mode)
    ref = runs["oracle"]
    assert ref[-1][2] == 1, ref
    halvings = ref[-1][1] - ref[-1][0]  # trials beyond the first of every search
    print("line-search pin (%s): iterations / trials at the cuts: %s" % ("fp64" if dtype else "fp32", {k: [(a, b) for a, b, _ in v] for k, v in runs.items()}))
    assert halvings >= 3, ("the case must exercise rejections", ref)
    for name, h in runs.items():
        assert h[-1][2] == 1, (name, h)
        for (i0, t0, _), (i1, t1, _) in zip(ref, h):
            if i0 == i1:
                assert t0 == t1, (name, ref, h)
        if dtype == 1:  # fp64: not a single decision moves before convergence; the converged counts agree to within the last search
            assert all(a[0] == b[0] for a, b in zip(ref[:-1], h[:-1])), (name, ref, h)
            assert abs(h[-1][0] - ref[-1][0]) <= 1, (name, ref, h)


def test_three_time_steps_against_oracle(hotlib, oracle):
    """Whole steps (sort -> P2G -> solve -> G2P) chained three times, fp64.  Each step ends at the solver's termination
    tolerance, so velocities agree to that level (relative to the initial velocity scale), positions much tighter."""
    out = {}
    v0 = None
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, dtype=1, levelCnt=2, cneps=1e-6, max_iterations=300)
        v0 = np.abs(c["V"]).max()
        its = []
        for _ in range(3):
            its.append(ctx.advance(1.0 / 24)["iterations"])
        out[name] = (ctx.get_particles(), its)
    pg, pcpu = out["gpu"][0], out["cpu"][0]
    assert rel(pg["X"], pcpu["X"]) < 1e-8, (rel(pg["X"], pcpu["X"]), out["gpu"][1], out["cpu"][1])
    assert np.abs(pg["V"].astype(np.float64) - pcpu["V"]).max() < 1e-4 * v0
    assert np.abs(pg["F"].astype(np.float64) - pcpu["F"]).max() < 1e-4


def test_three_time_steps_fp32_against_fp32_oracle(hotlib, oracle):
    """fp32 whole steps against the oracle's own float arithmetic, at the BASELINE time step dt = 1/24.  The level-0 system has
    cond ~ 1e8 ~ 1 / eps_float (low-mass boundary nodes), so float trajectories of two correct implementations drift apart chaotically as
    soon as ONE discrete decision differs (a line-search halving, one more top-level PCG iteration), and cannot be compared point-wise
    after that.  What is comparable: a bounded number of iterations from one and the same state, for as long as the discrete decisions
    agree.  At the start of each of three consecutive time steps (the trajectory itself is advanced by the HIP library's converged fp32
    solve) both sides take k = 1..4 L-BFGS iterations from identical particle data; while their counters (line-search trials, linear
    iterations, dropped pairs) agree, dv must agree within 1e-4 of max|dv| and the energies within 1e-5 on the first step, 3e-3 (6e-2 at k = 4) / 1e-2 on
    the later ones (relative, floor 1e-3).  The
    oracle runs its wide-sums variant (node sums and inner products in double, like the HIP build: tests/oracle_lib.py wide_sums).
    Round 2 needed dt = 0.03 and 5e-3 here: the B-spline fraction was then evaluated from the rounded product X / dx (3e-5 of a cell in
    float at X / dx ~ 500) where the host-compiled oracle fuses it into an fma (hot_common.h bspline); measured now 1e-6 - 1e-4.  The
    first iteration (no history, no decision yet) must always agree; in total at least 9 of the 12 (step, k) pairs must have been
    comparable.  The bounded runs use cneps = 1e-7 so that they do not terminate early; the trajectory is advanced with the HIP
    library's converged solve at cneps = 1e-4, which must converge and stay finite."""
    T = np.float32
    from hot_amd import synth
    c = synth.cube_cloud(8, ppc=8, dtype=T)
    state = dict(X=c["X"], V=c["V"], C_=None, F=None)
    o, nrm = synth.sticky_floor(5.0, c["dx"])
    # Round 6: the start states of the second and third round are FIXTURES (tests/golden/fp32_states.npz, made by tests/golden/make_fp32_states.py with
    # the HIP library: this cube after one and after four converged fp32 steps), no longer whatever the build under test arrives at: the trajectory is
    # chaotic, and on some of the states it passes through — a node of mass 5e-13 — the oracle's float PCG on the top level breaks down (the reference's
    # cg_smooth divides by du'A du unguarded), which used to cost a whole round of comparisons whenever a last-bit change of the device code moved the
    # trajectory onto such a state.  From every start state the build under test must still take a converged step that stays finite.
    fixtures = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp32_states.npz"))

    def ctx_for(lib, **kw):
        ctx = lib.context(dtype=0, dx=c["dx"], gravity=(0, -9.8, 0), levelCnt=2, **kw)
        ctx.set_particles(state["X"], state["V"], c["mass"], c["vol"], c["mu"], c["lam"], C_=state["C_"], F=state["F"])
        ctx.set_sticky_halfspaces(o, nrm)
        return ctx

    DT = 1.0 / 24
    compared = 0
    from tests.oracle_lib import wide_sums
    counters = ("iterations", "linesearch_trials", "linear_iterations", "dropped_pairs", "vcycles", "num_nodes")
    for step in range(3):
        e4 = None
        for its in (1, 2, 3, 4):
            res = {}
            for name, lib in (("gpu", hotlib), ("cpu", oracle)):
                with wide_sums(name == "cpu"):
                    ctx = ctx_for(lib, max_iterations=its, cneps=1e-7)
                    pc.prepare(ctx, DT)
                    st = ctx.solve()
                    res[name] = (ctx.get_dv().astype(np.float64), st)
                    del ctx
            (dg, sg), (dc, sc) = res["gpu"], res["cpu"]
            assert np.isfinite(dg).all() and np.isfinite(sg["energy"]) and sg["iterations"] == its
            e4 = sg["energy"]
            if not (np.isfinite(dc).all() and np.isfinite(sc["energy"])):
                print("fp32 step %d: the oracle's float solve breaks down on this start state" % step)
                break
            same = all(sg[k] == sc[k] for k in counters)
            err = np.abs(dg - dc).max() / np.abs(dc).max()
            print("fp32 step %d, %d iterations: |ddv| / max|dv| = %.3g, energies %.8g %.8g, counters %s" % (step, its, err, sg["energy"], sc["energy"], "equal" if same else "differ"))
            assert same or its > 1, (sg, sc)
            if not same:
                break
            # first step (F = I, one line-search trial per iteration): round-off; later steps need up to seven halvings per iteration at this
            # time step and the float errors are amplified by the solve (measured 3e-6 - 2.3e-3 and 1e-7 - 5e-3)
            # (on the later steps every iteration amplifies the difference 10 - 20 x: measured 2e-6, 1.4e-4, 1.3e-3, 3e-2 for k = 1..4, so the
            # fourth iteration is held to 1e-1 only and judged on the energy)
            assert err < (1e-4 if step == 0 else (3e-3 if its < 4 else 6e-2)), err  # growth 10 - 20 x per iteration on the later steps (cond ~ 1e8): 2e-6, 1.4e-4, 1.3e-3, 3e-2 measured
            assert abs(sg["energy"] - sc["energy"]) < (1e-5 if step == 0 else 1e-2) * max(abs(sc["energy"]), 1e-3)
            compared += 1
        full = ctx_for(hotlib, max_iterations=300, cneps=1e-4)
        stf = full.advance(DT)
        assert stf["converged"] == 1 and np.isfinite(stf["energy"]) and e4 is not None, (stf, e4)
        p = full.get_particles()
        assert np.isfinite(p["X"]).all() and np.isfinite(p["F"]).all()
        if step < 2:
            state = dict(X=fixtures["X%d" % (step + 1)], V=fixtures["V%d" % (step + 1)], C_=fixtures["C%d" % (step + 1)], F=fixtures["F%d" % (step + 1)])
    assert compared >= 9, compared


KNOB_CFGS = [
    # (hot_config overrides, max_iterations, dv tolerance)
    (dict(levelCnt=3, topDownMGS=1), 5, 1e-9),  # splitLevel 1, no pre-smoothing, PCG on every coarse level (MultigridPreconditioner.h:534-538)
    (dict(levelCnt=3, levelscale=1), 5, 1e-9),  # smoothing iterations grow with the level (:525-551)
    (dict(levelCnt=2, times=2), 5, 1e-9),
    (dict(levelCnt=3, times=3, levelscale=1, smoother=0, coarseSolver=0), 5, 1e-9),  # damped Jacobi with 3 x (times + level) top iterations
    (dict(levelCnt=2, useCN=0, cneps=1e-9), 5, 1e-9),  # plain l2 termination (ImplicitSolver.h:185-209)
    (dict(levelCnt=2, useAdaptiveHessian=1, cneps=1e-13), 19, 1e-6),  # Hessian + hierarchy rebuilt at iteration 16 (LBFGS.h:331-336)
]


@pytest.mark.parametrize("kw,its,tol", KNOB_CFGS, ids=[",".join(f"{k}={v}" for k, v in c[0].items()) for c in KNOB_CFGS])
def test_solver_knobs_against_oracle(hotlib, oracle, kw, its, tol):
    """HOTSettings knobs no other test reaches (Configurations.h:18-42): fixed iteration counts, same control flow, dv to round-off."""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, max_iterations=its, **{"cneps": 1e-7, **kw})
        pc.prepare(ctx)
        st = ctx.solve()
        out[name] = (ctx.get_dv(), st)
    sg, sc = out["gpu"][1], out["cpu"][1]
    for k in ("iterations", "vcycles", "num_levels", "dropped_pairs"):
        assert sg[k] == sc[k], (k, sg, sc)
    if tol < 1e-8:
        assert sg["linesearch_trials"] == sc["linesearch_trials"] and sg["linear_iterations"] == sc["linear_iterations"], (sg, sc)
    assert sg["iterations"] == its, sg  # the knob really ran for the whole budget (adaptiveH: past the rebuild at 16)
    assert rel(out["gpu"][0], out["cpu"][0]) < tol, rel(out["gpu"][0], out["cpu"][0])
    assert abs(sg["energy"] - sc["energy"]) < max(tol, 1e-10) * max(abs(sc["energy"]), 1e-6)


@pytest.mark.parametrize("ppc", [343, 80, 1])
def test_dense_and_sparse_cells_against_oracle(hotlib, oracle, ppc):
    """Cells with hundreds of particles make one cell straddle the 256-particle staging chunks of the scatter kernels
    and the 64-particle chunks of the Hessian tiles; 1 particle per cell is the other extreme."""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=4, ppc=ppc, levelCnt=2)
        pc.prepare(ctx)
        g = ctx.grid()
        e = ctx.update_state(ctx.get_dv())
        r = ctx.residual()
        ctx.build_hessian()
        ctx.build_mg()
        x = np.random.default_rng(1).standard_normal((ctx.Nn, 3))
        out[name] = (g["mass"], g["v"], r, ctx.spmv(0, x), ctx.vcycle(ctx.project(x)), e)
    a, b = out["gpu"], out["cpu"]
    for k in range(5):
        assert rel(a[k], b[k]) < 1e-11, (k, rel(a[k], b[k]))
    assert abs(a[5] - b[5]) < 1e-12 * abs(b[5])


def test_irregular_body_against_oracle(hotlib, oracle):
    """A hollow ball with a bar through it: ragged 4^3 colour blocks, uneven colours, coarse levels with holes."""
    from hot_amd import synth
    c = synth.cube_cloud(14, ppc=8)
    X = c["X"]
    ctr = X.mean(0)
    r = np.linalg.norm(X - ctr, axis=1)
    keep = ((r < 0.066) & (r > 0.03)) | ((np.abs(X[:, 0] - ctr[0]) < 0.012) & (np.abs(X[:, 1] - ctr[1]) < 0.012))
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx = lib.context(dtype=1, dx=c["dx"], gravity=(0, -9.8, 0), levelCnt=3, max_iterations=4, cneps=1e-7)
        ctx.set_particles(X[keep], c["V"][keep], c["mass"][keep], c["vol"][keep], c["mu"][keep], c["lam"][keep])
        o, n = synth.sticky_floor(ctr[1] - 0.07, c["dx"])
        ctx.set_sticky_halfspaces(o, n)
        ctx.sort(), ctx.p2g(), ctx.begin_step(1 / 24)
        st = ctx.solve()
        out[name] = (ctx.get_dv(), st, [ctx.level(l, coords=False)["nrows"] for l in range(st["num_levels"])])
    a, b = out["gpu"], out["cpu"]
    assert a[2] == b[2]
    for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles", "num_levels"):
        assert a[1][k] == b[1][k], (k, a[1], b[1])
    assert rel(a[0], b[0]) < 1e-9


@pytest.mark.parametrize("levelCnt", [1, 2, 3])
def test_incomplete_cholesky_top_solver_against_oracle(hotlib, oracle, levelCnt):
    """-coarseSolver 7 (IC_smooth, MultigridPreconditioner.h:320-323; setup :612-613,684-685).  The reference calls Eigen::IncompleteCholesky
    (AMD ordering, scaling, shift loop), which cannot be restated without Eigen; library and oracle share a block IC(0) in the smoother's
    order with Eigen's shift strategy (hot_amd/csrc/mg_ic.hip, oracle/sim_matrix.hpp setup_ic).  HIP against oracle: the V-cycle with the
    IC top solve to round-off, fixed L-BFGS iterations with equal counters; the V-cycle is symmetric positive; the smoother option 7
    stays rejected inside a hierarchy as in the reference."""
    from hot_amd.binding import HotError
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=10, levelCnt=levelCnt, coarseSolver=7, cneps=1e-7, max_iterations=5)
        pc.prepare(ctx)
        st = ctx.solve()
        rng = np.random.default_rng(3)
        x, y = ctx.project(rng.standard_normal((ctx.Nn, 3))), ctx.project(rng.standard_normal((ctx.Nn, 3)))
        out[name] = (ctx.get_dv(), st, ctx.vcycle(x), ctx.vcycle(y), x, y)
    g, c_ = out["gpu"], out["cpu"]
    for k in ("iterations", "linesearch_trials", "vcycles", "num_levels", "dropped_pairs"):
        assert g[1][k] == c_[1][k], (k, g[1], c_[1])
    assert rel(g[2], c_[2]) < 1e-9 and rel(g[3], c_[3]) < 1e-9
    assert rel(g[0], c_[0]) < 1e-9
    x, y, Mx, My = g[4], g[5], g[2].astype(np.float64), g[3].astype(np.float64)
    assert abs((y * Mx).sum() - (x * My).sum()) < 1e-9 * abs((y * Mx).sum()) and (x * Mx).sum() > 0 and (y * My).sum() > 0
    if levelCnt > 1:
        ctx, c = pc.make_ctx(hotlib, n=6, levelCnt=levelCnt, smoother=7)
        pc.prepare(ctx)
        ctx.update_state(ctx.get_dv())
        ctx.build_hessian()
        with pytest.raises(HotError):
            ctx.build_mg()


def test_incomplete_cholesky_converges_on_gpu(hotlib):
    """Whole solves with the IC top solver converge to the tolerance, from one level (IC is the whole preconditioner) to three."""
    for levelCnt in (1, 3):
        ctx, c = pc.make_ctx(hotlib, n=10, levelCnt=levelCnt, coarseSolver=7, cneps=1e-7, max_iterations=300)
        pc.prepare(ctx)
        e0 = ctx.update_state(ctx.get_dv())
        st = ctx.solve()
        assert st["converged"] == 1 and st["energy"] < e0, st


def test_tiny_and_empty_inputs(hotlib, oracle):
    """One particle (27 nodes, nothing to solve), two particles (a 42-node system) and the empty cloud, which the reference
    rejects with an assertion (MpmSimulationBase.cpp:1071-1072) and the C ABI with HOT_ERR_CAPACITY."""
    from hot_amd import synth
    from hot_amd.binding import HotError
    c = synth.cube_cloud(3, ppc=8)
    for k, tol in ((1, 1e-12), (2, 1e-5)):
        out = {}
        for name, lib in (("gpu", hotlib), ("cpu", oracle)):
            ctx = lib.context(dtype=1, dx=c["dx"], gravity=(0, -9.8, 0), levelCnt=2, cneps=1e-7)
            ctx.set_particles(c["X"][:k], c["V"][:k], c["mass"][:k], c["vol"][:k], c["mu"][:k], c["lam"][:k])
            st = ctx.advance(1 / 24)
            out[name] = (ctx.get_particles(), st)
        (pg, sg), (pcpu, sc) = out["gpu"], out["cpu"]
        assert sg["num_nodes"] == sc["num_nodes"] and abs(sg["iterations"] - sc["iterations"]) <= (0 if k == 1 else 2)
        # both stop at the same CN tolerance; the two-particle system has no boundary and barely any stiffness, so the
        # velocities agree at the solver tolerance only (the reductions on the device are not order-deterministic)
        assert rel(pg["X"], pcpu["X"]) < tol and np.abs(pg["V"] - pcpu["V"]).max() < 5e-3 * max(np.abs(pcpu["V"]).max(), 1e-3)
    ctx = hotlib.context(dtype=1, dx=c["dx"], gravity=(0, -9.8, 0))
    with pytest.raises(HotError):
        ctx.set_particles(c["X"][:0], c["V"][:0], c["mass"][:0], c["vol"][:0], c["mu"][:0], c["lam"][:0])


def test_cfl_step_and_frame_driver_against_oracle(hotlib, oracle):
    """calculateDt (CFL step from the particle speeds) and advanceOneFrame (TimeStepping::nextDt substeps): a fast
    cloud needs several substeps per frame; both implementations take the same ones."""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=6, noise=0.3, levelCnt=2, cneps=1e-7)  # |v| ~ 0.5 => cfl * dx / |v| ~ 1e-2 < 1/24
        cd = ctx.calculate_dt(1.0 / 24)
        n, its, st = ctx.advance_frame(1.0 / 24)
        out[name] = (cd, n, its, ctx.get_particles(), ctx.calculate_dt(1.0 / 24))
    g, c_ = out["gpu"], out["cpu"]
    assert abs(g[0]["dt"] - c_[0]["dt"]) < 1e-14 and abs(g[0]["max_speed"] - c_[0]["max_speed"]) < 1e-14
    assert np.array_equal(g[0]["min_corner"], c_[0]["min_corner"]) and np.array_equal(g[0]["max_corner"], c_[0]["max_corner"])
    assert g[0]["dt"] < 1.0 / 24 and g[1] == c_[1] and g[1] >= 2, (g[1], c_[1], g[0])
    assert abs(g[2] - c_[2]) <= max(2, c_[2] // 10)
    assert np.abs(g[3]["X"] - c_[3]["X"]).max() < 1e-3 * 0.01
    assert abs(g[4]["dt"] - c_[4]["dt"]) < 1e-3 * c_[4]["dt"]


@pytest.mark.parametrize("boundaryType", [0, 1])
def test_analytic_collision_objects_against_oracle(hotlib, oracle, boundaryType):
    """Collision-node generation on the device: a sticky floor, a slip wall with friction, a moving sticky sphere
    poking into the body, a separating half space and a sticky box (multiObjectCollision incl. the Gram-Schmidt of two
    slip normals and the rotation of slip nodes).  boundaryType 1 = the solver's slip mode (rotated dofs)."""
    from hot_amd.binding import BOX, HALFSPACE, SEPARATE, SLIP, SPHERE, STICKY
    objs = [
        dict(shape=HALFSPACE, type=SLIP, p0=(5.0 + 0.0151, 0, 0), p1=(1.0, 0, 0), friction=0.3),  # wall x <= 5.015
        dict(shape=HALFSPACE, type=SLIP, p0=(0, 0, 5.0 + 0.0151), p1=(0, 0.6, 0.8)),  # a second, oblique slip plane
        dict(shape=SPHERE, type=STICKY, p0=(5.03, 5.08, 5.03), p1=0.02, dbdt=(0.0, -0.5, 0.0)),  # moving ball pressed into the top
        dict(shape=HALFSPACE, type=SEPARATE, p0=(0, 5.0 + 0.0049, 0), p1=(0, 1.0, 0), friction=0.1),  # floor that lets go
        dict(shape=BOX, type=STICKY, p0=(5.05, 4.98, 5.05), p1=(5.08, 5.011, 5.08)),
    ]
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, bc=False, levelCnt=2, cneps=1e-7, max_iterations=4, boundaryType=boundaryType)
        ctx.set_collision_objects(objs)
        pc.prepare(ctx)
        dv0 = ctx.get_dv()
        st = ctx.solve()
        out[name] = (dv0, ctx.get_dv(), st)
    g, c_ = out["gpu"], out["cpu"]
    assert rel(g[0], c_[0]) < 1e-13  # Newton initial guess = resolved node velocities
    assert 50 < (np.abs(g[0] - np.array([0, -9.8 / 24, 0])).max(axis=1) > 1e-9).sum() < g[0].shape[0]  # many nodes did collide, not all
    for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles"):
        assert g[2][k] == c_[2][k], (k, g[2], c_[2])
    assert rel(g[1], c_[1]) < 1e-9


def test_composite_level_sets_against_oracle(hotlib, oracle):
    """DisjointUnionLevelSet / DifferenceLevelSet over the primitives (AnalyticLevelSet.h:58-120, AnalyticLevelSet.cpp:148-236): a slip union
    of two spheres and a torus moving into the body, a sticky box with a spherical pocket cut out of it (difference), and a turning union.
    The collision-node set is also checked against a numpy evaluation of min / max of the members' signed distances."""
    from hot_amd.binding import BOX, DIFFERENCE, SLIP, SPHERE, STICKY, TORUS, UNION
    objs = [
        dict(shape=UNION, type=SLIP, friction=0.2, dbdt=(0.0, -0.3, 0.0), members=[
            dict(shape=SPHERE, p0=(5.02, 5.085, 5.02), p1=0.025), dict(shape=SPHERE, p0=(5.055, 5.09, 5.05), p1=0.03),
            dict(shape=TORUS, p0=(5.04, 5.08, 5.04), p1=(0.03, 0.012, 0), lsq=(0.9, 0.1, 0.3, 0.2))]),
        dict(shape=DIFFERENCE, type=STICKY, members=[
            dict(shape=BOX, p0=(4.97, 4.97, 4.97), p1=(5.09, 5.0151, 5.09)), dict(shape=SPHERE, p0=(5.04, 5.0151, 5.04), p1=0.022)]),
        dict(shape=UNION, type=STICKY, b=(5.0, 5.0, 5.0), R=_rot((0, 1, 0), 0.3), omega=(0, 2.0, 0), members=[
            dict(shape=SPHERE, p0=(0.075, 0.04, 0.01), p1=0.02), dict(shape=SPHERE, p0=(0.01, 0.05, 0.075), p1=0.018)]),
    ]
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, bc=False, levelCnt=2, cneps=1e-7, max_iterations=4, boundaryType=1)
        ctx.set_collision_objects(objs)
        pc.prepare(ctx)
        dv0 = ctx.get_dv()
        grid = ctx.grid()
        st = ctx.solve()
        d = ctx.calculate_dt(1.0)
        out[name] = (dv0, ctx.get_dv(), st, grid, d)
    g, c_ = out["gpu"], out["cpu"]
    assert rel(g[0], c_[0]) < 1e-13
    for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles"):
        assert g[2][k] == c_[2][k], (k, g[2], c_[2])
    assert rel(g[1], c_[1]) < 1e-9
    assert abs(g[4]["dt"] - c_[4]["dt"]) < 1e-12 * c_[4]["dt"] and g[4]["max_speed"] > 0.3  # the turning union's corner speed enters the CFL step
    # independent evaluation of the three composites' signed distances at the grid nodes
    x = g[3]["id2coord"].astype(np.float64) * 0.01
    sph = lambda X, c, r: np.linalg.norm(X - np.asarray(c), axis=1) - r
    q = np.array([0.9, 0.1, 0.3, 0.2])
    q /= np.linalg.norm(q)
    w, a, b, cq = q
    Rl = np.array([[1 - 2 * (b * b + cq * cq), 2 * (a * b - w * cq), 2 * (a * cq + w * b)], [2 * (a * b + w * cq), 1 - 2 * (a * a + cq * cq), 2 * (b * cq - w * a)],
                   [2 * (a * cq - w * b), 2 * (b * cq + w * a), 1 - 2 * (a * a + b * b)]])
    P = (x - np.array([5.04, 5.08, 5.04])) @ Rl
    tor = np.hypot(np.hypot(P[:, 0], P[:, 2]) - 0.03, P[:, 1]) - 0.012
    u1 = np.minimum(np.minimum(sph(x, (5.02, 5.085, 5.02), 0.025), sph(x, (5.055, 5.09, 5.05), 0.03)), tor)
    lo, hi = np.array([4.97, 4.97, 4.97]), np.array([5.09, 5.0151, 5.09])
    dd = np.abs(x - (lo + hi) / 2) - (hi - lo) / 2
    box = np.minimum(dd.max(1), 0) + np.linalg.norm(np.maximum(dd, 0), axis=1)
    dif = np.maximum(box, -sph(x, (5.04, 5.0151, 5.04), 0.022))
    Xm = (x - 5.0) @ _rot((0, 1, 0), 0.3)  # R^T (x - b)
    u3 = np.minimum(sph(Xm, (0.075, 0.04, 0.01), 0.02), sph(Xm, (0.01, 0.05, 0.075), 0.018))
    inside = (u1 <= 0) | (dif <= 0) | (u3 <= 0)
    margin = np.minimum(np.minimum(np.abs(u1), np.abs(dif)), np.abs(u3)) > 1e-9  # nodes not sitting on a surface
    moved = np.abs(g[0] - np.array([0, -9.8 / 24, 0])).max(axis=1) > 1e-12
    assert inside.sum() > 100 and (u1 <= 0).sum() > 5 and (u3 <= 0).sum() > 5 and ((box <= 0) & (dif > 0)).sum() > 3  # the pocket really removes nodes
    assert np.array_equal(moved[margin & ~((u1 <= 0) & ~(dif <= 0) & ~(u3 <= 0))], inside[margin & ~((u1 <= 0) & ~(dif <= 0) & ~(u3 <= 0))])  # (slip-only nodes may keep dv = g dt by chance: excluded)


def _rot(axis, angle):
    a = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def test_turning_and_scaling_collision_objects_against_oracle(hotlib, oracle):
    """The full object transform x = R s X + b with its rates: a turning, growing slip sphere, a turned sticky box and
    an oblique slip plane given through R — resolved node velocities (object velocity omega x r + (s'/s) r + b'),
    the solve on top of them, and the CFL step that has to see the objects' corner speeds."""
    from hot_amd.binding import BOX, HALFSPACE, SLIP, SPHERE, STICKY
    R1, R2 = _rot((1, 2, 0.5), 0.7), _rot((0, 0, 1), 0.4)
    objs = [
        # material-space sphere of radius 0.02 at the origin, doubled, placed near the body's top corner, spinning and growing
        dict(shape=SPHERE, type=SLIP, p0=(0, 0, 0), p1=0.02, b=(5.02, 5.085, 5.03), R=R1, s=2.0, dsdt=0.3, omega=(0.0, 3.0, 1.0), dbdt=(0.1, -0.4, 0.0), friction=0.2),
        # unit-ish box turned about z, shrunk, sticky, turning
        dict(shape=BOX, type=STICKY, p0=(-0.04, -0.02, -0.04), p1=(0.04, 0.02, 0.04), b=(5.07, 5.0, 5.07), R=R2, s=0.5, omega=(0, 0, 2.0), dsdt=-0.1),
        # the plane y <= 0 in material space, tilted by R and lifted: an oblique slip floor
        dict(shape=HALFSPACE, type=SLIP, p0=(0, 0, 0), p1=(0, 1.0, 0), b=(5.0, 5.0049, 5.0), R=_rot((1, 0, 0), 0.05)),
    ]
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, bc=False, levelCnt=2, cneps=1e-7, max_iterations=4, boundaryType=1)
        ctx.set_collision_objects(objs)
        cd = ctx.calculate_dt(1.0)
        pc.prepare(ctx)
        dv0 = ctx.get_dv()
        st = ctx.solve()
        out[name] = (dv0, ctx.get_dv(), st, cd)
    g, c_ = out["gpu"], out["cpu"]
    assert rel(g[0], c_[0]) < 1e-12
    assert 30 < (np.abs(g[0] - np.array([0, -9.8 / 24, 0])).max(axis=1) > 1e-9).sum() < g[0].shape[0]
    for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles"):
        assert g[2][k] == c_[2][k], (k, g[2], c_[2])
    assert rel(g[1], c_[1]) < 1e-9
    # the particles are at rest: the step comes from the objects alone
    assert g[3]["max_speed"] > 0.4 and abs(g[3]["max_speed"] - c_[3]["max_speed"]) < 1e-13 and abs(g[3]["dt"] - c_[3]["dt"]) < 1e-15


def test_collision_object_validation(hotlib):
    from hot_amd.binding import HALFSPACE, SLIP, HotError
    ctx, c = pc.make_ctx(hotlib, n=4, bc=False)
    with pytest.raises(HotError):
        ctx.set_collision_objects([dict(shape=HALFSPACE, type=SLIP, p0=(0, 0, 0), p1=(0, 1, 0), s=0.0)])
    with pytest.raises(HotError):
        ctx.set_collision_objects([dict(shape=HALFSPACE, type=SLIP, p0=(0, 0, 0), p1=(0, 1, 0), R=2 * np.eye(3))])
    with pytest.raises(HotError):
        ctx.set_collision_objects([dict(shape=HALFSPACE, type=SLIP, p0=(0, 0, 0), p1=(0, 1, 0), omega=(0, 1, 0))])


@pytest.mark.parametrize("dtype,tol", [(1, 1e-10), (0, 5e-4)])
def test_baseline_geometric_multigrid_against_oracle(hotlib, oracle, dtype, tol):
    """--baseline: every coarse level is an MPM grid of doubled spacing (own sort, mass P2G, DOF numbering, boundaries from
    the collision objects at its own nodes, matrix re-rasterised from the particles); trilinear transfers in between."""
    from hot_amd.binding import HALFSPACE, SLIP, STICKY
    objs = [
        dict(shape=HALFSPACE, type=STICKY, p0=(0, 5.0 + 0.0049, 0), p1=(0, 1.0, 0)),
        dict(shape=HALFSPACE, type=SLIP, p0=(5.0 + 0.0151, 0, 0), p1=(0.8, 0, 0.6)),
    ]
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=12, dtype=dtype, bc=False, levelCnt=3, cneps=1e-7, useBaselineMultigrid=1, boundaryType=1, max_iterations=60)
        ctx.set_collision_objects(objs)
        pc.prepare(ctx)
        ctx.update_state(ctx.get_dv())
        ctx.build_hessian()
        ctx.build_mg()
        out[name] = ctx
    g, c = out["gpu"], out["cpu"]
    mg, mc = level_matrices(g, 3), level_matrices(c, 3)
    for l in range(3):
        assert np.array_equal(g.level(l)["id2coord"], c.level(l)["id2coord"])  # the coarse grids' DOF numbering is bit-exact
        assert abs(mg[l] - mc[l]).max() < tol * abs(mc[l]).max(), l
    assert g.level(1, coords=False)["nrows"] < g.level(0, coords=False)["nrows"] // 4
    for l in range(2):
        gp, cp = g.prolongation(l), c.prolongation(l)
        assert np.array_equal(gp[0], cp[0]) and np.array_equal(gp[1], cp[1])
    if dtype == 0:
        return
    rng = np.random.default_rng(11)
    b = c.project(rng.standard_normal((g.Nn, 3)))
    assert rel(g.vcycle(b), c.vcycle(b)) < 1e-8
    sg, sc = g.solve(), c.solve()
    for k in ("iterations", "converged", "linesearch_trials", "vcycles"):
        assert sg[k] == sc[k], (k, sg, sc)
    assert sc["converged"] == 1
    assert rel(g.get_dv(), c.get_dv()) < 1e-7


def test_baseline_multigrid_needs_analytic_boundaries(hotlib):
    from hot_amd.binding import HotError
    ctx, c = pc.make_ctx(hotlib, n=6, levelCnt=2, useBaselineMultigrid=1, bc=False)
    pc.prepare(ctx)
    nn = ctx.Nn
    ctx.set_bc(np.array([0], np.int32), np.zeros((1, 3, 3)))  # explicit node list: only describes level 0
    ctx.begin_step(1.0 / 24)
    ctx.update_state(ctx.get_dv())
    ctx.build_hessian()
    with pytest.raises(HotError):
        ctx.build_mg()


def test_torus_and_capped_cylinder_against_oracle(hotlib, oracle):
    """The two remaining analytic level sets of the reference scenes: a slip torus (normal = gradient of its distance) and
    a sticky capped cylinder, both behind their own rotation / translation and an object transform on top."""
    from hot_amd.binding import CAPPED_CYLINDER, ROTATED_BOX, SLIP, STICKY, TORUS
    c30, s30 = np.cos(0.3), np.sin(0.3)
    objs = [
        # ring around the top corner of the body, tube radius 0.012, tilted about z, the object itself drifting and turning
        dict(shape=TORUS, type=SLIP, p0=(5.04, 5.07, 5.04), p1=(0.03, 0.012, 0.0), lsq=(c30, 0.0, 0.0, s30), friction=0.1, dbdt=(0.0, -0.2, 0.0), omega=(0.0, 1.0, 0.0), b=(0.0, 0.0, 0.0)),
        # a peg standing in the body, axis tilted about x
        dict(shape=CAPPED_CYLINDER, type=STICKY, p0=(5.02, 5.03, 5.06), p1=(0.015, 0.05, 0.0), lsq=(np.cos(0.2), np.sin(0.2), 0.0, 0.0)),
        # a slab (AnalyticBox: half edges + own rotation) cutting through the far corner
        dict(shape=ROTATED_BOX, type=STICKY, p0=(5.07, 5.01, 5.01), p1=(0.02, 0.006, 0.03), lsq=(np.cos(0.25), 0.0, np.sin(0.25), 0.0)),
    ]
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, bc=False, levelCnt=2, cneps=1e-7, max_iterations=4, boundaryType=1)
        ctx.set_collision_objects(objs)
        cd = ctx.calculate_dt(1.0)
        pc.prepare(ctx)
        dv0 = ctx.get_dv()
        st = ctx.solve()
        out[name] = (dv0, ctx.get_dv(), st, cd)
    g, c_ = out["gpu"], out["cpu"]
    assert rel(g[0], c_[0]) < 1e-12
    assert 20 < (np.abs(g[0] - np.array([0, -9.8 / 24, 0])).max(axis=1) > 1e-9).sum() < g[0].shape[0]
    for k in ("iterations", "linesearch_trials", "linear_iterations", "vcycles"):
        assert g[2][k] == c_[2][k], (k, g[2], c_[2])
    assert rel(g[1], c_[1]) < 1e-9
    assert g[3]["max_speed"] > 0.2 and abs(g[3]["max_speed"] - c_[3]["max_speed"]) < 1e-13


def test_capped_cylinder_must_be_sticky(hotlib):
    from hot_amd.binding import CAPPED_CYLINDER, SLIP, HotError
    ctx, c = pc.make_ctx(hotlib, n=4, bc=False)
    with pytest.raises(HotError):
        ctx.set_collision_objects([dict(shape=CAPPED_CYLINDER, type=SLIP, p0=(5, 5, 5), p1=(0.1, 0.1, 0))])


@pytest.mark.parametrize("bc", [0, 1])
def test_objective_concept_members_against_oracle(hotlib, oracle, bc):
    """The solver-facing members one by one (hot_should_exit, hot_line_search, hot_recover_solution / hot_transform_residual,
    hot_compute_step): the same calls on the HIP library and on the oracle give the same numbers."""
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, levelCnt=2, boundaryType=bc, bc=(bc == 0))
        if bc == 1:  # a slip floor: recoverSolution / transformResidual rotate its nodes
            ctx.set_collision_objects([dict(shape=0, type=2, p0=(0.0, 5.015, 0.0), p1=(0.0, 1.0, 0.0), friction=0.0)])
        pc.prepare(ctx)
        ctx.update_state(ctx.get_dv())
        r = ctx.residual()
        ex, sc = ctx.should_exit(r)
        ctx.build_hessian(), ctx.build_mg()
        d = ctx.project(ctx.vcycle(r))
        x = np.random.default_rng(2).standard_normal(d.shape)
        rec, tr = ctx.recover_solution(x), ctx.transform_residual(x)
        dd, r2, alpha = ctx.line_search(d, 1.0)
        out[name] = (r, ex, sc, d, rec, tr, dd, r2, alpha, ctx.get_dv())
    g, c = out["gpu"], out["cpu"]
    assert g[1] == c[1] and abs(g[2] - c[2]) < 1e-9 * c[2] and g[8] == c[8]
    for k in (0, 3, 4, 5, 6, 7, 9):
        assert rel(g[k], c[k]) < 1e-9, (k, rel(g[k], c[k]))
    if bc == 1:
        assert rel(g[4], g[5]) > 1e-3  # the slip nodes really were rotated (recover != transform)


def test_compute_step_against_oracle(hotlib, oracle):
    out = {}
    for name, lib in (("gpu", hotlib), ("cpu", oracle)):
        ctx, c = pc.make_ctx(lib, n=8, levelCnt=2, lsolver=2)
        pc.prepare(ctx)
        ctx.update_state(ctx.get_dv())
        out[name] = ctx.compute_step(ctx.residual())
    assert rel(out["gpu"], out["cpu"]) < 1e-8, rel(out["gpu"], out["cpu"])
