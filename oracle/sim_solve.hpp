// ORACLE (test infrastructure only).  Nonlinear solvers: L-BFGS with the V-cycle as initial inverse Hessian
// (HOT), projected Newton with inexact (MG-)PCG, line search, termination tests, and the time-step driver.
#pragma once
#include "sim_matrix.hpp"
#include <chrono>
#include <functional>

namespace hot_oracle {

static inline double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the objective's `precondition` std::function (ImplicitSolver.h:47): the MG operator once the hierarchy
// exists (SparseMatrixFast.h:56), lumped-mass scaling before (MultigridSimulation.h:170-176)
template <class T>
void Sim<T>::precondition(const std::vector<TV>& in, std::vector<TV>& out)
{
    if (sysmats.empty() || sysmats[0].diagonalBlock.empty()) {
        out.resize(in.size());
        for (int i = 0; i < num_nodes; ++i) out[i] = in[i] * ((T)1 / mass_matrix[i]);
        return;
    }
    vcycle(in, out);
}

// reference ImplicitSolverObjective::shouldExitByCN (ImplicitSolver.h:174-211) / computeNorm (:158-171)
template <class T>
bool Sim<T>::should_exit(const std::vector<TV>& residual)
{
    if (wide()) { // wide-sums variant (sim_core.hpp): per-node terms in T, their sum in double
        double sn = 0;
        for (int i = 0; i < num_nodes; ++i) sn += (double)(cfg.useCN ? residual[i].squaredNorm() / (nodeCNTol[i] * nodeCNTol[i]) : residual[i].squaredNorm());
        if (!cfg.useCN) {
            stats.final_scaled_residual = std::sqrt(sn);
            return std::sqrt(sn) < cfg.cneps;
        }
        if (num_nodes == 0) return true;
        stats.final_scaled_residual = std::sqrt(sn / num_nodes);
        return sn < num_nodes;
    }
    if (!cfg.useCN) {
        T ns = 0;
        for (int i = 0; i < num_nodes; ++i) ns += residual[i].squaredNorm();
        T res = std::sqrt(ns);
        stats.final_scaled_residual = res;
        return res < (T)cfg.cneps;
    }
    T scaledNorm = 0;
    for (int i = 0; i < num_nodes; ++i) scaledNorm += residual[i].squaredNorm() / (nodeCNTol[i] * nodeCNTol[i]);
    if (num_nodes == 0) return true;
    stats.final_scaled_residual = std::sqrt(scaledNorm / num_nodes);
    return scaledNorm < num_nodes;
}

// reference ImplicitSolverObjective::lineSearch (ImplicitSolver.h:312-333)
template <class T>
T Sim<T>::line_search(std::vector<TV>& ddv, std::vector<TV>& residual, T alpha)
{
    std::vector<TV> dvnew(ddv.size());
    recover_solution(ddv);
    double Ek0 = Ek;
    int guard = 0;
    do {
        HOT_FAIR_FOR
        for (size_t i = 0; i < ddv.size(); ++i) dvnew[i] = dv0[i] + ddv[i] * alpha;
        update_state(dvnew);
        stats.linesearch_trials++;
        alpha *= (T)0.5;
        // `Ek > Ek0` in the reference.  Written so that a NaN energy (an exploded L-BFGS direction in float: the trial point is
        // outside anything representable) counts as a rejection and the step is halved, instead of ending the search with NaN
        // accepted; identical for every finite energy.  At most 60 halvings (2^-60 of the step is no step).
    } while (!(Ek <= Ek0) && ++guard < 60);
    alpha *= 2;
    HOT_FAIR_FOR
    for (size_t i = 0; i < ddv.size(); ++i) ddv[i] = ddv[i] * alpha;
    transform_residual(ddv);
    compute_residual(residual);
    updated = true;
    dv0 = dvnew;
    return alpha;
}

// reference LBFGS::solve (Lib/Ziran/Math/Nonlinear/LBFGS.h:300-437); RingBuffer :23-69 is emulated with
// vectors (history 8 => at most 8 stored pairs + the working slot)
template <class T>
bool Sim<T>::lbfgs_solve()
{
    constexpr int historySize = 8;
    std::vector<TV>& x = dv;
    std::vector<TV> residual(num_nodes);
    // updateState / computeResidual honour the `updated` flag (ImplicitSolver.h:132,241)
    auto updateState = [&]() {
        if (updated) return;
        update_state(x);
    };
    auto computeResidual = [&]() {
        if (updated) return;
        compute_residual(residual);
    };
    updateState();
    computeResidual();
    struct Pair {
        std::vector<TV> dx, dg;
        T dgTdx;
    };
    std::vector<Pair> hist; // oldest first; back() is the working slot
    hist.emplace_back();
    std::array<T, historySize + 1> ksi;
    auto push_back = [&]() {
        hist.emplace_back();
        if ((int)hist.size() > historySize + 1) hist.erase(hist.begin());
    };
    for (int it = 0; it < cfg.max_iterations; ++it) {
        stats.iterations = it;
        if (should_exit(residual)) {
            stats.converged = 1;
            return true;
        }
        bool rebuild = cfg.useAdaptiveHessian ? ((it & 0xf) == 0) : (it == 0);
        if (rebuild) {
            // HinvApproxInit (ImplicitSolver.h:335-353)
            double t0 = now_ms();
            build_matrix();
            double t1 = now_ms();
            build_mg();
            double t2 = now_ms();
            stats.ms_hessian += t1 - t0, stats.ms_mg_build += t2 - t1;
            hist.clear();
            hist.emplace_back();
        }
        hist.back().dg = residual;
        for (int i = (int)hist.size() - 2; i >= 0; --i) {
            ksi[i] = dot_product(hist[i].dx, residual) * hist[i].dgTdx;
            HOT_FAIR_FOR
            for (int n = 0; n < num_nodes; ++n) residual[n] -= hist[i].dg[n] * ksi[i];
        }
        hist.back().dx.resize(num_nodes);
        precondition(residual, hist.back().dx);
        project(hist.back().dx);
        for (int i = 0; i < (int)hist.size() - 1; ++i) {
            T c = ksi[i] - dot_product(hist[i].dg, hist.back().dx) * hist[i].dgTdx;
            HOT_FAIR_FOR
            for (int n = 0; n < num_nodes; ++n) hist.back().dx[n] += hist[i].dx[n] * c;
        }
        if (cfg.linesearch) line_search(hist.back().dx, residual, (T)1);
        recover_solution(hist.back().dx);
        HOT_FAIR_FOR
        for (int n = 0; n < num_nodes; ++n) x[n] += hist.back().dx[n];
        transform_residual(hist.back().dx);
        updateState();
        computeResidual();
        HOT_FAIR_FOR
        for (int n = 0; n < num_nodes; ++n) hist.back().dg[n] -= residual[n];
        hist.back().dgTdx = (T)1 / dot_product(hist.back().dg, hist.back().dx);
        if (hist.back().dgTdx <= 0) {
            hist.pop_back();
            stats.dropped_pairs++;
        }
        push_back();
    }
    stats.iterations = cfg.max_iterations;
    return false;
}

// reference ExtendedNewtonsMethod::solve (Lib/Ziran/Math/Nonlinear/ExtendedNewtonsMethod.h:39-66) +
// ImplicitSolverObjective::computeStep (ImplicitSolver.h:355-432) + InexactConjugateGradient::solve
// (Lib/Ziran/Math/Linear/InexactConjugateGradient.h:49-103)
// reference Minres<T, TM, TV>::solve + applyAllPreviousGivensRotationsAndDetermineNewGivens (Minres.h:69-178)
template <class T>
int Sim<T>::minres_solve(const std::function<void(const std::vector<TV>&, std::vector<TV>&)>& Amul,
    const std::function<void(const std::vector<TV>&, std::vector<TV>&)>& prec, std::vector<TV>& x, const std::vector<TV>& b, T relative_tolerance, T tolerance,
    int max_iterations)
{
    const int n = num_nodes;
    std::vector<TV> mk(n, TV::zero()), mkm1(n, TV::zero()), mkm2(n, TV::zero()), z(n, TV::zero()), qkm1(n, TV::zero()), qk(n, TV::zero()), qkp1(n, TV::zero());
    T gamma = 0, delta = 0, epsilon = 0, beta_kp1 = 0, alpha_k = 0, beta_k = 0, tk = 0;
    Givens<T> Gk(0, 1), Gkm1(0, 1), Gkm2(0, 1);
    auto rot2 = [](const Givens<T>& G, T& a, T& b2) { // rowRotation on a 2-vector
        T t1 = a, t2 = b2;
        a = G.c * t1 - G.s * t2;
        b2 = G.s * t1 + G.c * t2;
    };
    Amul(x, qkp1);
    HOT_FAIR_FOR
    for (int i = 0; i < n; ++i) qkp1[i] = b[i] - qkp1[i];
    project(qkp1);
    prec(qkp1, z);
    T rpn = std::sqrt(dot_product(z, qkp1));
    beta_kp1 = rpn;
    T local_tolerance = std::min(relative_tolerance * rpn, tolerance);
    if (rpn < local_tolerance) return 0;
    if (rpn > 0)
        for (int i = 0; i < n; ++i)
                for (int d = 0; d < 3; ++d) qkp1[i].a[d] /= beta_kp1, z[i].a[d] /= beta_kp1;
    T rhs0 = rpn, rhs1 = 0; // last two components of the Givens-transformed least-squares rhs
    for (int k = 0; k < max_iterations; ++k) {
        if (rpn < local_tolerance) return k;
        mkm2.swap(mkm1);
        mkm1.swap(mk);
        mk = z;
        beta_k = beta_kp1;
        qkm1.swap(qkp1);
        qkm1.swap(qk);
        Amul(mk, qkp1);
        project(qkp1);
        alpha_k = dot_product(mk, qkp1);
        HOT_FAIR_FOR
        for (int i = 0; i < n; ++i) qkp1[i] = qkp1[i] - qk[i] * alpha_k;
        HOT_FAIR_FOR
        for (int i = 0; i < n; ++i) qkp1[i] = qkp1[i] - qkm1[i] * beta_k;
        prec(qkp1, z);
        beta_kp1 = std::sqrt(std::max((T)0, dot_product(z, qkp1)));
        if (beta_kp1 > 0)
            for (int i = 0; i < n; ++i)
                for (int d = 0; d < 3; ++d) qkp1[i].a[d] /= beta_kp1, z[i].a[d] /= beta_kp1;
        // applyAllPreviousGivensRotationsAndDetermineNewGivens
        Gkm2 = Gkm1;
        Gkm1 = Gk;
        T e0 = 0, e1 = beta_k;
        rot2(Gkm2, e0, e1);
        epsilon = e0;
        T d0 = e1, d1 = alpha_k;
        rot2(Gkm1, d0, d1);
        delta = d0;
        T t0 = d1, t1 = beta_kp1;
        Gk.compute(t0, t1);
        rot2(Gk, t0, t1);
        gamma = t0;
        rot2(Gk, rhs0, rhs1);
        tk = rhs0;
        T res = rhs1;
        rhs0 = res, rhs1 = 0;
        rpn = res < 0 ? -res : res;
        for (int i = 0; i < n; ++i) {
            TV t = mk[i] - mkm1[i] * delta - mkm2[i] * epsilon;
            for (int d = 0; d < 3; ++d) mk[i].a[d] = t.a[d] / gamma;
        }
        HOT_FAIR_FOR
        for (int i = 0; i < n; ++i) x[i] += mk[i] * tk;
    }
    return max_iterations;
}

// reference ImplicitSolverObjective::computeStep (ImplicitSolver.h:355-432): rebuild the matrix and the hierarchy (or the
// matrix-free block diagonal), then InexactConjugateGradient::solve (InexactConjugateGradient.h:49-103) or Minres::solve
template <class T>
void Sim<T>::compute_step(const std::vector<TV>& residual, std::vector<TV>& step)
{
    // cg.tolerance: scene value 1e-4 (MultigridInit3D.h:2500-2501) unless useCN sets maxcntol (MultigridSimulation.h:201-208)
    T cg_tolerance = cfg.useCN ? max_cn_tolerance : (T)1e-4;
    // computeStep
    step.assign(num_nodes, TV::zero());
    std::function<void(const std::vector<TV>&, std::vector<TV>&)> prec;
    if (!cfg.matrixFree) {
        double t0 = now_ms();
        build_matrix();
        double t1 = now_ms();
        // ImplicitSolver.h:365: with the mass preconditioner (Ainv 2, lsolver 1/2 only) no hierarchy is built and
        // `precondition` stays the lumped-mass scaling installed by startBackwardEuler
        const bool massPrec = (cfg.lsolver == 1 || cfg.lsolver == 2) && cfg.Ainv == 2;
        if (!massPrec) build_mg();
        double t2 = now_ms();
        stats.ms_hessian += t1 - t0, stats.ms_mg_build += t2 - t1;
        if (massPrec)
            prec = [&](const std::vector<TV>& in, std::vector<TV>& out) {
                out.resize(in.size());
                for (int i = 0; i < num_nodes; ++i) out[i] = in[i] * ((T)1 / mass_matrix[i]);
            };
        else if (cfg.levelCnt == 1 && cfg.times == 1)
            prec = [&](const std::vector<TV>& in, std::vector<TV>& out) { scaler(in, out, sysmats[0]); };
        else
            prec = [&](const std::vector<TV>& in, std::vector<TV>& out) { vcycle(in, out); };
    }
    else {
        // buildDiagonal (ImplicitSolver.h:605-665): block diagonal of the matrix-free operator
        std::vector<TM> diag(num_nodes);
        HOT_FAIR_FOR
        for (int n = 0; n < num_nodes; ++n) diag[n] = (sharded() && comm.rank != 0) ? TM::zero() : TM::identity() * mass_matrix[n];
        bool proj = cfg.project != 0;
        const bool wd = wide();
        if (wd) wacc.assign((size_t)num_nodes * 9, 0.0);
        for_each_particle_colored([&](int g, int i) {
            CorotatedScratch<T> s;
            corotated_update_scratch(F[i], mu[i], lambda[i], proj, s);
            T ddF[81];
            corotated_first_piola_derivative(s, ddF);
            TM FnT = Fn[i].transpose();
            Spline sp;
            compute_spline(X[i], sp);
            iterate_kernel(sp, g, particle_base_offset[i], [&](const int*, T, const TV& dw, Node& gs) {
                if (gs.idx < 0) return;
                TV wi = FnT * dw;
                TM dFdX = TM::zero();
                for (int q = 0; q < 3; ++q)
                    for (int v = 0; v < 3; ++v)
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) dFdX(r, c) += ddF[(3 * v + r) + 9 * (3 * q + c)] * wi(v) * wi(q);
                if (wd) {
                    TM dl = dFdX * (dt * dt * vol[i]);
                    for (int k = 0; k < 9; ++k) wacc[(size_t)gs.idx * 9 + k] += (double)dl.a[k];
                    return;
                }
                diag[gs.idx] += dFdX * (dt * dt * vol[i]);
            });
        });
        if (wd)
            for (int n = 0; n < num_nodes; ++n)
                for (int k = 0; k < 9; ++k) diag[n].a[k] = (T)((double)diag[n].a[k] + wacc[(size_t)n * 9 + k]);
        allreduce(diag.data(), (int64_t)num_nodes * 9, REAL);
        std::vector<TM> dinv(num_nodes);
        for (int n = 0; n < num_nodes; ++n) {
            if (cfg.Ainv == 0) {
                dinv[n] = TM::zero();
                for (int k = 0; k < 3; ++k) dinv[n](k, k) = 1 / diag[n](k, k);
            }
            else
                dinv[n] = inverse(diag[n]);
        }
        prec = [dinv](const std::vector<TV>& in, std::vector<TV>& out) {
            out.resize(in.size());
            for (size_t n = 0; n < in.size(); ++n) out[n] = dinv[n] * in[n];
        };
    }
    auto Amul = [&](const std::vector<TV>& xx, std::vector<TV>& bb) {
        if (cfg.matrixFree)
            matfree_multiply(xx, bb);
        else
            multiply(sysmats[0], xx, bb);
    };
    std::vector<TV> b = residual;
    if (cfg.systemBCProject) {
        HOT_FAIR_FOR
        for (int n = 0; n < num_nodes; ++n) b[n] += dRhs[n];
    }
    if (cfg.lsolver == 1) {
        // Minres::solve (Lib/Ziran/Math/Linear/Minres.h:69-149) with relative tolerance from the Newton loop
        // (ExtendedNewtonsMethod.h:57) and tolerance = maxcntol (MultigridSimulation.h:204) or the scene's 1e-4
        T residual_norm = std::sqrt(dot_product(residual, residual));
        T newton_tol = cfg.useCN ? max_cn_tolerance : (T)cfg.cneps;
        T rel = std::min((T)0.5, std::sqrt(std::max(residual_norm, newton_tol)));
        stats.linear_iterations += minres_solve(Amul, prec, step, b, rel, cg_tolerance, cfg.linear_iteration_cap > 0 ? cfg.linear_iteration_cap : 10000);
    }
    else
    // inexact PCG
    {
        std::vector<TV> r(num_nodes), p(num_nodes), q(num_nodes), temp(num_nodes);
        Amul(step, temp);
        HOT_FAIR_FOR
        for (int n = 0; n < num_nodes; ++n) r[n] = b[n] - temp[n];
        project(r);
        prec(r, q);
        p = q;
        T zTrk = dot_product(r, q);
        T rpn = std::sqrt(zTrk);
        T forcing = std::min((T)0.5, std::sqrt(std::max(rpn, cg_tolerance)));
        T local_tol = forcing * rpn;
        int cnt = 0;
        for (; cnt < (cfg.linear_iteration_cap > 0 ? cfg.linear_iteration_cap : 10000); ++cnt) {
            if (rpn < local_tol) break;
            Amul(p, temp);
            project(temp);
            T alpha = zTrk / dot_product(temp, p);
            HOT_FAIR_FOR
            for (int n = 0; n < num_nodes; ++n) step[n] += p[n] * alpha, r[n] -= temp[n] * alpha;
            prec(r, q);
            T zTrk_last = zTrk;
            zTrk = dot_product(q, r);
            T beta = zTrk / zTrk_last;
            HOT_FAIR_FOR
            for (int n = 0; n < num_nodes; ++n) p[n] = q[n] + p[n] * beta;
            rpn = std::sqrt(zTrk);
        }
        stats.linear_iterations += cnt;
    }
}

template <class T>
bool Sim<T>::newton_solve()
{
    std::vector<TV>& x = dv;
    std::vector<TV> residual(num_nodes), step(num_nodes);
    for (int it = 0; it < cfg.max_iterations; ++it) {
        stats.iterations = it;
        if (!updated) {
            update_state(x);
            compute_residual(residual);
        }
        if (should_exit(residual)) {
            stats.converged = 1;
            return true;
        }
        compute_step(residual, step);
        if (cfg.linesearch) line_search(step, residual, (T)1);
        recover_solution(step);
        HOT_FAIR_FOR
        for (int n = 0; n < num_nodes; ++n) x[n] += step[n];
        transform_residual(step);
    }
    stats.iterations = cfg.max_iterations;
    return false;
}

// reference MultigridSimulation::backwardEulerStep (MultigridSimulation.h:188-233), after startBackwardEuler
template <class T>
int Sim<T>::solve()
{
    std::memset(&stats, 0, sizeof(stats));
    double t0 = now_ms();
    if (cfg.useCN) evaluate_cn_tolerance();
    // a fresh step has no hierarchy yet
    sysmats.clear();
    bool ok = cfg.lsolver == 3 ? lbfgs_solve() : newton_solve();
    (void)ok;
    stats.num_nodes = num_nodes;
    stats.num_levels = (int)sysmats.size();
    stats.energy = Ek;
    stats.ms_solve = now_ms() - t0;
    return 0;
}

// reference MultigridSimulation::advanceOneTimeStep (MultigridSimulation.h:235-297)
template <class T>
int Sim<T>::advance(double dt_)
{
    double t0 = now_ms();
    int rc = sort_particles();
    if (rc) return rc;
    double t1 = now_ms();
    particles_to_grid();
    double t2 = now_ms();
    begin_step((T)dt_);
    double t3 = now_ms();
    solve();
    double t4 = now_ms();
    grid_to_particles(dt_);
    double t5 = now_ms();
    stats.ms_sort = t1 - t0, stats.ms_p2g = t2 - t1, stats.ms_begin = t3 - t2, stats.ms_g2p = t5 - t4, stats.ms_total = t5 - t0;
    return 0;
}

// reference MpmSimulationBase::calculateDt (Lib/MPM/MpmSimulationBase.cpp:789-814) + evalMaxParticleSpeed (:1186-1218);
// collision objects are static (their evalMaxSpeed term is 0)
template <class T>
double Sim<T>::calculate_dt(double max_dt, double* max_speed, double* min_corner, double* max_corner)
{
    T ms = 0;
    T hi[3] = { -(T)3.4e38, -(T)3.4e38, -(T)3.4e38 }, nlo[3] = { -(T)3.4e38, -(T)3.4e38, -(T)3.4e38 };
    for (size_t p = 0; p < X.size(); ++p) {
        ms = std::max(ms, (T)std::sqrt(Vel[p].squaredNorm()));
        for (int d = 0; d < 3; ++d) hi[d] = std::max(hi[d], X[p](d)), nlo[d] = std::max(nlo[d], -X[p](d));
    }
    if (sharded()) {
        T m7[7] = { ms, hi[0], hi[1], hi[2], nlo[0], nlo[1], nlo[2] };
        allreduce(m7, 7, REAL, HOT_COMM_MAX);
        ms = m7[0];
        for (int d = 0; d < 3; ++d) hi[d] = m7[1 + d], nlo[d] = m7[4 + d];
    }
    auto prim_bounds = [](const hot_collision_object& o, double (&blo)[3], double (&bhi)[3]) {
        for (int d = 0; d < 3; ++d) {
            // ls->getBounds: Sphere, Torus (r0 + r1), CappedCylinder (sqrt(r^2 + (h/2)^2)), AnalyticBox (|half edges|), AxisAlignedAnalyticBox
            const double rad = o.shape == HOT_SHAPE_SPHERE ? o.p1[0] : o.shape == HOT_SHAPE_TORUS ? o.p1[0] + o.p1[1] : o.shape == HOT_SHAPE_ROTATED_BOX ? std::sqrt(o.p1[0] * o.p1[0] + o.p1[1] * o.p1[1] + o.p1[2] * o.p1[2]) : std::sqrt(o.p1[0] * o.p1[0] + 0.25 * o.p1[1] * o.p1[1]);
            const bool round_ = o.shape == HOT_SHAPE_SPHERE || o.shape == HOT_SHAPE_TORUS || o.shape == HOT_SHAPE_CAPPED_CYLINDER || o.shape == HOT_SHAPE_ROTATED_BOX;
            blo[d] = round_ ? o.p0[d] - rad : o.p0[d];
            bhi[d] = round_ ? o.p0[d] + rad : o.p1[d];
        }
    };
    for (size_t ko = 0; ko < cobjs.size(); ko += 1 + members_of(cobjs[ko])) { // MpmSimulationBase.cpp:802-806 with AnalyticCollisionObject::evalMaxSpeed (CollisionObject.cpp:200-238)
        const auto& o = cobjs[ko];
        const double wn = std::sqrt(o.omega[0] * o.omega[0] + o.omega[1] * o.omega[1] + o.omega[2] * o.omega[2]);
        double best = 0;
        if (o.dsdt == 0 && wn == 0)
            best = std::sqrt(o.dbdt[0] * o.dbdt[0] + o.dbdt[1] * o.dbdt[1] + o.dbdt[2] * o.dbdt[2]);
        else {
            double pmin[3], pmax[3], blo[3], bhi[3];
            for (int d = 0; d < 3; ++d) pmin[d] = (double)(-nlo[d] - (T)4 * dx), pmax[d] = (double)(hi[d] + (T)4 * dx); // particle box expanded by (degree + 2) dx
            if (o.shape == HOT_SHAPE_UNION) { // DisjointUnionLevelSet::getBounds (AnalyticLevelSet.cpp:157-167)
                for (int d = 0; d < 3; ++d) blo[d] = 1.7e308, bhi[d] = -1.7e308;
                for (int m = 1; m <= members_of(o); ++m) {
                    double l[3], h[3];
                    prim_bounds(cobjs[ko + m], l, h);
                    for (int d = 0; d < 3; ++d) blo[d] = std::min(blo[d], l[d]), bhi[d] = std::max(bhi[d], h[d]);
                }
            }
            else if (o.shape == HOT_SHAPE_DIFFERENCE) // DifferenceLevelSet::getBounds (:239-242)
                prim_bounds(cobjs[ko + 1], blo, bhi);
            else
                prim_bounds(o, blo, bhi);
            std::vector<std::array<double, 3>> corners;
            for (int i = 0; i < 8; ++i) {
                std::array<double, 3> x, X;
                for (int d = 0; d < 3; ++d) x[d] = (i & (1 << d)) ? pmin[d] : pmax[d];
                for (int k = 0; k < 3; ++k) X[k] = (o.R[3 * k] * (x[0] - o.b[0]) + o.R[3 * k + 1] * (x[1] - o.b[1]) + o.R[3 * k + 2] * (x[2] - o.b[2])) / o.s;
                if ((blo[0] < X[0] || blo[1] < X[1] || blo[2] < X[2]) && (X[0] < bhi[0] || X[1] < bhi[1] || X[2] < bhi[2])) corners.push_back(x);
            }
            for (int i = 0; i < 8; ++i) {
                std::array<double, 3> x, X;
                for (int d = 0; d < 3; ++d) X[d] = (i & (1 << d)) ? blo[d] : bhi[d];
                for (int k = 0; k < 3; ++k) x[k] = (o.R[k] * X[0] + o.R[3 + k] * X[1] + o.R[6 + k] * X[2]) * o.s + o.b[k];
                if ((pmin[0] < x[0] || pmin[1] < x[1] || pmin[2] < x[2]) && (x[0] < pmax[0] || x[1] < pmax[1] || x[2] < pmax[2])) corners.push_back(x);
            }
            for (const auto& x : corners) {
                const double xb[3] = { x[0] - o.b[0], x[1] - o.b[1], x[2] - o.b[2] }, ss = o.dsdt / o.s;
                const double v[3] = { o.omega[1] * xb[2] - o.omega[2] * xb[1] + ss * xb[0] + o.dbdt[0], o.omega[2] * xb[0] - o.omega[0] * xb[2] + ss * xb[1] + o.dbdt[1],
                    o.omega[0] * xb[1] - o.omega[1] * xb[0] + ss * xb[2] + o.dbdt[2] };
                best = std::max(best, std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
            }
        }
        ms = std::max(ms, (T)best);
    }
    T dtc = (T)max_dt;
    if (ms) dtc = (T)cfg.cfl * dx / ms;
    if (max_speed) *max_speed = (double)ms;
    for (int d = 0; d < 3; ++d) {
        if (max_corner) max_corner[d] = (double)hi[d];
        if (min_corner) min_corner[d] = -(double)nlo[d];
    }
    return (double)dtc;
}

// reference SimulationBase::advanceOneFrame (Lib/Ziran/Sim/SimulationBase.h:291-327) with TimeStepping::nextDt / advance
// (Lib/Ziran/Sim/TimeStepping.h:45-76)
template <class T>
int Sim<T>::advance_frame(double frame_dt, double min_dt, double max_dt, int* substeps, int* iterations_total)
{
    double since = 0;
    int n = 0, its = 0, rc = 0;
    for (;;) {
        double dtc = calculate_dt(max_dt, nullptr, nullptr, nullptr);
        double d = (dtc < min_dt) ? min_dt : (dtc > max_dt) ? max_dt : dtc;
        if (since + d >= frame_dt)
            d = frame_dt - since;
        else if (since + 2 * d > frame_dt)
            d = (frame_dt - since) / 2;
        rc = advance(d);
        if (rc) break;
        its += stats.iterations;
        ++n;
        since += d;
        if (since >= frame_dt) break;
    }
    if (substeps) *substeps = n;
    if (iterations_total) *iterations_total = its;
    return rc;
}

} // namespace hot_oracle
