// ORACLE (test infrastructure only).  Particle-state update, elastic force rasterisation, residual, energy,
// characteristic-norm tolerances and the matrix-free Hessian product.
#pragma once
#include "sim_core.hpp"

namespace hot_oracle {

// reference MpmForceBase::evalInterpolantAndGradient (Lib/MPM/Force/MpmForceBase.cpp:213-248)
template <class T>
void Sim<T>::eval_interpolant_and_gradient(const std::vector<TV>& f)
{
#pragma omp parallel for schedule(static)
    for (size_t s = 0; s < nodes.size(); ++s) nodes[s].new_v = TV::zero();
    iterate_grid([&](const int*, Node& g) { g.new_v = f[g.idx]; });
    for_each_particle_colored([&](int g, int i) {
        TM grad = TM::zero();
        TV val = TV::zero();
        Spline s;
        compute_spline(X[i], s);
        iterate_kernel(s, g, particle_base_offset[i], [&](const int*, T w, const TV& dw, Node& gs) {
            grad += outer(gs.new_v, dw);
            val += gs.new_v * w;
        });
        scratch_gradV[i] = grad;
        scratch_vp[i] = val;
    });
}

// reference MpmForceBase::updatePositionBasedState (MpmForceBase.cpp:309-328): computeVAndGradV (:86-91),
// restoreStrain / evolveStrain (FBasedMpmForceHelper.cpp:35-43,99-114), updateParticleImplicitState
// (MpmForceBase.cpp:184-208) -> FBasedMpmForceHelper::updateImplicitState (FBasedMpmForceHelper.cpp:70-97)
template <class T>
void Sim<T>::update_position_based_state()
{
    std::vector<TV> f(num_nodes);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < num_nodes; ++n) f[n] = vn[n] + dv[n];
    eval_interpolant_and_gradient(f);
    bool proj = cfg.project != 0;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < Np; ++p) {
        F[p] = Fn[p]; // restoreStrain
        F[p] = (TM::identity() + scratch_gradV[p] * dt) * F[p]; // evolveStrain
        CorotatedScratch<T> s;
        corotated_update_scratch(F[p], mu[p], lambda[p], proj, s);
        TM P = corotated_first_piola(s, mu[p], lambda[p]);
        scratch_stress[p] = (P * vol[p]) * Fn[p].transpose();
    }
}

// reference MpmForceBase::totalEnergy (MpmForceBase.cpp:351-369) -> FBasedMpmForceHelper::totalEnergy
// (FBasedMpmForceHelper.cpp:116-135); accumulated in double whatever T is
template <class T>
double Sim<T>::force_total_energy()
{
    double e = 0;
    bool proj = cfg.project != 0;
#pragma omp parallel for schedule(static) reduction(+ : e)
    for (int64_t p = 0; p < Np; ++p) {
        CorotatedScratch<T> s;
        corotated_update_scratch(F[p], mu[p], lambda[p], proj, s);
        e += vol[p] * (psi_invariants_flag() ? corotated_psi_product_form(s, mu[p], lambda[p]) : corotated_psi(s, mu[p], lambda[p]));
    }
    allreduce(&e, 1, HOT_COMM_F64); // sharded: sum of the shards' strain energies
    return e;
}

// reference ImplicitSolverObjective::totalEnergy (ImplicitSolver.h:254-275) + MassLumpedInertia::totalEnergy
// (Lib/Ziran/Physics/LagrangianForce/Inertia.cpp:16-29)
template <class T>
double Sim<T>::total_energy()
{
    double result = (T)force_total_energy();
    double ke = 0, ge = 0;
#pragma omp parallel for schedule(static) reduction(+ : ke, ge)
    for (int n = 0; n < num_nodes; ++n) {
        ke += dv[n].squaredNorm() * mass_matrix[n];
        ge += gravity.dot(dv[n]) * mass_matrix[n];
    }
    result += ke / 2;
    result -= dt * ge;
    return result;
}

// reference ImplicitSolverObjective::updateState (ImplicitSolver.h:237-252) + moveNodes (MpmSimulationBase.cpp:736-747)
template <class T>
void Sim<T>::update_state(const std::vector<TV>& dv_in)
{
    if (&dv_in != &dv) dv = dv_in;
    update_position_based_state();
    if (cfg.linesearch) Ek = total_energy();
}

// reference MpmForceBase::rasterizeForceToTVStack<false> (MpmForceBase.cpp:100-153)
template <class T>
void Sim<T>::rasterize_force(T scale, std::vector<TV>& force)
{
    iterate_grid([&](const int*, Node& g) { g.new_v = TV::zero(); });
    const bool wd = wide();
    if (wd) wacc.assign(nodes.size() * 3, 0.0);
    for_each_particle_colored([&](int g, int i) {
        const TM& stress = scratch_stress[i];
        Spline s;
        compute_spline(X[i], s);
        iterate_kernel(s, g, particle_base_offset[i], [&](const int*, T w, const TV& dw, Node& gs) {
            TV delta = stress * dw; // fp == 0 for F-based MPM forces
            if (wd) {
                double* a = &wacc[(size_t)(&gs - nodes.data()) * 3];
                for (int d = 0; d < 3; ++d) a[d] -= (double)(delta(d) * scale);
                return;
            }
            gs.new_v -= delta * scale;
        });
    });
    if (wd) iterate_grid([&](const int*, Node& g) {
        const double* a = &wacc[(size_t)(&g - nodes.data()) * 3];
        g.new_v = TV{ { (T)a[0], (T)a[1], (T)a[2] } };
    });
    if (sharded()) {
        std::vector<T> buf(nodes.size() * 3);
        for (size_t s = 0; s < nodes.size(); ++s)
            for (int d = 0; d < 3; ++d) buf[3 * s + d] = nodes[s].new_v(d);
        allreduce(buf.data(), (int64_t)buf.size(), REAL);
        for (size_t s = 0; s < nodes.size(); ++s) nodes[s].new_v = TV{ { buf[3 * s], buf[3 * s + 1], buf[3 * s + 2] } };
    }
    iterate_grid([&](const int*, Node& g) { force[g.idx] += g.new_v; });
}

// reference ImplicitSolverObjective::computeResidual (ImplicitSolver.h:128-155), MassLumpedInertia::addScaledForces
// (Inertia.cpp:33-41)
template <class T>
void Sim<T>::compute_residual(std::vector<TV>& residual)
{
    residual.resize(num_nodes);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < num_nodes; ++n) residual[n] = (gravity * dt) * mass_matrix[n];
    rasterize_force(dt, residual);
    T scale = dt / dt;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < num_nodes; ++n) residual[n] -= dv[n] * (scale * mass_matrix[n]);
    transform_residual(residual);
    project(residual);
    rhs = residual;
}

// reference ImplicitSolverObjective::evaluatePerNodeCNTolerance (ImplicitSolver.h:667-696) with
// FBasedMpmForceHelper::computePerNodeCNTolerance (FBasedMpmForceHelper.h:123-157): dPdF at F = I, Frobenius norm
template <class T>
void Sim<T>::evaluate_cn_tolerance()
{
    nodeCNTol.assign(num_nodes, 0);
    bool proj = cfg.project != 0;
    T max_nrm = -1;
    for (int64_t p = 0; p < Np; ++p) { // computeCharacteristicNorm: dPdFNorm_max over the particles' models
        if (p > 0 && mu[p] == mu[p - 1] && lambda[p] == lambda[p - 1]) continue;
        CorotatedScratch<T> s;
        corotated_update_scratch(TM::identity(), mu[p], lambda[p], proj, s);
        T dPdF[81];
        corotated_first_piola_derivative(s, dPdF);
        T nrm = 0;
        for (int k = 0; k < 81; ++k) nrm += dPdF[k] * dPdF[k];
        max_nrm = std::max(max_nrm, std::sqrt(nrm));
    }
    allreduce(&max_nrm, 1, REAL, HOT_COMM_MAX);
    max_cn_tolerance = (T)cfg.cneps * dt * 24 * std::sqrt((T)num_nodes) * dx * dx * max_nrm;
    const bool wd = wide();
    if (wd) wacc.assign(num_nodes, 0.0);
    for_each_particle_colored([&](int g, int i) {
        CorotatedScratch<T> s;
        corotated_update_scratch(TM::identity(), mu[i], lambda[i], proj, s);
        T dPdF[81];
        corotated_first_piola_derivative(s, dPdF);
        T nrm = 0;
        for (int k = 0; k < 81; ++k) nrm += dPdF[k] * dPdF[k];
        nrm = std::sqrt(nrm);
        Spline sp;
        compute_spline(X[i], sp);
        iterate_kernel(sp, g, particle_base_offset[i], [&](const int*, T w, const TV&, Node& gs) {
            if (gs.idx < 0) return;
            if (wd)
                wacc[gs.idx] += (double)(w * mass[i] * nrm);
            else
                nodeCNTol[gs.idx] += w * mass[i] * nrm;
        });
    });
    if (wd)
        for (int n = 0; n < num_nodes; ++n) nodeCNTol[n] = (T)wacc[n];
    allreduce(nodeCNTol.data(), num_nodes, REAL);
    T eps = (T)cfg.cneps;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < num_nodes; ++n) nodeCNTol[n] *= (eps * 24 * dx * dx * dt) / mass_matrix[n];
}

// reference ImplicitSolverObjective::multiply, matrix-free branch (ImplicitSolver.h:741-758):
// b = M x + dt^2 * K x  via MassLumpedInertia::addScaledForceDifferential (Inertia.cpp:45-53) and
// MpmForceBase::addScaledForceDifferential (MpmForceBase.cpp:262-306) ->
// FBasedMpmForceHelper::computeStressDifferential (FBasedMpmForceHelper.cpp:137-160)
template <class T>
void Sim<T>::matfree_multiply(const std::vector<TV>& x, std::vector<TV>& b)
{
    b.assign(num_nodes, TV::zero());
    T scale = -(dt * dt);
    // inertia: df -= scale/dt^2 * m * dx
#pragma omp parallel for schedule(static)
    for (int n = 0; n < num_nodes; ++n) b[n] -= x[n] * ((scale / (dt * dt)) * mass_matrix[n]);
    eval_interpolant_and_gradient(x);
    bool proj = cfg.project != 0;
    std::vector<TM> saved = scratch_stress;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < Np; ++p) {
        CorotatedScratch<T> s;
        corotated_update_scratch(F[p], mu[p], lambda[p], proj, s);
        TM dP = corotated_first_piola_differential(s, scratch_gradV[p] * Fn[p]);
        scratch_stress[p] = (dP * vol[p]) * Fn[p].transpose();
    }
    rasterize_force(scale, b);
    scratch_stress.swap(saved);
}

} // namespace hot_oracle
