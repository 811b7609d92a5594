// hot_adapter.hpp — header-only C++ adapter over the C ABI of libhotmi355x (include/hot_mi355x.h).
//
// It re-exposes the accelerated hot path under the member names of the reference objects that own it, so that
// the reference's solver templates (LBFGS<Objective>, ExtendedNewtonsMethod<Objective>, InexactConjugateGradient,
// MultigridOperator's smoother plug point) can be pointed at it:
//
//   hotmi::Simulation<T>   <->  MultigridSimulation<T,3> / MpmSimulationBase<T,3>
//        sortParticlesAndPolluteGrid()          Lib/MPM/MpmSimulationBase.cpp:1066-1137
//        particlesToGrid()                      Lib/MPM/MpmSimulationBase.cpp:461-533
//        startBackwardEuler(dt)                 Projects/multigrid/MultigridSimulation.h:167-186
//        backwardEulerStep()                    Projects/multigrid/MultigridSimulation.h:188-233  (whole solve on the device)
//        gridToParticles(dt)                    Lib/MPM/MpmSimulationBase.cpp:903-1042
//        advanceOneTimeStep(dt)                 Projects/multigrid/MultigridSimulation.h:235-297
//   hotmi::Objective<T>    <->  ImplicitSolverObjective<Simulation>  (Projects/multigrid/ImplicitSolver.h)
//        updateState / totalEnergy / computeResidual / shouldExitByCN / HinvApproxInit / multiply / precondition / project /
//        lineSearch / recoverSolution / transformResidual / computeStep / innerProduct, with raw-pointer and Vec& overloads
//        (tests/cpp/adapter_lbfgs.cpp drives a two-loop L-BFGS written in the member-call shape of LBFGS.h:300-437 through it)
//   hotmi::smoothFunc      <->  MultigridOperator::regular.smoothFunc (Projects/multigrid/MultigridPreconditioner.h:67-79)
//
// Vectors are "TVStack" (3 x N column-major == xyz interleaved) exactly like the reference's
// Eigen::Matrix<T,3,Dynamic>; pass `stack.data()`.  Errors: the reference asserts/throws (ZIRAN_ASSERT,
// Lib/Ziran/CS/Util/Debug.h:19); here every non-zero hot_status becomes std::runtime_error with hot_last_error().
#pragma once
#include "hot_mi355x.h"
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>

// HOTSettings is a namespace of inline statics on the reference side (Projects/multigrid/Configurations.h:18-42),
// so the copy into hot_config is a macro; field names are identical on both sides.
#define HOTMI_CONFIG_FROM_HOTSETTINGS(cfg)                                                                 \
    do {                                                                                                   \
        (cfg).lsolver = HOTSettings::lsolver, (cfg).Ainv = HOTSettings::Ainv;                              \
        (cfg).smoother = HOTSettings::smoother, (cfg).coarseSolver = HOTSettings::coarseSolver;            \
        (cfg).levelCnt = HOTSettings::levelCnt, (cfg).times = HOTSettings::times;                          \
        (cfg).levelscale = HOTSettings::levelscale, (cfg).omega = HOTSettings::omega;                      \
        (cfg).topomega = HOTSettings::topomega, (cfg).cneps = HOTSettings::cneps;                          \
        (cfg).useCN = HOTSettings::useCN, (cfg).project = HOTSettings::project;                            \
        (cfg).systemBCProject = HOTSettings::systemBCProject, (cfg).linesearch = HOTSettings::linesearch;  \
        (cfg).matrixFree = HOTSettings::matrixFree, (cfg).boundaryType = HOTSettings::boundaryType;        \
        (cfg).useAdaptiveHessian = HOTSettings::useAdaptiveHessian;                                        \
        (cfg).topDownMGS = HOTSettings::topDownMGS;                                                        \
        (cfg).useBaselineMultigrid = HOTSettings::useBaselineMultigrid;                                    \
    } while (0)

namespace hotmi {

inline void check(hot_ctx* c, int rc, const char* what)
{
    if (rc != HOT_OK) throw std::runtime_error(std::string(what) + ": " + (c ? hot_last_error(c) : "no context") + " (status " + std::to_string(rc) + ")");
}

template <class T>
class Simulation {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "T must be float or double");

public:
    hot_ctx* ctx = nullptr;
    hot_config cfg;
    hot_stats stats;

    explicit Simulation(hot_config c)
        : cfg(c)
    {
        cfg.dtype = std::is_same<T, double>::value ? 1 : 0;
        int rc = hot_create(&cfg, &ctx);
        if (rc != HOT_OK) throw std::runtime_error("hot_create failed (no GPU? there is no CPU fallback), status " + std::to_string(rc));
    }
    Simulation(const Simulation&) = delete;
    Simulation& operator=(const Simulation&) = delete;
    ~Simulation() { hot_destroy(ctx); }

    static hot_config defaults()
    {
        hot_config c;
        hot_default_config(&c);
        return c;
    }
    // particles.X / V / mass / C / F / "element measure" / CorotatedIsotropic(mu, lambda)
    void setParticles(int64_t Np, const T* X, const T* V, const T* mass, const T* C, const T* F, const T* vol, const T* mu, const T* lambda, const T* Jp = nullptr)
    {
        check(ctx, hot_set_particles(ctx, Np, X, V, mass, C, F, vol, mu, lambda, Jp), "hot_set_particles");
    }
    void getParticles(T* X, T* V, T* C, T* F, T* mu = nullptr, T* lambda = nullptr, T* Jp = nullptr) { check(ctx, hot_get_particles(ctx, X, V, C, F, mu, lambda, Jp), "hot_get_particles"); }
    void sortParticlesAndPolluteGrid() { check(ctx, hot_sort(ctx), "hot_sort"); }
    void particlesToGrid() { check(ctx, hot_p2g(ctx), "hot_p2g"); }
    int numNodes()
    {
        int32_t nn = 0;
        check(ctx, hot_get_counts(ctx, nullptr, nullptr, nullptr, &nn), "hot_get_counts");
        return nn;
    }
    // MpmGrid::iterateGrid replacement for host-side collision queries: id2coord[3*id + d] is the integer node
    // coordinate of DOF `id` (world position = coord * dx); mass / v may be null.
    void gridNodes(int32_t* id2coord, T* mass = nullptr, T* v = nullptr) { check(ctx, hot_get_grid(ctx, id2coord, mass, v), "hot_get_grid"); }
    // collision_nodes as produced by buildInitialDvAndVnForNewton's collision query (host geometry code)
    void setCollisionNodes(int Nc, const int32_t* node_id, const T* P, const T* R, const T* Rinv, const uint8_t* shouldRotate, const T* dv = nullptr)
    {
        check(ctx, hot_set_bc(ctx, Nc, node_id, P, R, Rinv, shouldRotate, dv), "hot_set_bc");
    }
    // AnalyticCollisionObject list (half spaces, spheres, tori, sticky boxes / capped cylinders; STICKY / SLIP / SEPARATE) evaluated per node on the device
    void setCollisionObjects(int n, const hot_collision_object* objects) { check(ctx, hot_set_collision_objects(ctx, n, objects), "hot_set_collision_objects"); }
    void startBackwardEuler(double dt) { check(ctx, hot_begin_step(ctx, dt), "hot_begin_step"); }
    void getDv(T* dv) { check(ctx, hot_get_dv(ctx, dv), "hot_get_dv"); } // simulation.dv (3 x numNodes)
    void backwardEulerStep() { check(ctx, hot_solve(ctx, &stats), "hot_solve"); }
    void gridToParticles(double dt)
    {
        int32_t flags = 0;
        check(ctx, hot_g2p(ctx, dt, &flags), "hot_g2p");
        faster_than_grid_cell = flags & 1, faster_than_half_grid_cell = (flags & 2) != 0;
    }
    void advanceOneTimeStep(double dt) { check(ctx, hot_advance(ctx, dt, &stats), "hot_advance"); }
    // MpmSimulationBase::calculateDt (Lib/MPM/MpmSimulationBase.cpp:789-814); max_dt = step.max_dt
    double calculateDt(double max_dt)
    {
        double dt = 0;
        check(ctx, hot_calculate_dt(ctx, max_dt, &dt, nullptr, nullptr, nullptr), "hot_calculate_dt");
        return dt;
    }
    // SimulationBase::advanceOneFrame (Lib/Ziran/Sim/SimulationBase.h:291-327): returns the number of substeps
    int advanceOneFrame(double frame_dt, double min_dt = 1e-6, double max_dt = -1)
    {
        int32_t n = 0, its = 0;
        check(ctx, hot_advance_frame(ctx, frame_dt, min_dt, max_dt > 0 ? max_dt : frame_dt, &n, &its, &stats), "hot_advance_frame");
        return n;
    }
    // MpmSimulationBase::writeState's particle .bgeo (Lib/MPM/MpmSimulationBase.cpp:186-224) and SimulationBase's restart write / read
    // (Lib/Ziran/Sim/SimulationBase.h:215-263)
    void writePartio(const std::string& path) { check(ctx, hot_write_partio(ctx, path.c_str()), "hot_write_partio"); }
    void writeRestart(const std::string& path) { check(ctx, hot_write_restart(ctx, path.c_str()), "hot_write_restart"); }
    void readRestart(const std::string& path) { check(ctx, hot_read_restart(ctx, path.c_str()), "hot_read_restart"); }
    bool faster_than_grid_cell = false, faster_than_half_grid_cell = false;
};

// The objective concept LBFGS<Objective> / ExtendedNewtonsMethod<Objective> / InexactConjugateGradient / Minres are templated on
// (Projects/multigrid/ImplicitSolver.h; calls made by Lib/Ziran/Math/Nonlinear/LBFGS.h:300-437 and ExtendedNewtonsMethod.h:39-66).
// `Vec` is whatever the solver template uses for TVStack — any type with data() / size() over 3 x N column-major scalars
// (Eigen::Matrix<T,3,Dynamic>, std::vector<T>, ...).  Raw-pointer overloads are kept for C-style callers.
//
// Two reference behaviours are reproduced on purpose, because the solver templates rely on them (DESIGN.md "reference quirks"):
//   * `updated`: after lineSearch the objective's state IS the accepted point, so updateState / computeResidual return
//     immediately for the rest of the step (ImplicitSolver.h:132,241, resetLSFlag at startBackwardEuler);
//   * aliasing: the reference solves in place on simulation.dv, and lineSearch moves simulation.dv to the accepted point
//     (moveNodes, MpmSimulationBase.cpp:736-747), i.e. the solver's x changes under it.  Here x is the caller's memory, so
//     updateState remembers where x lives and lineSearch writes the moved dv back to it.
template <class T>
class Objective {
public:
    using Scalar = T;
    Simulation<T>& simulation;
    bool matrix_free = false;
    bool updated = false;
    T Ek = 0;
    explicit Objective(Simulation<T>& s)
        : simulation(s) {}
    hot_ctx* c() const { return simulation.ctx; }
    int64_t numNodes() const { return simulation.numNodes(); }
    void resetLSFlag() { updated = false, x_alias = nullptr; } // startBackwardEuler (MultigridSimulation.h:167-186); the solver's x of the previous step is forgotten
    // The solver's iterate x IS simulation.dv in the reference (LBFGS / Newton are handed `simulation.dv` by reference), so lineSearch moves it.
    // Here x lives in the caller's memory: name it once per solve with setX (mutable, must stay alive and un-reallocated until the solve
    // returns).  Without setX, lineSearch falls back to the pointer of the last updateState call — which then must be that same vector.
    void setX(T* x) { x_alias = x, x_explicit = true; }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    void setX(Vec& x) { setX(x.data()); }

    // ---- raw pointers
    void updateState(const T* dv)
    {
        if (!x_explicit) x_alias = const_cast<T*>(dv); // legacy aliasing rule (see setX)
        if (updated) return;
        double e = 0;
        check(c(), hot_update_state(c(), dv, &e), "hot_update_state");
        Ek = (T)e;
    }
    T totalEnergy() const { return Ek; }
    void computeResidual(T* residual)
    {
        if (updated) return;
        check(c(), hot_residual(c(), residual), "hot_residual");
    }
    bool shouldExitByCN(const T* residual)
    {
        int32_t e = 0;
        check(c(), hot_should_exit(c(), residual, &e, nullptr), "hot_should_exit");
        return e != 0;
    }
    void HinvApproxInit()
    {
        check(c(), hot_build_hessian(c()), "hot_build_hessian");
        check(c(), hot_build_mg(c()), "hot_build_mg");
    }
    void multiply(const T* x, T* b) const { check(c(), matrix_free ? hot_matfree_multiply(c(), x, b) : hot_spmv(c(), 0, x, b), "multiply"); }
    void precondition(const T* in, T* out) const { check(c(), hot_vcycle(c(), in, out), "hot_vcycle"); }
    void project(T* v) const { check(c(), hot_project(c(), v), "hot_project"); }
    T lineSearch(T* ddv, T* residual, T alpha)
    {
        double a = 0;
        check(c(), hot_line_search(c(), ddv, residual, (double)alpha, &a), "hot_line_search");
        updated = true;
        if (x_alias) check(c(), hot_get_dv(c(), x_alias), "hot_get_dv"); // the solver's x is simulation.dv in the reference
        return (T)a;
    }
    void recoverSolution(T* v) const { check(c(), hot_recover_solution(c(), v), "hot_recover_solution"); }
    void transformResidual(T* v) const { check(c(), hot_transform_residual(c(), v), "hot_transform_residual"); }
    // computeStep(step, residual, relative_tolerance): the tolerances are derived inside from the residual as in :355-432
    void computeStep(T* step, const T* residual, T /*relative_tolerance*/ = 0) { check(c(), hot_compute_step(c(), residual, step), "hot_compute_step"); }
    // evaluatePerNodeCNTolerance (ImplicitSolver.h:667-696): per-node characteristic-norm tolerance
    void cnTolerance(T* node_tol) const { check(c(), hot_cn_tolerance(c(), node_tol), "hot_cn_tolerance"); }
    T innerProduct(const T* a, const T* b, int64_t numNodes) const
    {
        T s = 0;
        for (int64_t i = 0; i < 3 * numNodes; ++i) s += a[i] * b[i];
        return s;
    }

    // ---- Vec& overloads: the member-call shape of the reference's solver templates
    template <class Vec, class = decltype(std::declval<const Vec&>().data())>
    void updateState(const Vec& x) { updateState(x.data()); }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    void computeResidual(Vec& r) { computeResidual(r.data()); }
    template <class Vec, class = decltype(std::declval<const Vec&>().data())>
    bool shouldExitByCN(const Vec& r) { return shouldExitByCN(r.data()); }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    void multiply(const Vec& x, Vec& b) const { multiply(x.data(), b.data()); }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    void precondition(const Vec& in, Vec& out) const { precondition(in.data(), out.data()); }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    void project(Vec& v) const { project(v.data()); }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    T lineSearch(Vec& ddv, Vec& residual, T alpha) { return lineSearch(ddv.data(), residual.data(), alpha); }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    void recoverSolution(Vec& v) const { recoverSolution(v.data()); }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    void transformResidual(Vec& v) const { transformResidual(v.data()); }
    template <class Vec, class = decltype(std::declval<Vec&>().data())>
    void computeStep(Vec& step, const Vec& residual, T relative_tolerance = 0) { computeStep(step.data(), residual.data(), relative_tolerance); }
    template <class Vec, class = decltype(std::declval<const Vec&>().data())>
    T innerProduct(const Vec& a, const Vec& b) const { return innerProduct(a.data(), b.data(), (int64_t)(a.size() / 3)); }

private:
    T* x_alias = nullptr;
    bool x_explicit = false;
};

// void (*smoothFunc)(TVStack& u, TVStack& r, TVStack& du, TVStack& dAu, MPMSpMat& A, int iterations, T tolerance):
// `A` is replaced by (context, level); du / dAu are library-internal work vectors.
template <class T>
inline void smoothFunc(hot_ctx* ctx, int level, int kind /* -smoother numbering */, T* u, T* r, int iterations, T tolerance, const T* initialResidual = nullptr)
{
    check(ctx, hot_smooth(ctx, level, kind, iterations, (double)tolerance, u, r, initialResidual), "hot_smooth");
}

} // namespace hotmi
