// The fine-grained boundary, exercised: a two-loop L-BFGS written here in the member-call shape of the reference's
// LBFGS<Objective>::solve (Lib/Ziran/Math/Nonlinear/LBFGS.h:300-437: updateState, computeResidual, shouldExitByCN, HinvApproxInit,
// precondition, project, lineSearch, recoverSolution, transformResidual on `objective`, Vec& arguments) is instantiated with
// hotmi::Objective<double> and must reproduce what the library's own device-side solve (hot_solve) computes for the same
// iteration budget.  Also calls updateState -> computeResidual -> precondition -> project once by hand.
// Exit codes: 0 ok, 42 no GPU (the constructor throws: there is no CPU fallback), anything else = mismatch.
#include "hot_adapter.hpp"
#include <array>
#include <cmath>
#include <cstdio>
#include <vector>

struct Vec { // TVStack stand-in: 3 x N column-major
    std::vector<double> v;
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
    size_t size() const { return v.size(); }
    void resizeLike(const Vec& o) { v.assign(o.v.size(), 0.0); }
    Vec& operator+=(const Vec& o)
    {
        for (size_t i = 0; i < v.size(); ++i) v[i] += o.v[i];
        return *this;
    }
    Vec& operator-=(const Vec& o)
    {
        for (size_t i = 0; i < v.size(); ++i) v[i] -= o.v[i];
        return *this;
    }
    void axpy(double a, const Vec& o)
    {
        for (size_t i = 0; i < v.size(); ++i) v[i] += a * o.v[i];
    }
};
static double dotProduct(const Vec& a, const Vec& b)
{
    double s = 0;
    for (size_t i = 0; i < a.v.size(); ++i) s += a.v[i] * b.v[i];
    return s;
}

template <class Objective>
struct LocalLBFGS {
    Objective& objective;
    int max_iterations = 5;
    int iterations = 0;
    explicit LocalLBFGS(Objective& o)
        : objective(o) {}
    bool solve(Vec& x, bool useLinesearch)
    {
        constexpr int historySize = 8;
        Vec residual;
        residual.resizeLike(x);
        objective.updateState(x);
        objective.computeResidual(residual);
        std::vector<Vec> dxx(1), dg(1); // oldest first; back() is the working slot (RingBuffer, LBFGS.h:23-69)
        std::vector<double> dgTdx(1, 0.0);
        std::array<double, historySize + 1> ksi{};
        dxx[0].resizeLike(x), dg[0].resizeLike(x);
        auto push = [&]() {
            dxx.emplace_back(), dg.emplace_back(), dgTdx.push_back(0.0);
            dxx.back().resizeLike(x), dg.back().resizeLike(x);
            if ((int)dxx.size() > historySize + 1) dxx.erase(dxx.begin()), dg.erase(dg.begin()), dgTdx.erase(dgTdx.begin());
        };
        for (int it = 0; it < max_iterations; ++it) {
            iterations = it;
            if (objective.shouldExitByCN(residual)) return true;
            if (it == 0) {
                objective.HinvApproxInit();
                dxx.resize(1), dg.resize(1), dgTdx.resize(1);
            }
            dg.back() = residual;
            for (int i = (int)dxx.size() - 2; i >= 0; --i) {
                ksi[i] = dotProduct(dxx[i], residual) * dgTdx[i];
                residual.axpy(-ksi[i], dg[i]);
            }
            objective.precondition(residual, dxx.back());
            objective.project(dxx.back());
            for (int i = 0; i < (int)dxx.size() - 1; ++i) dxx.back().axpy(ksi[i] - dotProduct(dg[i], dxx.back()) * dgTdx[i], dxx[i]);
            if (useLinesearch) objective.lineSearch(dxx.back(), residual, 1.0);
            objective.recoverSolution(dxx.back());
            x += dxx.back();
            objective.transformResidual(dxx.back());
            objective.updateState(x);
            objective.computeResidual(residual);
            dg.back() -= residual;
            dgTdx.back() = 1.0 / dotProduct(dg.back(), dxx.back());
            if (dgTdx.back() <= 0.0) dxx.pop_back(), dg.pop_back(), dgTdx.pop_back();
            push();
        }
        iterations = max_iterations;
        return false;
    }
};

static void cloud(const hot_config& cfg, std::vector<double>& X, std::vector<double>& V, std::vector<double>& m, std::vector<double>& vol, std::vector<double>& mu, std::vector<double>& la)
{
    const int n = 6, ppc = 8;
    unsigned s = 12345u;
    auto rnd = [&]() { return (s = s * 1664525u + 1013904223u, (double)(s >> 8) / 16777216.0); };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k)
                for (int q = 0; q < ppc; ++q) {
                    double o[3] = { 0.15 + 0.5 * (q & 1) + 0.2 * rnd(), 0.15 + 0.5 * ((q >> 1) & 1) + 0.2 * rnd(), 0.15 + 0.5 * (q >> 2) + 0.2 * rnd() };
                    X.insert(X.end(), { 5 + (i + o[0]) * cfg.dx, 5 + (j + o[1]) * cfg.dx, 5 + (k + o[2]) * cfg.dx });
                    V.insert(V.end(), { 0.3 * (rnd() - 0.5), -0.2 + 0.3 * (rnd() - 0.5), 0.3 * (rnd() - 0.5) });
                    m.push_back(2000 * cfg.dx * cfg.dx * cfg.dx / ppc), vol.push_back(cfg.dx * cfg.dx * cfg.dx / ppc), mu.push_back(19230.77), la.push_back(28846.15);
                }
}

int main()
{
    hot_config cfg = hotmi::Simulation<double>::defaults();
    cfg.levelCnt = 2, cfg.max_iterations = 5, cfg.cneps = 1e-9;
    try {
        std::vector<double> X, V, m, vol, mu, la;
        cloud(cfg, X, V, m, vol, mu, la);
        const double origin[3] = { 0, 5.0 + 0.5 * cfg.dx, 0 }, normal[3] = { 0, 1, 0 };
        auto prepare = [&](hotmi::Simulation<double>& sim) {
            sim.setParticles((int64_t)m.size(), X.data(), V.data(), m.data(), nullptr, nullptr, vol.data(), mu.data(), la.data());
            hotmi::check(sim.ctx, hot_set_sticky_halfspaces(sim.ctx, 1, origin, normal), "hot_set_sticky_halfspaces");
            sim.sortParticlesAndPolluteGrid();
            sim.particlesToGrid();
            sim.startBackwardEuler(1.0 / 24);
        };
        // A: the library's own device-side L-BFGS
        hotmi::Simulation<double> simA(cfg);
        prepare(simA);
        simA.backwardEulerStep();
        Vec dvA;
        dvA.v.resize(3 * (size_t)simA.numNodes());
        simA.getDv(dvA.data());
        // B: the same iterations driven from the host through the objective concept
        hotmi::Simulation<double> simB(cfg);
        prepare(simB);
        hotmi::Objective<double> obj(simB);
        obj.resetLSFlag();
        Vec x;
        x.v.resize(3 * (size_t)simB.numNodes());
        simB.getDv(x.data());
        obj.setX(x); // the solver's iterate: lineSearch moves it, as it moves simulation.dv in the reference
        {   // by hand, once: updateState -> computeResidual -> HinvApproxInit -> precondition -> project (state unchanged: x is dv)
            Vec r, z;
            r.resizeLike(x), z.resizeLike(x);
            obj.updateState(x);
            obj.computeResidual(r);
            obj.HinvApproxInit();
            obj.precondition(r, z);
            obj.project(z);
            if (!(obj.innerProduct(r, z) > 0)) return std::printf("V-cycle direction is not a descent direction\n"), 3;
        }
        LocalLBFGS<hotmi::Objective<double>> lbfgs(obj);
        lbfgs.max_iterations = cfg.max_iterations;
        lbfgs.solve(x, cfg.linesearch != 0);
        if (x.size() != dvA.size()) return std::printf("node counts differ\n"), 4;
        double err = 0, mag = 0;
        for (size_t i = 0; i < x.size(); ++i) err = std::fmax(err, std::fabs(x.v[i] - dvA.v[i])), mag = std::fmax(mag, std::fabs(dvA.v[i]));
        std::printf("adapter L-BFGS vs hot_solve: %d iterations each, max |ddv| / max |dv| = %.3e\n", simA.stats.iterations, err / mag);
        if (simA.stats.iterations != lbfgs.iterations) return 5;
        return err <= 1e-9 * mag ? 0 : 6;
    }
    catch (const std::exception& e) {
        std::printf("adapter threw: %s\n", e.what());
        return 42;
    }
}
