// libhotmi355x — nonlinear solvers and the time-step driver (host control flow, device data).
//
//   lbfgs_solve     LBFGS::solve (reference Lib/Ziran/Math/Nonlinear/LBFGS.h:300-437): history 8 ring buffer, two-loop
//                   recursion around the V-cycle, curvature pairs dropped when y^T s <= 0, Hessian + hierarchy rebuilt at
//                   iteration 0 (or every 16 with useAdaptiveHessian).  The two-loop scalars live on the device
//                   (k_lbfgs_scalar), so the <= 16 dots + 16 axpys per iteration run without host round trips.
//   line_search     ImplicitSolverObjective::lineSearch (Projects/multigrid/ImplicitSolver.h:312-333)
//   should_exit     shouldExitByCN (:174-211) / computeNorm (:158-171)
//   newton_solve    ExtendedNewtonsMethod::solve (Lib/Ziran/Math/Nonlinear/ExtendedNewtonsMethod.h:39-66) + computeStep
//                   (ImplicitSolver.h:355-432) + InexactConjugateGradient::solve (Lib/Ziran/Math/Linear/InexactConjugateGradient.h:49-103)
//   advance         MultigridSimulation::advanceOneTimeStep (Projects/multigrid/MultigridSimulation.h:235-297)
//
// Reference quirk kept on purpose (DESIGN.md "reference quirks"): with --linesearch, lineSearch's updateState(dvnew)
// aliases the solver's x through moveNodes (MpmSimulationBase.cpp:736-747) and `updated` stays true for the rest of
// the step, so the caller's `x += step` (LBFGS.h:413, ExtendedNewtonsMethod.h:61) lands on top of an x that already
// contains the step: the dv handed to G2P is (last accepted iterate + last accepted step).
#include "hot_impl.h"
#include "hot_svd.h"
#include "hot_collision.h"
#include <cmath>

namespace hot {

template <class T>
__global__ __launch_bounds__(256) void k_scaled_norm(const T* __restrict__ r, const T* __restrict__ tol, int nn, int useCN, double* out, GridRed gr, const uint8_t* __restrict__ mask)
{
    __shared__ double red[4];
    double s = 0;
    const int stride = gridDim.x * 256;
    for (int n0 = blockIdx.x * 256 + threadIdx.x; n0 < nn; n0 += 2 * stride) { // two nodes per trip in flight
        const int n1 = n0 + stride < nn ? n0 + stride : n0;
        if (mask) { // sharded, halo mode: the rows this rank owns (the other entries of r are not maintained here)
            double q = 0;
            for (int u = 0; u < 2; ++u) {
                const int n = u ? n1 : n0;
                if ((u && n1 == n0) || !mask[n]) continue;
                const T a = r[3 * n], b = r[3 * n + 1], c = r[3 * n + 2], t = useCN ? tol[n] : (T)1;
                T qq = a * a + b * b + c * c;
                if (useCN) qq = qq / (t * t);
                q += (double)qq;
            }
            s += q;
            continue;
        }
        const T a0 = r[3 * n0], b0 = r[3 * n0 + 1], c0 = r[3 * n0 + 2], t0 = useCN ? tol[n0] : (T)1;
        const T a1 = r[3 * n1], b1 = r[3 * n1 + 1], c1 = r[3 * n1 + 2], t1 = useCN ? tol[n1] : (T)1;
        T q = a0 * a0 + b0 * b0 + c0 * c0;
        if (useCN) q = q / (t0 * t0);
        s += (double)q;
        if (n1 != n0) {
            T q1 = a1 * a1 + b1 * b1 + c1 * c1;
            if (useCN) q1 = q1 / (t1 * t1);
            s += (double)q1;
        }
    }
    double t = block_sum_256<double>(s, red);
    grid_sum_store(t, 0.0, 1, gr, out, nullptr, red);
}
// out = dv0 + alpha * ddv
template <class T>
__global__ void k_combine(size_t n, const T* __restrict__ a, T alpha, const T* __restrict__ b, T* out, T* out2)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = out2[i] = a[i] + b[i] * alpha;
}
template <class T>
__global__ void k_scal(size_t n, T alpha, T* x)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= alpha;
}
// two-loop scalars: what 0: ksi[i] = tmp * dgTdx[i] ; what 1: coef = ksi[i] - tmp * dgTdx[i]
__global__ void k_lbfgs_scalar(double* s, int i, int what)
{
    if (what == 0)
        s[50 + i] = s[70] * s[60 + i];
    else
        s[80] = s[50 + i] - s[70] * s[60 + i];
}

// ---- L-BFGS two-loop recursion (LBFGS.h:359-392) in five launches per nonlinear iteration, whatever the history length.
// The recursion is sequential in the reference: ksi_i = rho_i <dx_i, q>, q -= ksi_i dg_i for the newest pair first, then the second
// loop oldest first; written like that it costs one dependent launch (or one dependent all-reduce, sharded) per stored pair and loop.
// Every inner product it needs is an inner product of the CURRENT vector with a history vector, and the current vector is the loop's
// start vector minus a combination of history vectors, so with the Gram matrix M[i][j] = <dx_i, dg_j> of the stored pairs (two new
// rows / columns per iteration, k_lbfgs_pair) the recursion runs on scalars:
//   first loop   b_i = <dx_i, r>  (one batch of m dots),   ksi_i = rho_i (b_i - sum_{j>i} ksi_j M[i][j]),   q = r - sum ksi_i dg_i
//   second loop  e_i = <dg_i, z0> (one batch of m dots),   c_i = ksi_i - rho_i (e_i + sum_{j<i} c_j M[j][i]),   z = z0 + sum c_i dx_i
// The same recursion in the same order (the vector updates are applied element-wise in the reference's order); what changes is the
// rounding of the inner products (a difference of products instead of a product with a difference), far below the parity tolerance.
// Sharded runs (hot_set_comm) all-reduce each batch once: three small all-reduces per iteration instead of 2 m + 1.
constexpr int LB_MAXV = 17; // 2 * historySize + 1 dots in the largest batch
constexpr int LB_B = 512, LB_E = 536, LB_G = 560, LB_M = 600; // dscal slots: first-loop dots, second-loop dots, new-pair dots, Gram matrix (9 x 9 physical slots)
template <class T>
struct LbVecs {
    const T* dx[8]; // stored pairs, oldest first
    const T* dg[8];
    int ph[8]; // their physical history slots
    int m;
};
// out[k] = <a_k, y> for the m stored pairs (a = dx: first loop, a = dg: second loop); mask: rows this rank owns (sharded), or null
template <class T>
__global__ __launch_bounds__(256) void k_lbfgs_dots(size_t n, LbVecs<T> hv, int use_dg, const T* __restrict__ y, double* out, GridRed gr, const uint8_t* __restrict__ mask)
{
    __shared__ double red[4 * 8];
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (mask && !mask[i / 3]) continue;
        const T yv = y[i];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < hv.m) acc[k] += (double)(use_dg ? hv.dg[k][i] : hv.dx[k][i]) * (double)yv; // products in double also in the fp32 build: the Gram recursion subtracts them
    }
    const int m = hv.m;
    block_sum_256_n<8>(acc, [m](int k) { return k < m; }, red);
    grid_sum_store_n<8>(acc, [m](int k) { return k < m ? k : -1; }, m, gr, out, red);
}
// mode 0: q = r - sum ksi_i dg_i (newest pair first), ksi stored at s[50 + ph]; optional keep[i] = r[i] (the working pair's dg starts as
// the old residual).  mode 1: z = z0 + sum c_i dx_i (oldest first).  Every workgroup runs the scalar recursion itself (m <= 8).
template <class T>
__global__ __launch_bounds__(256) void k_lbfgs_apply(size_t n, double* s, LbVecs<T> hv, int mode, T* __restrict__ y, T* __restrict__ keep, const uint8_t* __restrict__ mask)
{
    __shared__ double coef[8];
    if (threadIdx.x == 0) {
        const int m = hv.m;
        double t[8];
        if (mode == 0) {
            for (int i = m - 1; i >= 0; --i) {
                double v = s[LB_B + i];
                for (int j = i + 1; j < m; ++j) v -= t[j] * s[LB_M + 9 * hv.ph[i] + hv.ph[j]];
                t[i] = v * s[60 + hv.ph[i]];
                coef[i] = -t[i];
                if (blockIdx.x == 0) s[50 + hv.ph[i]] = t[i];
            }
        }
        else {
            for (int i = 0; i < m; ++i) {
                double v = s[LB_E + i];
                for (int j = 0; j < i; ++j) v += t[j] * s[LB_M + 9 * hv.ph[j] + hv.ph[i]];
                t[i] = s[50 + hv.ph[i]] - s[60 + hv.ph[i]] * v;
                coef[i] = t[i];
            }
        }
    }
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (mask && !mask[i / 3]) continue;
        T v = y[i];
        if (keep) keep[i] = v;
        if (mode == 0) {
#pragma unroll
            for (int k = 7; k >= 0; --k)
                if (k < hv.m) v = v + (T)coef[k] * hv.dg[k][i];
        }
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < hv.m) v = v + (T)coef[k] * hv.dx[k][i];
        }
        y[i] = v;
    }
}
// The new curvature pair of slot wk: dg_wk -= r_new, then out[0] = <dx_wk, dg_wk>, out[1 + j] = <dx_wk, dg_j>, out[1 + m + j] = <dx_j, dg_wk>
template <class T>
__global__ __launch_bounds__(256) void k_lbfgs_pair(size_t n, LbVecs<T> hv, const T* __restrict__ dxw, T* __restrict__ dgw, const T* __restrict__ rnew, double* out, GridRed gr, const uint8_t* __restrict__ mask)
{
    __shared__ double red[4 * LB_MAXV];
    double acc[LB_MAXV];
#pragma unroll
    for (int k = 0; k < LB_MAXV; ++k) acc[k] = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (mask && !mask[i / 3]) continue;
        const T g = dgw[i] - rnew[i], x = dxw[i];
        dgw[i] = g;
        acc[0] += (double)g * (double)x;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < hv.m) acc[1 + k] += (double)x * (double)hv.dg[k][i];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < hv.m) acc[9 + k] += (double)hv.dx[k][i] * (double)g;
    }
    // acc: [0] | [1, 8] | [9, 16], of which m each are in use; compact order of the deposits: [0] | [1, m] | [m + 1, 2 m]
    const int m = hv.m;
    block_sum_256_n<LB_MAXV>(acc, [m](int k) { return k == 0 || ((k - 1) & 7) < m; }, red);
    grid_sum_store_n<LB_MAXV>(acc, [m](int k) { return k == 0 ? 0 : (((k - 1) & 7) < m ? (k <= 8 ? k : k - 8 + m) : -1); }, 2 * m + 1, gr, out, red);
}
// files the (all-reduced, when sharded) dots of k_lbfgs_pair: curvature -> s[71] and the pinned host slot, rho -> s[60 + wk] in the
// arithmetic of the host code it replaces ((T)1 / (T)d, LBFGS.h:424-434), Gram rows / columns of slot wk
template <class T>
__global__ void k_lbfgs_file(double* s, LbVecs<T> hv, int wk, double* host_curv)
{
    const double* g = s + LB_G;
    s[71] = g[0];
    *host_curv = g[0];
    s[60 + wk] = (double)((T)1 / (T)g[0]);
    for (int j = 0; j < hv.m; ++j) s[LB_M + 9 * wk + hv.ph[j]] = g[1 + j], s[LB_M + 9 * hv.ph[j] + wk] = g[1 + hv.m + j];
}
// rho of a new curvature pair: s[dst] = 1 / (y's) in the arithmetic of the host code it replaces ((T)1 / (T)d, LBFGS.h:424-434)
template <class T>
__global__ void k_lbfgs_rho(double* s, int src, int dst)
{
    s[dst] = (double)((T)1 / (T)s[src]);
}

template <class T>
bool Ctx<T>::should_exit(const T* r)
{
    if (Nn == 0) return true;
    {
        const int grid = std::min(div_up(Nn, 512), 1024);
        HOT_LAUNCH(this, "exit_norm", k_scaled_norm<T>, grid, 256, 0, r, cnTol.p, Nn, cfg.useCN, dscal.p + 90, gred(grid, hscal + 90, true), vmask);
    }
    wait_ticket();
    if (vmask) c_allreduce(hscal + 90, 1, HOT_COMM_F64, HOT_COMM_SUM, false); // partitioned vectors: the ranks' sums over their own rows
    double v = hscal[90];
    HOT_CHECK(v == v, HOT_ERR_NUMERIC, "NaN in the residual norm");
    if (!cfg.useCN) {
        double res = std::sqrt(v);
        stats.final_scaled_residual = res;
        return res < cfg.cneps;
    }
    stats.final_scaled_residual = std::sqrt(v / Nn);
    return (T)v < (T)Nn;
}

template <class T>
T Ctx<T>::line_search(T* ddv, T* residual_out, T alpha)
{
    size_t n3 = 3 * (size_t)Nn;
    T* dvnew = work3.p;
    transform_dev(ddv, true); // recoverSolution
    // (an energy-only trial sums psi from the singular values, mu sum (sigma_i - 1)^2; the full pass — the reference's formula, mu |F - R|^2 — emits
    // that sum as well, Ek_sigma, so that a trial is compared with a base of its own rounding: as the step shrinks the trial energy tends to
    // Ek_sigma of the base point, not to Ek, and a positive round-off gap between the two would reject every halving)
    const double Ek0 = Ek, Ek0_sigma = Ek_sigma;
    int guard = 0;
    // Energy-only trials (hot_config.ls_energy_only): a rejected trial pays for singular values and the sum, not for U, V, the stress and
    // 18 stores per particle; the accepted point then gets the full pass.  0 = adaptive: the first trial is a full pass unless the previous
    // search had to halve (one trial per iteration, the common case at moderate stiffness, costs what it did), every trial after a rejection
    // is energy-only; 1 = never; 2 = always.
    const int eo_mode = cfg.ls_energy_only;
    bool eo = eo_mode == 2 || (eo_mode == 0 && ls_prev_trials > 1), last_eo = false;
    int trials = 0;
    // Energy-only trials in batches (trial_batch, force.hip; sharded runs too: two halo exchanges and one all-reduce per batch instead of one each per trial): the energies of alpha, alpha / 2, ... from ONE pass (equal to the passes they
    // replace up to the rounding of the trial F), looked at in order — the search accepts what it would have accepted and counts the trials it would have run.  (A/B build: HOT_LS_NO_BATCH = 1 runs them one by one.)
    const bool batched = !ab_flag("HOT_LS_NO_BATCH");
    bool last_from_batch = false;
    int batches = 0;
    do {
        if (eo && batched && guard + 2 <= 59) {
            // a batch costs a fixed part (C4: 0.63 ms, the chain of a pass with two gathers) + 0.15 ms per trial: as many as the previous search needed,
            // (2 / 4 / 8 / 16), eights once a batch has failed
            const int pv = ls_prev_trials;
            int K = batches == 0 ? (pv <= 2 ? 2 : (pv <= 4 ? 4 : (pv <= 8 ? 8 : 16))) : 8;
            ++batches;
            while (guard + K > 59) K >>= 1; // (the search gives up after 60 trials)
            double Eb[16];
            trial_batch(ddv, alpha, K, Eb);
            int k = 0;
            for (;; ++k) {
                Ek = Eb[k];
                ++trials, stats.linesearch_trials++;
                if (ab_flag("HOT_DEBUG")) fprintf(stderr, "[hot]   linesearch alpha=%g Ek=%.12e Ek0=%.12e (batch of %d)\n", (double)alpha, Ek, Ek0, K);
                alpha *= (T)0.5;
                if (Ek <= Ek0_sigma || k == K - 1) break;
                ++guard;
            }
            last_eo = true, last_from_batch = true;
            continue; // (the loop test below: rejected -> ++guard and the next batch)
        }
        last_from_batch = false;
        HOT_LAUNCH(this, "linesearch_combine", k_combine<T>, div_up(n3, 256), 256, 0, n3, dv0.p, alpha, ddv, dvnew, dv.p); // the trial point, also as moveNodes' dv
        Ek = state_pass(dv.p, false, eo); // a trial needs the energy only; the force is rasterised once, at the accepted point
        last_eo = eo;
        if (eo_mode != 1) eo = true;
        ++trials;
        stats.linesearch_trials++;
        alpha *= (T)0.5;
        if (ab_flag("HOT_DEBUG")) fprintf(stderr, "[hot]   linesearch alpha=%g Ek=%.12e Ek0=%.12e\n", (double)alpha * 2, Ek, Ek0);
        // `Ek > Ek0` in the reference (ImplicitSolver.h:325), written so that a NaN energy — an exploded L-BFGS direction in float puts
        // the trial point outside anything representable — is a rejection like any other and the step is halved; the reference would
        // end the search there with NaN accepted.  Identical for every finite energy.
    } while (!(Ek <= (last_eo ? Ek0_sigma : Ek0)) && ++guard < 60);
    if (last_from_batch) // a batch never wrote its trial points: the accepted one (after 60 rejections: the last), as the one-by-one search leaves it in dv and work3
        HOT_LAUNCH(this, "linesearch_combine", k_combine<T>, div_up(n3, 256), 256, 0, n3, dv0.p, alpha * 2, ddv, dvnew, dv.p);
    if (!(Ek == Ek)) {
        // sixty halvings and still no number: is the search direction itself non-finite?
        double dd = dot_host(n3, ddv, ddv), d0 = dot_host(n3, dv0.p, dv0.p);
        char msg[256];
        snprintf(msg, sizeof(msg), "NaN energy in line search (iteration %d, after %d halvings, |direction|^2 %g, |dv|^2 %g, E0 %g)", stats.iterations, guard, dd, d0, Ek0);
        HOT_CHECK(false, HOT_ERR_NUMERIC, msg);
    }
    alpha *= 2;
    ls_prev_trials = trials;
    if (last_eo) Ek = state_pass(dv.p, false, false); // the accepted point: trial F, stresses and the energy the next search compares with
    HOT_LAUNCH(this, "scal", k_scal<T>, div_up(n3, 256), 256, 0, n3, alpha, ddv);
    transform_dev(ddv, false); // transformResidual
    force_pass(); // the stresses of the last (accepted) trial are still in place
    residual_dev(residual_out);
    updated = true;
    std::swap(dv0.p, work3.p), std::swap(dv0.cap, work3.cap); // dv0 := the accepted point (work3 holds it: dvnew), by exchanging the buffers
    return alpha;
}

template <class T>
bool Ctx<T>::lbfgs_solve()
{
    constexpr int historySize = 8;
    size_t n3 = 3 * (size_t)Nn;
    T* x = dv.p;
    T* residual = work2.p;
    for (int k = 0; k < historySize + 1; ++k) hist_dx[k].reserve(n3, 1.25), hist_dg[k].reserve(n3, 1.25);
    if (!updated) {
        Ek = state_pass(x, true);
        residual_dev(residual);
    }
    std::vector<int> order; // physical slots, oldest first; back() is the working slot
    std::vector<int> freeSlots;
    for (int k = historySize; k >= 0; --k) freeSlots.push_back(k);
    auto push_back = [&]() {
        if ((int)order.size() == historySize + 1) {
            freeSlots.push_back(order.front());
            order.erase(order.begin());
        }
        order.push_back(freeSlots.back());
        freeSlots.pop_back();
    };
    auto pop_back = [&]() {
        freeSlots.push_back(order.back());
        order.pop_back();
    };
    push_back();
    double* s = dscal.p;
    // The curvature y's of a new pair is reduced on the device (1 / y's goes to the pair's slot there as well); the host needs its sign
    // only to keep or drop the pair, which it decides at the next stream sync — the exit test of the following iteration — instead of
    // in a sync of its own.  Nothing in between depends on the decision.
    bool pending = false;
    auto resolve_pair = [&]() {
        if (!pending) return;
        pending = false;
        T dgTdx = (T)1 / (T)hscal[91];
        if (dgTdx <= (T)0 || !(dgTdx == dgTdx)) {
            if (!(dgTdx <= (T)0)) HOT_CHECK(false, HOT_ERR_NUMERIC, "NaN curvature in L-BFGS");
            pop_back();
            stats.dropped_pairs++;
        }
        push_back();
    };
    for (int it = 0; it < cfg.max_iterations; ++it) {
        stats.iterations = it;
        bool ex = should_exit(residual);
        resolve_pair();
        if (ab_flag("HOT_DEBUG")) fprintf(stderr, "[hot] lbfgs it=%d scaled_res=%.6e Ek=%.12e hist=%d\n", it, stats.final_scaled_residual, Ek, (int)order.size() - 1);
        if (ex) {
            stats.converged = 1;
            return true;
        }
        bool rebuild = cfg.useAdaptiveHessian ? ((it & 0xf) == 0) : (it == 0);
        if (rebuild) {
            build_hessian();
            build_mg();
            while (!order.empty()) pop_back();
            push_back();
        }
        int wk = order.back();
        const int m = (int)order.size() - 1; // stored curvature pairs
        const bool unfused = ab_flag("HOT_LBFGS_UNFUSED"); // A/B build only: dot / scalar / axpy as separate launches
#ifndef HOT_LB_PER_WG // elements of a DOF vector per workgroup of the two-loop's launches, and the most workgroups (tools/variant.sh experiments)
#define HOT_LB_PER_WG 1024
#define HOT_LB_MAX_WG 1024
#endif
        const int vgrid = (int)std::min<size_t>(div_up(n3, HOT_LB_PER_WG), HOT_LB_MAX_WG);
        LbVecs<T> hv{};
        if (unfused) {
            copy(n3, residual, hist_dg[wk].p);
            for (int i = m - 1; i >= 0; --i) {
                int ph = order[i];
                dot_to(n3, hist_dx[ph].p, residual, s + 70);
                HOT_LAUNCH(this, "lbfgs_scalar", k_lbfgs_scalar, 1, 1, 0, s, ph, 0);
                axpy_dev(n3, s + 50 + ph, -1.0, hist_dg[ph].p, residual);
            }
        }
        else {
            for (int i = 0; i < m; ++i) hv.dx[i] = hist_dx[order[i]].p, hv.dg[i] = hist_dg[order[i]].p, hv.ph[i] = order[i];
            hv.m = m;
            if (m > 0) {
                HOT_LAUNCH(this, "lbfgs_dots", k_lbfgs_dots<T>, vgrid, 256, 0, n3, hv, 0, residual, s + LB_B, gred_n(vgrid, 8), vmask);
                reduce_scalars(s + LB_B, m);
            }
            HOT_LAUNCH(this, "lbfgs_apply", k_lbfgs_apply<T>, vgrid, 256, 0, n3, s, hv, 0, residual, hist_dg[wk].p, vmask);
        }
        precondition_dev(residual, hist_dx[wk].p);
        project_dev(hist_dx[wk].p);
        if (unfused) {
            for (int i = 0; i < m; ++i) {
                int ph = order[i];
                dot_to(n3, hist_dg[ph].p, hist_dx[wk].p, s + 70);
                HOT_LAUNCH(this, "lbfgs_scalar", k_lbfgs_scalar, 1, 1, 0, s, ph, 1);
                axpy_dev(n3, s + 80, 1.0, hist_dx[ph].p, hist_dx[wk].p);
            }
        }
        else if (m > 0) {
            HOT_LAUNCH(this, "lbfgs_dots", k_lbfgs_dots<T>, vgrid, 256, 0, n3, hv, 1, hist_dx[wk].p, s + LB_E, gred_n(vgrid, 8), vmask);
            reduce_scalars(s + LB_E, m);
            HOT_LAUNCH(this, "lbfgs_apply", k_lbfgs_apply<T>, vgrid, 256, 0, n3, s, hv, 1, hist_dx[wk].p, (T*)nullptr, vmask);
        }
        if (cfg.linesearch) line_search(hist_dx[wk].p, residual, (T)1);
        transform_dev(hist_dx[wk].p, true); // recoverSolution
        axpy(n3, (T)1, hist_dx[wk].p, x);
        transform_dev(hist_dx[wk].p, false);
        if (!updated) {
            Ek = state_pass(x, true);
            residual_dev(residual);
        }
        if (unfused) {
            axpy(n3, (T)-1, residual, hist_dg[wk].p);
            dot_to(n3, hist_dg[wk].p, hist_dx[wk].p, s + 71, hscal + 91);
            HOT_LAUNCH(this, "lbfgs_rho", k_lbfgs_rho<T>, 1, 1, 0, s, 71, 60 + wk);
        }
        else {
            HOT_LAUNCH(this, "lbfgs_pair", k_lbfgs_pair<T>, vgrid, 256, 0, n3, hv, hist_dx[wk].p, hist_dg[wk].p, residual, s + LB_G, gred_n(vgrid, LB_MAXV), vmask);
            reduce_scalars(s + LB_G, 2 * m + 1);
            HOT_LAUNCH(this, "lbfgs_file", k_lbfgs_file<T>, 1, 1, 0, s, hv, wk, hscal + 91);
        }
        pending = true;
    }
    sync();
    resolve_pair();
    stats.iterations = cfg.max_iterations;
    return false;
}

template <class T>
__global__ void k_sub(size_t n, const T* __restrict__ a, const T* __restrict__ b, T* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}

// out_i = in_i / m_i : the lumped-mass preconditioner (MultigridSimulation.h:170-176, ImplicitSolver.h:365 with Ainv 2)
template <class T>
__global__ void k_mass_scale(const T* __restrict__ mass, const T* __restrict__ in, T* out, int nn)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * nn) return;
    out[e] = in[e] * ((T)1 / mass[e / 3]);
}
// y = (y - a x1 - b x2) / c   and   y = y / c   (MINRES three-term recurrences, Minres.h:94-95,127-138)
template <class T>
__global__ void k_minres_comb(size_t n, T* y, T a, const T* __restrict__ x1, T b, const T* __restrict__ x2, T c)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (y[i] - a * x1[i] - b * x2[i]) / c;
}
template <class T>
__global__ void k_div2(size_t n, T* y, T* z, T c)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = y[i] / c, z[i] = z[i] / c;
}

// Minres<T,TM,TV>::solve + applyAllPreviousGivensRotationsAndDetermineNewGivens (reference Lib/Ziran/Math/Linear/Minres.h:69-178)
template <class T>
int Ctx<T>::minres_dev(const std::function<void(const T*, T*)>& Amul, const std::function<void(const T*, T*)>& prec, T* x, const T* b, T relative_tolerance, T tolerance, int max_iterations)
{
    size_t n3 = 3 * (size_t)Nn;
    DBuf<T> bufs[7];
    for (auto& v : bufs) v.reserve(n3), zero(n3, v.p);
    T *mk = bufs[0].p, *mkm1 = bufs[1].p, *mkm2 = bufs[2].p, *z = bufs[3].p, *qkm1 = bufs[4].p, *qk = bufs[5].p, *qkp1 = bufs[6].p;
    T gamma = 0, delta = 0, epsilon = 0, beta_kp1 = 0, alpha_k = 0, beta_k = 0, tk = 0;
    struct G2 {
        T c = 1, s = 0;
        void compute(T a, T b)
        {
            T d = a * a + b * b;
            c = 1, s = 0;
            T sq = std::sqrt(d);
            if (sq) {
                T t = 1 / sq;
                c = a * t, s = -b * t;
            }
        }
        void rot(T& a, T& b) const
        {
            T t1 = a, t2 = b;
            a = c * t1 - s * t2, b = s * t1 + c * t2;
        }
    } Gk, Gkm1, Gkm2;
    Amul(x, qkp1);
    HOT_LAUNCH(this, "sub", k_sub<T>, div_up(n3, 256), 256, 0, n3, b, qkp1, qkp1);
    project_dev(qkp1);
    prec(qkp1, z);
    T rpn = (T)std::sqrt(dot_host(n3, z, qkp1));
    beta_kp1 = rpn;
    HOT_CHECK(beta_kp1 == beta_kp1, HOT_ERR_NUMERIC, "NaN in MINRES");
    T local_tolerance = std::min(relative_tolerance * rpn, tolerance);
    if (rpn < local_tolerance) return 0;
    if (rpn > 0) HOT_LAUNCH(this, "minres_div", k_div2<T>, div_up(n3, 256), 256, 0, n3, qkp1, z, beta_kp1);
    T rhs0 = rpn, rhs1 = 0;
    for (int k = 0; k < max_iterations; ++k) {
        if (rpn < local_tolerance) return k;
        std::swap(mkm2, mkm1);
        std::swap(mkm1, mk);
        copy(n3, z, mk);
        beta_k = beta_kp1;
        std::swap(qkm1, qkp1);
        std::swap(qkm1, qk);
        Amul(mk, qkp1);
        project_dev(qkp1);
        alpha_k = (T)dot_host(n3, mk, qkp1);
        axpy(n3, -alpha_k, qk, qkp1);
        axpy(n3, -beta_k, qkm1, qkp1);
        prec(qkp1, z);
        beta_kp1 = (T)std::sqrt(std::max(0.0, dot_host(n3, z, qkp1)));
        if (beta_kp1 > 0) HOT_LAUNCH(this, "minres_div", k_div2<T>, div_up(n3, 256), 256, 0, n3, qkp1, z, beta_kp1);
        Gkm2 = Gkm1;
        Gkm1 = Gk;
        T e0 = 0, e1 = beta_k;
        Gkm2.rot(e0, e1);
        epsilon = e0;
        T d0 = e1, d1 = alpha_k;
        Gkm1.rot(d0, d1);
        delta = d0;
        T t0 = d1, t1 = beta_kp1;
        Gk.compute(t0, t1);
        Gk.rot(t0, t1);
        gamma = t0;
        Gk.rot(rhs0, rhs1);
        tk = rhs0;
        T res = rhs1;
        rhs0 = res, rhs1 = 0;
        rpn = res < 0 ? -res : res;
        HOT_LAUNCH(this, "minres_comb", k_minres_comb<T>, div_up(n3, 256), 256, 0, n3, mk, delta, mkm1, epsilon, mkm2, gamma);
        axpy(n3, tk, mk, x);
    }
    return max_iterations;
}

// ImplicitSolverObjective::computeStep (ImplicitSolver.h:355-432): rebuild the matrix and the hierarchy (or the matrix-free
// block diagonal), then InexactConjugateGradient::solve (InexactConjugateGradient.h:49-103) or Minres::solve for the step
template <class T>
void Ctx<T>::compute_step_dev(const T* residual, T* step)
{
    size_t n3 = 3 * (size_t)Nn;
    DBuf<T>&r = nw_r, &p = nw_p, &q = nw_q, &temp = nw_t;
    r.reserve(n3, 1.25), p.reserve(n3, 1.25), q.reserve(n3, 1.25), temp.reserve(n3, 1.25);
    T cg_tolerance = cfg.useCN ? max_cn_tolerance : (T)1e-4; // MultigridSimulation.h:201-208 / MultigridInit3D.h:2500-2501
    zero(n3, step);
    const bool massPrec = !cfg.matrixFree && (cfg.lsolver == 1 || cfg.lsolver == 2) && cfg.Ainv == 2; // :365
    if (cfg.matrixFree) {
        nw_diag.reserve(9 * (size_t)Nn, 1.25);
        matfree_diagonal(nw_diag.p); // buildDiagonal :605-665
    }
    else {
        build_hessian();
        if (!massPrec) build_mg();
    }
    bool diagPrec = (cfg.levelCnt == 1 && cfg.times == 1);
    std::function<void(const T*, T*)> prec = [&](const T* in, T* out) {
        if (cfg.matrixFree)
            block_apply_dev(nw_diag.p, in, out, Nn);
        else if (massPrec)
            HOT_LAUNCH(this, "mass_scale", k_mass_scale<T>, div_up(n3, 256), 256, 0, mass.p, in, out, Nn);
        else if (diagPrec)
            scale_dev(*levels[0], in, out); // forced diagonal preconditioner (ImplicitSolver.h:376-391)
        else
            vcycle_dev(in, out);
    };
    std::function<void(const T*, T*)> Amul = [&](const T* xx, T* bb) {
        if (cfg.matrixFree)
            matfree_dev(xx, bb);
        else
            spmv_dev(*levels[0], xx, bb);
    };
    if (cfg.lsolver == 1) {
        // relative tolerance from the Newton loop (ExtendedNewtonsMethod.h:57), tolerance = maxcntol
        // (MultigridSimulation.h:204) or the scene value 1e-4 (MultigridInit3D.h:85-86)
        T residual_norm = (T)std::sqrt(dot_host(n3, residual, residual));
        T newton_tol = cfg.useCN ? max_cn_tolerance : (T)cfg.cneps;
        T rel = std::min((T)0.5, (T)std::sqrt(std::max(residual_norm, newton_tol)));
        stats.linear_iterations += minres_dev(Amul, prec, step, residual, rel, cg_tolerance, cfg.linear_iteration_cap > 0 ? cfg.linear_iteration_cap : 10000);
        return;
    }
    // b = residual (+ dRhs == 0)
    Amul(step, temp.p);
    HOT_LAUNCH(this, "sub", k_sub<T>, div_up(n3, 256), 256, 0, n3, residual, temp.p, r.p);
    project_dev(r.p);
    prec(r.p, q.p);
    copy(n3, q.p, p.p);
    double zTrk = dot_host(n3, r.p, q.p);
    T rpn = (T)std::sqrt(zTrk);
    T forcing = std::min((T)0.5, (T)std::sqrt(std::max(rpn, cg_tolerance)));
    T local_tol = forcing * rpn;
    int cnt = 0;
    for (; cnt < (cfg.linear_iteration_cap > 0 ? cfg.linear_iteration_cap : 10000); ++cnt) {
        if (rpn < local_tol) break;
        Amul(p.p, temp.p);
        project_dev(temp.p);
        T alpha = (T)(zTrk / dot_host(n3, temp.p, p.p));
        axpy(n3, alpha, p.p, step);
        axpy(n3, -alpha, temp.p, r.p);
        prec(r.p, q.p);
        double zlast = zTrk;
        zTrk = dot_host(n3, q.p, r.p);
        T beta = (T)(zTrk / zlast);
        HOT_LAUNCH(this, "scal", k_scal<T>, div_up(n3, 256), 256, 0, n3, beta, p.p);
        axpy(n3, (T)1, q.p, p.p);
        rpn = (T)std::sqrt(zTrk);
    }
    stats.linear_iterations += cnt;
}

// ExtendedNewtonsMethod::solve (Lib/Ziran/Math/Nonlinear/ExtendedNewtonsMethod.h:39-66)
template <class T>
bool Ctx<T>::newton_solve()
{
    size_t n3 = 3 * (size_t)Nn;
    T* x = dv.p;
    T* residual = work2.p;
    nw_step.reserve(n3, 1.25);
    for (int it = 0; it < cfg.max_iterations; ++it) {
        stats.iterations = it;
        if (!updated) {
            Ek = state_pass(x, true);
            residual_dev(residual);
        }
        if (should_exit(residual)) {
            stats.converged = 1;
            return true;
        }
        compute_step_dev(residual, nw_step.p);
        if (cfg.linesearch) line_search(nw_step.p, residual, (T)1);
        transform_dev(nw_step.p, true);
        axpy(n3, (T)1, nw_step.p, x);
        transform_dev(nw_step.p, false);
    }
    stats.iterations = cfg.max_iterations;
    return false;
}

// ---- the solver-facing members of the objective, one by one (C ABI: hot_line_search, hot_should_exit, hot_recover_solution,
//      hot_transform_residual, hot_compute_step), so that a host-side LBFGS / ExtendedNewtonsMethod template can drive the device
template <class T>
void Ctx<T>::line_search_api(void* ddv, void* residual, double alpha, double* alpha_out)
{
    need(Nn > 0 && dt > 0, "hot_line_search before hot_begin_step / hot_update_state");
    size_t n3 = 3 * (size_t)Nn;
    HOT_HIP(hipMemcpyAsync(work0.p, ddv, n3 * sizeof(T), hipMemcpyDefault, stream));
    T a = line_search(work0.p, work1.p, (T)alpha);
    if (halo_mode()) gather_all(*levels[0], work0.p), gather_all(*levels[0], work1.p); // the C ABI hands out complete vectors
    download(ddv, work0.p, n3), download(residual, work1.p, n3);
    sync();
    if (alpha_out) *alpha_out = (double)a;
}
template <class T>
void Ctx<T>::should_exit_api(const void* residual, int32_t* exit_now, double* scaled)
{
    need(Nn > 0 && dt > 0, "hot_should_exit before hot_begin_step");
    if (cfg.useCN) cn_tolerance_dev();
    HOT_HIP(hipMemcpyAsync(work0.p, residual, 3 * (size_t)Nn * sizeof(T), hipMemcpyDefault, stream));
    bool e = should_exit(work0.p);
    if (exit_now) *exit_now = e ? 1 : 0;
    if (scaled) *scaled = stats.final_scaled_residual;
}
template <class T>
void Ctx<T>::transform_api(void* v, bool inverse)
{
    need(Nn > 0, "no grid");
    HOT_HIP(hipMemcpyAsync(work0.p, v, 3 * (size_t)Nn * sizeof(T), hipMemcpyDefault, stream));
    transform_dev(work0.p, inverse);
    download(v, work0.p, 3 * (size_t)Nn);
    sync();
}
template <class T>
void Ctx<T>::compute_step_api(const void* residual, void* step)
{
    need(Nn > 0 && dt > 0, "hot_compute_step before hot_update_state");
    need(cfg.lsolver == 1 || cfg.lsolver == 2, "hot_compute_step is the projected-Newton step (lsolver 1 / 2)");
    size_t n3 = 3 * (size_t)Nn;
    if (cfg.useCN) cn_tolerance_dev();
    nw_step.reserve(n3, 1.25);
    HOT_HIP(hipMemcpyAsync(work2.p, residual, n3 * sizeof(T), hipMemcpyDefault, stream));
    with_gs_retry([&] {
        release_levels(halo_mode() ? 1 : 0);
        compute_step_dev(work2.p, nw_step.p);
        if (halo_mode()) gather_all(*levels[0], nw_step.p);
        sync();
    });
    download(step, nw_step.p, n3);
    sync();
}

template <class T>
void Ctx<T>::solve(hot_stats* st)
{
    need(Nn > 0 && dt > 0, "hot_solve before hot_begin_step");
    need(cfg.lsolver == 1 || cfg.lsolver == 2 || cfg.lsolver == 3, "lsolver must be 1 (PN + MINRES), 2 (PN + PCG) or 3 (L-BFGS); 0 performs no linear solve in the reference and 4 is the SimplicialLLT direct solver (out of scope)");
    need(!(cfg.matrixFree && cfg.lsolver == 3), "matrixFree applies to the projected-Newton solvers (lsolver 1 / 2)");
    need(!(cfg.Ainv == 2 && cfg.lsolver == 3), "Ainv 2 (lumped-mass preconditioner) applies to lsolver 1 / 2 (Configurations.h:30)");
    double keep_sort = stats.ms_sort, keep_p2g = stats.ms_p2g, keep_begin = stats.ms_begin;
    std::memset(&stats, 0, sizeof(stats));
    stats.ms_sort = keep_sort, stats.ms_p2g = keep_p2g, stats.ms_begin = keep_begin;
    double t0 = wall_ms();
    if (cfg.useCN) cn_tolerance_dev();
    // what the solve overwrites of its own inputs, kept for the (never yet observed) redo after a timed-out chained sweep
    const size_t n3 = 3 * (size_t)Nn;
    solve_keep.reserve(2 * n3, 1.25);
    copy(n3, dv.p, solve_keep.p), copy(n3, dv0.p, solve_keep.p + n3);
    const double Ek_in = Ek;
    const hot_stats stats_in = stats;
    bool first_try = true;
    with_gs_retry([&] {
        if (!first_try) {
            copy(n3, solve_keep.p, dv.p), copy(n3, solve_keep.p + n3, dv0.p);
            updated = false, Ek = Ek_in, stats = stats_in; // the state pass is redone from dv (the failed attempt overwrote the force tiles)
        }
        first_try = false;
        release_levels(halo_mode() ? 1 : 0); // halo mode: level 0 (coordinates, row ownership, exchange lists) exists since hot_p2g
        if (cfg.lsolver == 3)
            lbfgs_solve();
        else
            newton_solve();
        sync();
    });
    prof.collect();
    stats.num_nodes = Nn;
    stats.num_levels = (halo_mode() && cfg.matrixFree) ? 0 : (int)levels.size(); // halo mode keeps a matrix-less level 0 (row ownership, exchange lists) also for the matrix-free solvers
    stats.energy = cfg.linesearch ? Ek : 0.0; // the incremental potential at the last accepted line-search point; without a line search nobody evaluates it there (ImplicitSolver.h:237-252)
    stats.ms_solve = wall_ms() - t0;
    export_comm_stats();
    if (st) *st = stats;
}

template <class T>
void Ctx<T>::advance(double dt_, hot_stats* st)
{
    double t0 = wall_ms();
    sort();
    p2g();
    begin_step(dt_);
    solve(nullptr);
    int32_t f = 0;
    g2p(dt_, &f);
    stats.ms_total = wall_ms() - t0;
    export_comm_stats();
    if (st) *st = stats;
}

// evalMaxParticleSpeed (MpmSimulationBase.cpp:1186-1218): max |v_p| and the particle bounding box.  Block maxima are
// written to a small array and folded by one more workgroup (max is exact, so the order does not matter).
template <class T>
__global__ __launch_bounds__(256) void k_max_speed(const T* __restrict__ X, const T* __restrict__ V, int64_t Np, T* __restrict__ part /*[gridDim][8]*/)
{
    __shared__ T red[8][4];
    T r[7];
    r[0] = (T)0;
#pragma unroll
    for (int d = 1; d < 7; ++d) r[d] = -(T)3.4e38;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < Np; p += (int64_t)gridDim.x * 256) {
        const T vx = V[p], vy = V[Np + p], vz = V[2 * Np + p];
        r[0] = fmax(r[0], hsqrt(vx * vx + vy * vy + vz * vz));
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const T x = X[(int64_t)d * Np + p];
            r[1 + d] = fmax(r[1 + d], x), r[4 + d] = fmax(r[4 + d], -x);
        }
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        T v = r[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 7) part[blockIdx.x * 8 + threadIdx.x] = fmax(fmax(red[threadIdx.x][0], red[threadIdx.x][1]), fmax(red[threadIdx.x][2], red[threadIdx.x][3]));
}
template <class T>
__global__ void k_max_fold(const T* __restrict__ part, int nb, double* out)
{
    const int q = threadIdx.x;
    if (q >= 7) return;
    T v = part[q];
    for (int b = 1; b < nb; ++b) v = fmax(v, part[b * 8 + q]);
    out[q] = (double)v;
}

template <class T>
void Ctx<T>::calculate_dt(double max_dt, double* dt_out, double* max_speed, double* min_corner, double* max_corner)
{
    need(Np > 0, "hot_calculate_dt before hot_set_particles");
    const int nb = (int)std::min<int64_t>(div_up(Np, 1024), 256);
    speed_part.reserve(8 * 256); // kept across calls: once per substep in advance_frame
    HOT_LAUNCH(this, "max_speed", k_max_speed<T>, nb, 256, 0, pX.p, pV.p, Np, speed_part.p);
    HOT_LAUNCH(this, "max_speed_fold", k_max_fold<T>, 1, 64, 0, speed_part.p, nb, dscal.p + 200);
    HOT_HIP(hipMemcpyAsync(hscal + 200, dscal.p + 200, 7 * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    if (sharded()) c_allreduce(hscal + 200, 7, HOT_COMM_F64, HOT_COMM_MAX, false); // max speed, max corner, -min corner over the shards
    T ms = (T)hscal[200];
    if (!cobjs.empty()) { // :802-806 collision objects inside the particle box expanded by (degree + 2) dx
        double lo[3], hi[3];
        for (int d = 0; d < 3; ++d) lo[d] = (double)((T)-hscal[204 + d] - (T)4 * dx), hi[d] = (double)((T)hscal[201 + d] + (T)4 * dx);
        for (size_t k = 0; k < cobjs.size(); k += 1 + ((cobjs[k].shape == HOT_SHAPE_UNION || cobjs[k].shape == HOT_SHAPE_DIFFERENCE) ? (size_t)cobjs[k].p1[0] : 0))
            ms = std::max(ms, (T)co_max_speed(&cobjs[k], lo, hi)); // members of a composite ride along behind it
    }
    T dtc = (T)max_dt;
    if (ms) dtc = (T)cfg.cfl * dx / ms; // :807-809, in the scalar type of the simulation
    if (dt_out) *dt_out = (double)dtc;
    if (max_speed) *max_speed = (double)ms; // max(particles, collision objects), what the step is computed from
    for (int d = 0; d < 3; ++d) {
        if (max_corner) max_corner[d] = hscal[201 + d];
        if (min_corner) min_corner[d] = -hscal[204 + d];
    }
}

template <class T>
void Ctx<T>::advance_frame(double frame_dt, double min_dt, double max_dt, int32_t* substeps, int32_t* iterations_total, hot_stats* st)
{
    need(frame_dt > 0 && min_dt > 0 && max_dt > 0, "hot_advance_frame: frame_dt, min_dt and max_dt must be positive");
    double since = 0;
    int n = 0, its = 0;
    hot_stats last;
    std::memset(&last, 0, sizeof(last));
    for (;;) {
        double dtc = 0;
        calculate_dt(max_dt, &dtc, nullptr, nullptr, nullptr);
        // TimeStepping::nextDt (TimeStepping.h:45-59)
        HOT_CHECK(dtc > 0, HOT_ERR_NUMERIC, "calculateDt returned a non-positive step");
        double d = (dtc < min_dt) ? min_dt : (dtc > max_dt) ? max_dt : dtc;
        if (since + d >= frame_dt)
            d = frame_dt - since;
        else if (since + 2 * d > frame_dt)
            d = (frame_dt - since) / 2;
        advance(d, &last);
        its += last.iterations;
        ++n;
        since += d; // TimeStepping::advance (:67-76)
        if (since >= frame_dt) break;
    }
    if (substeps) *substeps = n;
    if (iterations_total) *iterations_total = its;
    if (st) *st = last;
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
