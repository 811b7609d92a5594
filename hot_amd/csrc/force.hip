// libhotmi355x — backward-Euler objective pieces on the device.
//
//   begin_step      MultigridSimulation::startBackwardEuler (reference Projects/multigrid/MultigridSimulation.h:167-186),
//                   MpmSimulationBase::buildInitialDvAndVnForNewton (Lib/MPM/MpmSimulationBase.cpp:1139-1184, collision
//                   query result supplied through hot_set_bc or evaluated here for sticky half spaces),
//                   FBasedMpmForceHelper::backupStrain (Lib/MPM/Force/FBasedMpmForceHelper.cpp:24-33), resetLSFlag.
//   k_state         one particle pass for objective.updateState + totalEnergy:  evalInterpolantAndGradient of vn+dv
//                   (Lib/MPM/Force/MpmForceBase.cpp:213-248), restoreStrain/evolveStrain (FBasedMpmForceHelper.cpp:35-43,
//                   99-114), updateImplicitState (:70-97), FBased totalEnergy (:116-135).  The reference runs these as
//                   separate particle sweeps with two SVDs per particle; here: one launch, one SVD, LDS-staged node tile.
//   k_force_cells2  rasterizeForceToTVStack<false> (MpmForceBase.cpp:100-153): (cell half, node row) items summed in
//                   registers, one partial tile per particle group, ordered reduce (no colour passes, no global
//                   atomics).  k_force_scatter (one LDS atomic per particle and node) and k_force_cells (3-node items,
//                   27 staged scalars) are the earlier versions, kept in the A/B build.
//   k_residual      computeResidual (Projects/multigrid/ImplicitSolver.h:128-155) incl. MassLumpedInertia::addScaledForces
//                   (Lib/Ziran/Physics/LagrangianForce/Inertia.cpp:33-41), transformResidual (:117-125), project.
//   k_matfree       matrix-free Hessian product (ImplicitSolver.h:741-758, MpmForceBase.cpp:262-306,
//                   FBasedMpmForceHelper.cpp:137-160).
#include "hot_impl.h"
#include "hot_constitutive.h"
#include "hot_collision.h"
#include <cmath>

namespace hot {

template <class T>
__device__ __forceinline__ int tile_slot2(int t, const int32_t* __restrict__ nb8)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2;
    int tz = t % TZ, ty = (t / TZ) % TY, tx = t / (TZ * TY);
    int ox = tx >> G::xb, oy = ty >> G::yb, oz = tz >> G::zb;
    int elem = ((tx & (G::BX - 1)) << (G::yb + G::zb)) | ((ty & (G::BY - 1)) << G::zb) | (tz & (G::BZ - 1));
    return nb8[ox * 4 + oy * 2 + oz] * G::EPB + elem;
}

// ------------------------------------------------------------------------------------------------ BCs
template <class T>
__global__ void k_hs_flag(const int32_t* __restrict__ id2coord, const double* __restrict__ hs, int nhs, int32_t* flags, int nn, T dx)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    bool inside = false;
    for (int h = 0; h < nhs; ++h) {
        double s = 0;
        for (int d = 0; d < 3; ++d) s += ((double)((T)id2coord[3 * n + d] * dx) - hs[6 * h + d]) * hs[6 * h + 3 + d];
        if (s <= 0) inside = true;
    }
    flags[n] = inside ? 1 : 0;
}
template <class T>
__global__ void k_hs_fill(const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int32_t* bcNode, T* P, T* R, T* Rinv, uint8_t* slip, uint8_t* hasdv, int nn)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn || !flags[n]) return;
    int c = scan[n];
    bcNode[c] = n;
    for (int k = 0; k < 9; ++k) {
        P[9 * c + k] = (T)0;
        R[9 * c + k] = Rinv[9 * c + k] = (k % 4 == 0) ? (T)1 : (T)0;
    }
    slip[c] = 0;
    hasdv[c] = 0;
}
// analytic collision objects (hot_collision.h): pass 1 flags the colliding nodes, pass 2 (after the scan) writes the
// compacted CollisionNode records {node, P = I - nb nb^T, R, R^-1, isSlip} and the Newton initial guess dv = vi - v
template <class T>
__global__ void k_co_flag(const int32_t* __restrict__ id2coord, const T* __restrict__ nodeV, const CollObj<T>* __restrict__ objs, int nobj, int32_t* flags, int nn, T dx)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    const T x[3] = { (T)id2coord[3 * n] * dx, (T)id2coord[3 * n + 1] * dx, (T)id2coord[3 * n + 2] * dx };
    T v[3] = { nodeV[3 * n], nodeV[3 * n + 1], nodeV[3 * n + 2] }, nb[9], wn[3];
    flags[n] = co_multi(objs, nobj, x, v, nb, wn) ? 1 : 0;
}
template <class T>
__global__ void k_co_fill(const int32_t* __restrict__ id2coord, const T* __restrict__ nodeV, const CollObj<T>* __restrict__ objs, int nobj, const int32_t* __restrict__ flags,
    const int32_t* __restrict__ scan, int32_t* bcNode, T* P, T* R, T* Rinv, T* bcDv, uint8_t* slip, uint8_t* hasdv, int nn, T dx)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn || !flags[n]) return;
    const T x[3] = { (T)id2coord[3 * n] * dx, (T)id2coord[3 * n + 1] * dx, (T)id2coord[3 * n + 2] * dx };
    const T v0[3] = { nodeV[3 * n], nodeV[3 * n + 1], nodeV[3 * n + 2] };
    T v[3] = { v0[0], v0[1], v0[2] }, nb[9], wn[3];
    co_multi(objs, nobj, x, v, nb, wn);
    const int c = scan[n];
    bcNode[c] = n;
    // P = I - nb nb^T (column-major)
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) {
            T acc = (T)0;
            for (int k = 0; k < 3; ++k) acc += nb[row + 3 * k] * nb[col + 3 * k];
            P[9 * c + row + 3 * col] = (row == col ? (T)1 : (T)0) - acc;
        }
    const bool isSlip = wn[0] != (T)0 || wn[1] != (T)0 || wn[2] != (T)0;
    Mat3<T> Rm;
    if (isSlip)
        co_rotate_to_x(wn, Rm.a);
    else
        for (int k = 0; k < 9; ++k) Rm.a[k] = (k % 4 == 0) ? (T)1 : (T)0;
    Mat3<T> Ri = m3_inverse(Rm);
    for (int k = 0; k < 9; ++k) R[9 * c + k] = Rm.a[k], Rinv[9 * c + k] = Ri.a[k];
    slip[c] = isSlip ? 1 : 0;
    hasdv[c] = 1;
    for (int k = 0; k < 3; ++k) bcDv[3 * c + k] = v[k] - v0[k];
}
__global__ void k_bc_index(const int32_t* __restrict__ bcNode, int32_t* bcIdx, int nc)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nc) bcIdx[bcNode[c]] = c;
}
template <class T>
__global__ void k_begin(const T* __restrict__ nodeV, const int32_t* __restrict__ bcIdx, const T* __restrict__ bcDv, const uint8_t* __restrict__ hasdv, T* dv, T* vn, T* dv0,
    int nn, T g0, T g1, T g2, T dt)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    T v[3] = { nodeV[3 * n], nodeV[3 * n + 1], nodeV[3 * n + 2] };
    int c = bcIdx[n];
    T d[3];
    if (c >= 0) {
        if (hasdv[c])
            d[0] = bcDv[3 * c], d[1] = bcDv[3 * c + 1], d[2] = bcDv[3 * c + 2];
        else
            d[0] = -v[0], d[1] = -v[1], d[2] = -v[2];
    }
    else
        d[0] = g0 * dt, d[1] = g1 * dt, d[2] = g2 * dt;
    for (int k = 0; k < 3; ++k) dv[3 * n + k] = d[k], dv0[3 * n + k] = d[k], vn[3 * n + k] = v[k];
}

template <class T>
void Ctx<T>::set_bc(int32_t nc, const int32_t* node_id, const void* P, const void* R, const void* Rinv, const uint8_t* slip, const void* dvc)
{
    need(Nn > 0, "hot_set_bc before hot_p2g");
    need(nc >= 0 && (nc == 0 || (node_id && P)), "hot_set_bc: node_id and P are required");
    hs_origin.clear(), hs_normal.clear();
    Nc = nc;
    size_t n = std::max(nc, 1);
    bcNode.reserve(n), bcP.reserve(9 * n), bcR.reserve(9 * n), bcRinv.reserve(9 * n), bcDv.reserve(3 * n), bcSlip.reserve(n), bcHasDv.reserve(n);
    if (nc == 0) return;
    std::vector<int32_t> ids(nc);
    HOT_HIP(hipMemcpy(ids.data(), node_id, nc * sizeof(int32_t), hipMemcpyDefault));
    for (int c = 0; c < nc; ++c) HOT_CHECK(ids[c] >= 0 && ids[c] < Nn, HOT_ERR_INVALID, "collision node id out of range");
    upload(bcNode, ids.data(), nc);
    sync();
    upload(bcP, P, 9 * (size_t)nc);
    std::vector<T> eye(9 * (size_t)nc, (T)0);
    for (int c = 0; c < nc; ++c) eye[9 * c] = eye[9 * c + 4] = eye[9 * c + 8] = (T)1;
    upload(bcR, R ? R : (const void*)eye.data(), 9 * (size_t)nc);
    upload(bcRinv, Rinv ? Rinv : (const void*)eye.data(), 9 * (size_t)nc);
    if (slip)
        upload(bcSlip, slip, nc);
    else
        HOT_HIP(hipMemsetAsync(bcSlip.p, 0, nc, stream));
    HOT_HIP(hipMemsetAsync(bcHasDv.p, dvc ? 1 : 0, nc, stream));
    if (dvc) upload(bcDv, dvc, 3 * (size_t)nc);
    sync();
}

template <class T>
void Ctx<T>::set_halfspaces(int32_t n, const double* origin, const double* normal)
{
    cobjs.clear();
    hs_origin.clear(), hs_normal.clear();
    if (n > 0) hs_origin.assign(origin, origin + 3 * n), hs_normal.assign(normal, normal + 3 * n);
    if (n == 0) Nc = 0;
}

template <class T>
void Ctx<T>::set_collision_objects(int32_t n, const hot_collision_object* objs)
{
    need(n >= 0 && n <= 64, "hot_set_collision_objects: 0 <= n <= 64");
    for (int i = 0, members_left = 0; i < n; ++i) {
        need(objs[i].shape >= HOT_SHAPE_HALFSPACE && objs[i].shape <= HOT_SHAPE_DIFFERENCE, "unknown collision shape");
        const bool composite = objs[i].shape == HOT_SHAPE_UNION || objs[i].shape == HOT_SHAPE_DIFFERENCE;
        if (members_left > 0) { // a member of the composite before it: a primitive, only shape / p0 / p1 / lsq are read
            --members_left;
            need(!composite, "a composite level set cannot be a member of a composite");
            if (objs[i].shape == HOT_SHAPE_CAPPED_CYLINDER || objs[i].shape == HOT_SHAPE_TORUS || objs[i].shape == HOT_SHAPE_ROTATED_BOX)
                need(objs[i].lsq[0] != 0 || objs[i].lsq[1] != 0 || objs[i].lsq[2] != 0 || objs[i].lsq[3] != 0, "lsq must be a rotation quaternion ((1,0,0,0) = none)");
            need(!(objs[i].shape == HOT_SHAPE_HALFSPACE && objs[i].p1[0] == 0 && objs[i].p1[1] == 0 && objs[i].p1[2] == 0), "half space: the outward normal p1 must be non-zero");
            continue;
        }
        if (composite) {
            const int nm = (int)objs[i].p1[0];
            need(nm >= 1 && (double)nm == objs[i].p1[0] && i + nm < n, "composite level set: p1[0] = number of member records that follow it");
            need(objs[i].shape != HOT_SHAPE_DIFFERENCE || nm == 2, "DIFFERENCE takes exactly two members (A, then B)");
            for (int m = 1; m <= nm; ++m) {
                const int sh = objs[i + m].shape;
                need(!((sh == HOT_SHAPE_BOX || sh == HOT_SHAPE_CAPPED_CYLINDER || sh == HOT_SHAPE_ROTATED_BOX) && objs[i].type != HOT_COLLISION_STICKY),
                    "a composite with a box / capped-cylinder member must be STICKY (their automatic-differentiation normal is not restated)");
                need(!(sh == HOT_SHAPE_HALFSPACE && (objs[i].dsdt != 0 || objs[i].omega[0] != 0 || objs[i].omega[1] != 0 || objs[i].omega[2] != 0)),
                    "a composite with a half-space member cannot turn or scale (no bounds available for its speed)");
            }
            members_left = nm;
        }
        need(objs[i].type >= HOT_COLLISION_STICKY && objs[i].type <= HOT_COLLISION_SEPARATE, "collision type must be STICKY (1), SLIP (2) or SEPARATE (3)");
        need(!((objs[i].shape == HOT_SHAPE_BOX || objs[i].shape == HOT_SHAPE_CAPPED_CYLINDER || objs[i].shape == HOT_SHAPE_ROTATED_BOX) && objs[i].type != HOT_COLLISION_STICKY),
            "boxes and capped cylinders must be STICKY (the reference's automatic-differentiation normal is undefined inside them)");
        if (objs[i].shape == HOT_SHAPE_CAPPED_CYLINDER || objs[i].shape == HOT_SHAPE_TORUS || objs[i].shape == HOT_SHAPE_ROTATED_BOX)
            need(objs[i].lsq[0] != 0 || objs[i].lsq[1] != 0 || objs[i].lsq[2] != 0 || objs[i].lsq[3] != 0, "lsq must be a rotation quaternion ((1,0,0,0) = none)");
        need(objs[i].s > 0, "collision object scaling s must be > 0 (1 = none)");
        need(!(objs[i].shape == HOT_SHAPE_HALFSPACE && objs[i].p1[0] == 0 && objs[i].p1[1] == 0 && objs[i].p1[2] == 0), "half space: the outward normal p1 must be non-zero (it is normalised here, as HalfSpace's constructor does)");
        need(!(objs[i].shape == HOT_SHAPE_HALFSPACE && (objs[i].dsdt != 0 || objs[i].omega[0] != 0 || objs[i].omega[1] != 0 || objs[i].omega[2] != 0)),
            "a half space cannot turn or scale (no bounds available for its speed: AnalyticLevelSet.cpp:122-125)");
        double dev = 0; // R^T R = I
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double g = 0;
                for (int k = 0; k < 3; ++k) g += objs[i].R[3 * a + k] * objs[i].R[3 * b + k];
                dev = std::max(dev, std::abs(g - (a == b ? 1.0 : 0.0)));
            }
        need(dev < 1e-6, "collision object R must be a rotation matrix (identity when the object does not turn)");
    }
    cobjs.assign(objs, objs + n);
    hs_origin.clear(), hs_normal.clear();
    if (n == 0) Nc = 0;
}

template <class T>
void Ctx<T>::eval_collision_objects()
{
    if (cobjs.empty()) return;
    const int nobj = (int)cobjs.size();
    std::vector<CollObj<T>> h(nobj);
    for (int i = 0; i < nobj; ++i) {
        h[i].shape = cobjs[i].shape, h[i].type = cobjs[i].type, h[i].friction = (T)cobjs[i].friction;
        for (int d = 0; d < 3; ++d) h[i].p0[d] = (T)cobjs[i].p0[d], h[i].p1[d] = (T)cobjs[i].p1[d], h[i].b[d] = (T)cobjs[i].b[d], h[i].dbdt[d] = (T)cobjs[i].dbdt[d], h[i].omega[d] = (T)cobjs[i].omega[d];
        for (int d = 0; d < 9; ++d) h[i].R[d] = (T)cobjs[i].R[d];
        if (cobjs[i].shape == HOT_SHAPE_HALFSPACE) { // HalfSpace stores outward_normal.normalized() (AnalyticLevelSet.cpp:111-115)
            const T nn = std::sqrt(h[i].p1[0] * h[i].p1[0] + h[i].p1[1] * h[i].p1[1] + h[i].p1[2] * h[i].p1[2]);
            for (int d = 0; d < 3; ++d) h[i].p1[d] = h[i].p1[d] / nn;
        }
        h[i].inv_s = (T)1 / (T)cobjs[i].s, h[i].dsdt = (T)cobjs[i].dsdt;
        h[i].nmember = (cobjs[i].shape == HOT_SHAPE_UNION || cobjs[i].shape == HOT_SHAPE_DIFFERENCE) ? (int32_t)cobjs[i].p1[0] : 0;
        double Rls[9];
        co_quat_to_matrix(cobjs[i].lsq, Rls);
        for (int d = 0; d < 9; ++d) h[i].Rls[d] = (T)Rls[d];
    }
    d_cobjs.reserve(nobj * sizeof(CollObj<T>));
    HOT_HIP(hipMemcpyAsync(d_cobjs.p, h.data(), nobj * sizeof(CollObj<T>), hipMemcpyHostToDevice, stream));
    sync(); // h goes out of scope
    const CollObj<T>* objs = (const CollObj<T>*)d_cobjs.p;
    flags.reserve(Nn), scan.reserve(Nn);
    HOT_LAUNCH(this, "bc_objects_flag", k_co_flag<T>, div_up(Nn, 256), 256, 0, id2coord.p, nodeV.p, objs, nobj, flags.p, Nn, dx);
    Nc = exclusive_scan_i32(flags.p, scan.p, Nn);
    size_t n = std::max(Nc, 1);
    bcNode.reserve(n), bcP.reserve(9 * n), bcR.reserve(9 * n), bcRinv.reserve(9 * n), bcDv.reserve(3 * n), bcSlip.reserve(n), bcHasDv.reserve(n);
    HOT_LAUNCH(this, "bc_objects_fill", k_co_fill<T>, div_up(Nn, 256), 256, 0, id2coord.p, nodeV.p, objs, nobj, flags.p, scan.p, bcNode.p, bcP.p, bcR.p, bcRinv.p, bcDv.p, bcSlip.p, bcHasDv.p, Nn, dx);
}

template <class T>
void Ctx<T>::eval_halfspaces()
{
    if (hs_origin.empty()) return;
    int nhs = (int)hs_origin.size() / 3;
    std::vector<double> h(6 * nhs);
    for (int i = 0; i < nhs; ++i)
        for (int d = 0; d < 3; ++d) h[6 * i + d] = hs_origin[3 * i + d], h[6 * i + 3 + d] = hs_normal[3 * i + d];
    upload(d_hs, h.data(), h.size());
    flags.reserve(Nn), scan.reserve(Nn);
    HOT_LAUNCH(this, "bc_halfspace_flag", k_hs_flag<T>, div_up(Nn, 256), 256, 0, id2coord.p, d_hs.p, nhs, flags.p, Nn, dx);
    Nc = exclusive_scan_i32(flags.p, scan.p, Nn);
    size_t n = std::max(Nc, 1);
    bcNode.reserve(n), bcP.reserve(9 * n), bcR.reserve(9 * n), bcRinv.reserve(9 * n), bcDv.reserve(3 * n), bcSlip.reserve(n), bcHasDv.reserve(n);
    HOT_LAUNCH(this, "bc_halfspace_fill", k_hs_fill<T>, div_up(Nn, 256), 256, 0, flags.p, scan.p, bcNode.p, bcP.p, bcR.p, bcRinv.p, bcSlip.p, bcHasDv.p, Nn);
}

template <class T>
void Ctx<T>::begin_step(double dt_)
{
    need(Nn > 0, "hot_begin_step before hot_p2g");
    double t0 = wall_ms();
    rearm_chain();
    dt = (T)dt_;
    eval_halfspaces();
    eval_collision_objects();
    HOT_HIP(hipMemsetAsync(bcIdx.p, 0xff, (size_t)Nn * sizeof(int32_t), stream));
    if (Nc > 0) HOT_LAUNCH(this, "bc_index", k_bc_index, div_up(Nc, 256), 256, 0, bcNode.p, bcIdx.p, Nc);
    HOT_LAUNCH(this, "begin_step", k_begin<T>, div_up(Nn, 256), 256, 0, nodeV.p, bcIdx.p, bcDv.p, bcHasDv.p, dv.p, vn.p, dv0.p, Nn, (T)cfg.gravity[0], (T)cfg.gravity[1], (T)cfg.gravity[2], dt);
    HOT_HIP(hipMemcpyAsync(pFn.p, pF.p, 9 * (size_t)Np * sizeof(T), hipMemcpyDeviceToDevice, stream));
    updated = false, ls_prev_trials = 1;
    release_levels(halo_mode() ? 1 : 0); // halo mode: level 0 (coordinates, row ownership, exchange lists) was set up by hot_p2g and lasts for the step
    stats.ms_begin = wall_ms() - t0;
}

template <class T>
void Ctx<T>::get_dv(void* out)
{
    need(Nn > 0, "no grid");
    if (halo_mode()) gather_all(*levels[0], dv.p); // the C ABI hands out complete vectors
    download(out, dv.p, 3 * (size_t)Nn);
    sync();
}
template <class T>
void Ctx<T>::set_dv(const void* in)
{
    need(Nn > 0, "no grid");
    HOT_HIP(hipMemcpyAsync(dv.p, in, 3 * (size_t)Nn * sizeof(T), hipMemcpyDefault, stream));
    sync();
}

// ------------------------------------------------------------------------------------------------ state pass
// rotate the (j,k) visiting order per lane with compile-time loop structure: the particles of one cell sit in
// adjacent lanes and would otherwise hit the same LDS address in the same ds_add instruction
template <class T>
__device__ __forceinline__ void rot3(const T (&in)[3], int r, T (&out)[3])
{
    out[0] = r == 0 ? in[0] : (r == 1 ? in[1] : in[2]);
    out[1] = r == 0 ? in[1] : (r == 1 ? in[2] : in[0]);
    out[2] = r == 0 ? in[2] : (r == 1 ? in[0] : in[1]);
}

// grad(v) at a particle from the LDS node tile (evalInterpolantAndGradient's sum over the 27 nodes): one body for the state pass and
// the batched line-search trials, so that both round alike.  fp64: sum-factorised like k_g2p's (round 5): the tensor-product weights are
// contracted one axis at a time — along z per (i, j): value and derivative sums (18 multiply-adds), along y per i (27), along x (27) —
// 280 operations instead of the 378 of the node-by-node sum (135 weight products + 243 multiply-adds); the state passes are bound by FP64 issue.
template <class T>
__device__ __forceinline__ void gather_grad(const T* __restrict__ n0, const T* __restrict__ n1, const T* __restrict__ n2, int t0, int sx, int sy, const T (&w)[3][3], const T (&dw)[3][3],
    T one_over_dx, T (&gv)[9])
{
#ifdef HOT_GATHER_NODEWISE // (experiment builds, tools/variant.sh: the node-by-node sum in fp64 too)
    constexpr bool nodewise = true;
#else
    constexpr bool nodewise = sizeof(T) == 4;
#endif
    if constexpr (nodewise) { // fp32: node by node, the reference's summation order in float (the fp32 whole-step parity test compares energies to 1e-5); not bound by issue there
#pragma unroll
        for (int c = 0; c < 9; ++c) gv[c] = (T)0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T wi = w[0][i], dwi = one_over_dx * dw[0][i];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                T wij = wi * w[1][j];
                T dwij_i = dwi * w[1][j], dwij_j = wi * one_over_dx * dw[1][j];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    T g0 = dwij_i * w[2][k], g1 = dwij_j * w[2][k], g2 = wij * one_over_dx * dw[2][k];
                    int t = t0 + i * sx + j * sy + k;
                    T v0 = n0[t], v1 = n1[t], v2 = n2[t];
                    gv[0] += v0 * g0, gv[1] += v1 * g0, gv[2] += v2 * g0;
                    gv[3] += v0 * g1, gv[4] += v1 * g1, gv[5] += v2 * g1;
                    gv[6] += v0 * g2, gv[7] += v1 * g2, gv[8] += v2 * g2;
                }
            }
        }
        return;
    }
    T gx[3] = { 0, 0, 0 }, gy[3] = { 0, 0, 0 }, gz[3] = { 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T a[3] = { 0, 0, 0 }, b[3] = { 0, 0, 0 }, c[3] = { 0, 0, 0 }; // sums over (j, k) with weights w1 w2, dw1 w2, w1 dw2
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int t = t0 + i * sx + j * sy;
            const T v0[3] = { n0[t], n0[t + 1], n0[t + 2] }, v1[3] = { n1[t], n1[t + 1], n1[t + 2] }, v2[3] = { n2[t], n2[t + 1], n2[t + 2] };
            const T s0 = w[2][0] * v0[0] + w[2][1] * v0[1] + w[2][2] * v0[2], d0 = dw[2][0] * v0[0] + dw[2][1] * v0[1] + dw[2][2] * v0[2];
            const T s1 = w[2][0] * v1[0] + w[2][1] * v1[1] + w[2][2] * v1[2], d1 = dw[2][0] * v1[0] + dw[2][1] * v1[1] + dw[2][2] * v1[2];
            const T s2 = w[2][0] * v2[0] + w[2][1] * v2[1] + w[2][2] * v2[2], d2 = dw[2][0] * v2[0] + dw[2][1] * v2[1] + dw[2][2] * v2[2];
            a[0] += w[1][j] * s0, a[1] += w[1][j] * s1, a[2] += w[1][j] * s2;
            b[0] += dw[1][j] * s0, b[1] += dw[1][j] * s1, b[2] += dw[1][j] * s2;
            c[0] += w[1][j] * d0, c[1] += w[1][j] * d1, c[2] += w[1][j] * d2;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) gx[q] += dw[0][i] * a[q], gy[q] += w[0][i] * b[q], gz[q] += w[0][i] * c[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) gv[q] = one_over_dx * gx[q], gv[3 + q] = one_over_dx * gy[q], gv[6 + q] = one_over_dx * gz[q];
}

// pass A: gather grad(vn+dv) from the LDS node tile -> trial F -> one SVD -> psi, P -> stress = V_p P Fn^T, energy
// ENERGY_ONLY (a line-search trial that may be rejected, ImplicitSolver.h:312-333 reads nothing but the energy): no trial-F / stress
// stores, and no SVD: psi from the invariants of F^T F (corotated_psi_invariants, hot_constitutive.h; where those decline — det F <= 0.1 — from the
// singular values alone: the rotations that would accumulate U and V are dead code).  mu |F - R|_F^2 up to round-off (CorotatedIsotropic.h:151-155);
// the full pass emits the same evaluation of its F beside its own energy (es), which is what a trial is compared with.  The accepted
// point is always re-evaluated by the full pass, whose energy is the one the solver keeps.
// Occupancy is stated (waves per SIMD: 4 / 3 in fp64, 6 / 4 in fp32 — 128 / 168 registers without spills): a workgroup's life is a chain of
// dependent round trips and the trial pass is bound by how many of them a compute unit holds, not by arithmetic or bytes.
template <class T, bool ENERGY_ONLY = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ENERGY_ONLY ? (sizeof(T) == 8 ? 4 : 6) : (sizeof(T) == 8 ? 3 : 4)))) void k_state(const T* __restrict__ X, const T* __restrict__ Fn, const T* __restrict__ Vol, const T* __restrict__ Mu, const T* __restrict__ Lam,
    T* __restrict__ Ft, T* __restrict__ stress_out, T* __restrict__ gradV_out, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin,
    const int32_t* __restrict__ group_nb, const int32_t* __restrict__ gIdx, const T* __restrict__ vn, const T* __restrict__ dv, T dx, T one_over_dx, T dt, double* energy, GridRed gr)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    __shared__ T nv[3][TILE];
    __shared__ double red[4];
    const int g = blockIdx.x;
    const int first = group_first[g], last = group_first[g + 1];
    // position and Fn of this thread's first particle are requested before the tile gather: two of the workgroup's dependent
    // round trips (tile indices -> nodal values, particle data) then run side by side
    const int p0 = first + threadIdx.x;
    T xpre[3] = { 0, 0, 0 }, fpre[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) fpre[c] = (T)0;
    if (p0 < last) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xpre[d] = X[(int64_t)d * Np + p0];
#pragma unroll
        for (int c = 0; c < 9; ++c) fpre[c] = Fn[(int64_t)c * Np + p0];
    }
    for (int t = threadIdx.x; t < TILE; t += 256) {
        int idx = gIdx[(int64_t)g * TILE + t]; // gIdx here = tileDof: the tile's DOF ids, one load ahead of the values
        T a = 0, b = 0, c = 0;
        if (idx >= 0) a = vn[3 * idx] + dv[3 * idx], b = vn[3 * idx + 1] + dv[3 * idx + 1], c = vn[3 * idx + 2] + dv[3 * idx + 2];
        nv[0][t] = a, nv[1][t] = b, nv[2][t] = c;
    }
    __syncthreads();
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    double e = 0, es = 0;
    for (int p = first + threadIdx.x; p < last; p += 256) {
        Mat3<T> Fnew;
        {
            T xp[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) xp[d] = p == p0 ? xpre[d] : X[(int64_t)d * Np + p];
            int base[3];
            T w[3][3], dw[3][3];
#pragma unroll
            for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xp[d], base[d], w[d], dw[d]);
            const int cx = base[0] - ox, cy = base[1] - oy, cz = base[2] - oz;
            T gv[9];
            gather_grad(nv[0], nv[1], nv[2], ((cx * TY) + cy) * TZ + cz, TY * TZ, TZ, w, dw, one_over_dx, gv);
            if (gradV_out) {
#pragma unroll
                for (int c = 0; c < 9; ++c) gradV_out[(int64_t)c * Np + p] = gv[c];
            }
            Mat3<T> A, Fo;
#pragma unroll
            for (int c = 0; c < 9; ++c) A.a[c] = dt * gv[c] + ((c % 4 == 0) ? (T)1 : (T)0), Fo.a[c] = p == p0 ? fpre[c] : Fn[(int64_t)c * Np + p];
            Fnew = m3_mul(A, Fo);
        }
        T mu = Mu[p], la = Lam[p];
        T vol = Vol[p];
        if constexpr (ENERGY_ONLY) {
            e += (double)(vol * corotated_psi_sigma(Fnew, mu, la));
            continue;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) Ft[(int64_t)c * Np + p] = Fnew.a[c];
        T psi, psis;
        Mat3<T> P;
        corotated_state(Fnew, mu, la, psi, P, &psis);
        e += (double)(vol * psi);
        es += (double)(vol * psis); // the sum an energy-only trial of the next line search is compared with: the same formula, the same singular values
        // stress = V_p P Fn^T  (Fn re-read after the SVD instead of being kept live across it)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            T f0 = Fn[(int64_t)(0 * 3 + c) * Np + p], f1 = Fn[(int64_t)(1 * 3 + c) * Np + p], f2 = Fn[(int64_t)(2 * 3 + c) * Np + p]; // Fn(c, 0..2)
#pragma unroll
            for (int r = 0; r < 3; ++r) stress_out[(int64_t)(c * 3 + r) * Np + p] = vol * (P(r, 0) * f0 + P(r, 1) * f1 + P(r, 2) * f2);
        }
    }
    double tot = block_sum_256<double>(e, red);
    if constexpr (ENERGY_ONLY)
        grid_sum_store(tot, 0.0, 1, gr, energy, nullptr, red);
    else {
        const double tots = block_sum_256<double>(es, red);
        grid_sum_store(tot, tots, 2, gr, energy, energy + 3, red);
    }
}

// K line-search trials in ONE pass (round 5): the energies at dv0 + alpha_k ddv, alpha_k = alpha / 2^k.  grad v is linear in the nodal values and the
// trial deformation gradient affine in alpha:  F(alpha) = (I + dt grad(vn + dv0 + alpha ddv)) Fn = A0 + alpha A1,  A0 = (I + dt G0) Fn, A1 = dt G1 Fn with
// G0 = grad(vn + dv0), G1 = grad(ddv) — TWO gathers per particle (two node tiles in LDS) and two 3 x 3 products whatever K is, then 9 multiply-adds and the
// SVD-free psi (corotated_psi_sigma) per trial: ~250 instructions a trial instead of the ~900 of a pass that gathers its own gradient.  G0 and A0 are the
// full pass's own numbers (same tile values, same gather, same product), so F(alpha) -> the base point's F bit for bit as alpha -> 0 and a trial's energy
// tends to Ek_sigma of the base point exactly (what line_search compares with).  Against a single energy-only pass of the same trial point (sharded runs,
// A/B switch HOT_LS_NO_BATCH) F differs by rounding — eps |F|, the level at which the accepted point's own full pass differs as well.
// lineSearch (ImplicitSolver.h:312-333) halves until the energy has not risen; at the stiffness of C3 - C5 that is 4 - 8 trials an iteration, each of
// which used to be a launch bound by its workgroups' chains of dependent round trips (header -> tile ids -> nodal values -> barrier -> particles ->
// block sum -> deposit), not by arithmetic or bytes: K trials per chain.
template <class T, int K>
struct TrialAlphas {
    T a[K];
};
template <class T, int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_state_trials(const T* __restrict__ X, const T* __restrict__ Fn, const T* __restrict__ Vol,
    const T* __restrict__ Mu, const T* __restrict__ Lam, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin, const int32_t* __restrict__ gIdx,
    const T* __restrict__ vn, const T* __restrict__ dv0, const T* __restrict__ ddv, TrialAlphas<T, K> al, T one_over_dx, T dt, double* energy, GridRed gr)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    __shared__ T nv[2][3][TILE]; // [0]: vn + dv0 (the base point, as the full pass stages it), [1]: ddv
    __shared__ double red[4 * K];
    __shared__ double esum[K][256]; // the thread's K running sums (in registers they are 2 K VGPRs live across everything: K = 16 would spill)
    const int g = blockIdx.x;
    const int first = group_first[g], last = group_first[g + 1];
    for (int t = threadIdx.x; t < TILE; t += 256) {
        const int idx = gIdx[(int64_t)g * TILE + t];
        T a = 0, b = 0, c = 0, d0 = 0, d1 = 0, d2 = 0;
        if (idx >= 0) {
            a = vn[3 * idx] + dv0[3 * idx], b = vn[3 * idx + 1] + dv0[3 * idx + 1], c = vn[3 * idx + 2] + dv0[3 * idx + 2]; // k_state's expression
            d0 = ddv[3 * idx], d1 = ddv[3 * idx + 1], d2 = ddv[3 * idx + 2];
        }
        nv[0][0][t] = a, nv[0][1][t] = b, nv[0][2][t] = c;
        nv[1][0][t] = d0, nv[1][1][t] = d1, nv[1][2][t] = d2;
    }
    __syncthreads();
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
#pragma unroll
    for (int k = 0; k < K; ++k) esum[k][threadIdx.x] = 0;
    for (int p = first + threadIdx.x; p < last; p += 256) {
        Mat3<T> A0, A1;
        {
            T xp[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) xp[d] = X[(int64_t)d * Np + p];
            int base[3];
            T w[3][3], dw[3][3];
#pragma unroll
            for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xp[d], base[d], w[d], dw[d]);
            const int t0 = (((base[0] - ox) * TY) + (base[1] - oy)) * TZ + (base[2] - oz);
            Mat3<T> Fo;
#pragma unroll
            for (int c = 0; c < 9; ++c) Fo.a[c] = Fn[(int64_t)c * Np + p];
            T gv[9];
            gather_grad(nv[0][0], nv[0][1], nv[0][2], t0, TY * TZ, TZ, w, dw, one_over_dx, gv);
            Mat3<T> A;
#pragma unroll
            for (int c = 0; c < 9; ++c) A.a[c] = dt * gv[c] + ((c % 4 == 0) ? (T)1 : (T)0); // k_state's expressions: A0 is the base point's F
            A0 = m3_mul(A, Fo);
            // (the second gather after the first: its weight products recomputed — shared, they would be 81 live values — behind an empty asm)
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(w[d][i]), "+v"(dw[d][i]));
            gather_grad(nv[1][0], nv[1][1], nv[1][2], t0, TY * TZ, TZ, w, dw, one_over_dx, gv);
#pragma unroll
            for (int c = 0; c < 9; ++c) A.a[c] = dt * gv[c];
            A1 = m3_mul(A, Fo);
        }
        const T mu = Mu[p], la = Lam[p], vol = Vol[p];
#pragma unroll 1 // one trial after the other: the Newton iterations of K trials interleaved would only cost registers
        for (int k = 0; k < K; ++k) {
            T ak = al.a[0];
#pragma unroll
            for (int kk = 1; kk < K; ++kk) ak = kk == k ? al.a[kk] : ak;
            Mat3<T> Fk;
#pragma unroll
            for (int c = 0; c < 9; ++c) Fk.a[c] = A0.a[c] + ak * A1.a[c];
            esum[k][threadIdx.x] += (double)(vol * corotated_psi_sigma(Fk, mu, la));
        }
    }
    double e[K];
#pragma unroll
    for (int k = 0; k < K; ++k) e[k] = esum[k][threadIdx.x];
    block_sum_256_n<K>(e, [](int) { return true; }, red);
    grid_sum_store_n<K>(e, [](int k) { return k; }, K, gr, energy, red);
}

// the inertia / gravity sums of k_inertia_energy for the K trial points (k_combine's trial point, the same sums in the same order: bit-identical)
template <class T, int K>
__global__ __launch_bounds__(256) void k_inertia_energy_trials(const T* __restrict__ dv0, const T* __restrict__ ddv, TrialAlphas<T, K> al, const T* __restrict__ mass, int nn, T g0, T g1, T g2,
    double* out /*[2 K]: K kinetic sums, K gravity sums*/, GridRed gr, const uint8_t* __restrict__ mask /*the rows this rank owns (sharded, halo mode), else null*/)
{
    __shared__ double red[8 * K];
    double s[2 * K];
#pragma unroll
    for (int k = 0; k < 2 * K; ++k) s[k] = 0;
    const int stride = gridDim.x * 256;
    for (int n0 = blockIdx.x * 256 + threadIdx.x; n0 < nn; n0 += 4 * stride) {
        T b[4][3], d[4][3], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + u * stride < nn ? n0 + u * stride : n0;
#pragma unroll
            for (int c = 0; c < 3; ++c) b[u][c] = dv0[3 * n + c], d[u][c] = ddv[3 * n + c];
            m[u] = mass[n];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (n0 + u * stride < nn && (!mask || mask[n0 + u * stride])) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const T a0 = b[u][0] + d[u][0] * al.a[k], a1 = b[u][1] + d[u][1] * al.a[k], a2 = b[u][2] + d[u][2] * al.a[k]; // k_combine's expression
                    s[k] += (double)((a0 * a0 + a1 * a1 + a2 * a2) * m[u]);
                    s[K + k] += (double)((g0 * a0 + g1 * a1 + g2 * a2) * m[u]);
                }
            }
    }
    block_sum_256_n<2 * K>(s, [](int) { return true; }, red);
    grid_sum_store_n<2 * K>(s, [](int k) { return k; }, 2 * K, gr, out, red);
}

// energies of the K trial points dv0 + (alpha / 2^k) ddv, k = 0 .. K - 1: what K calls of state_pass(.., energy_only = true) return, up to the rounding of F (above)
template <class T>
void Ctx<T>::trial_batch(const T* ddv, T alpha, int K, double* Ek_out)
{
    HOT_CHECK(K == 2 || K == 4 || K == 8 || K == 16, HOT_ERR_INVALID, "trial_batch: 2 / 4 / 8 / 16 trials");
    if (halo_mode()) { // the base point and the direction at the nodes of this rank's particle tiles that other ranks own: two exchanges for K trials (a single pass: one per trial)
        halo_gather(*levels[0], dv0.p);
        halo_gather(*levels[0], const_cast<T*>(ddv));
    }
    constexpr int S0 = 140, S1 = 160; // dscal / hscal slots: K <= 16 strain energies, then K kinetic + K gravity sums
    const int grid = std::min(div_up(Nn, 1024), 1024);
    auto run = [&](auto kc) {
        constexpr int KK = decltype(kc)::value;
        TrialAlphas<T, KK> al;
        T a = alpha;
        for (int k = 0; k < KK; ++k) al.a[k] = a, a *= (T)0.5;
        GridRed g1 = gred_n(Ng, KK);
        g1.mirror = hscal + S0;
        HOT_LAUNCH(this, KK == 2 ? "state_trials2" : (KK == 4 ? "state_trials4" : (KK == 8 ? "state_trials8" : "state_trials16")), (k_state_trials<T, KK>), Ng, 256, 0, pX.p, pFn.p, pVol.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, tileDof.p, vn.p, dv0.p, ddv, al, (T)1 / dx, dt,
            dscal.p + S0, g1);
        GridRed g2 = gred_n(grid, 2 * KK);
        g2.mirror = hscal + S1, g2.ticket = hscal + 251, g2.ticket_val = new_ticket();
        HOT_LAUNCH(this, "inertia_trials", (k_inertia_energy_trials<T, KK>), grid, 256, 0, dv0.p, ddv, al, mass.p, Nn, (T)cfg.gravity[0], (T)cfg.gravity[1], (T)cfg.gravity[2], dscal.p + S1, g2, vmask);
    };
    if (K == 2) run(std::integral_constant<int, 2>());
    else if (K == 4) run(std::integral_constant<int, 4>());
    else if (K == 8) run(std::integral_constant<int, 8>());
    else run(std::integral_constant<int, 16>());
    wait_ticket();
    if (sharded()) { // state_pass's sums over the ranks, for the K trials at once: the shards' strain energies; in halo mode the inertia terms of the rows every rank owns too
        double buf[48];
        const int nb = halo_mode() ? 3 * K : K;
        for (int k = 0; k < K; ++k) buf[k] = hscal[S0 + k];
        for (int k = 0; k < 2 * K; ++k) buf[K + k] = hscal[S1 + k];
        c_allreduce(buf, nb, HOT_COMM_F64, HOT_COMM_SUM, false);
        for (int k = 0; k < K; ++k) hscal[S0 + k] = buf[k];
        if (halo_mode())
            for (int k = 0; k < 2 * K; ++k) hscal[S1 + k] = buf[K + k];
    }
    for (int k = 0; k < K; ++k) {
        double r = (double)(T)hscal[S0 + k]; // state_pass's assembly
        r += hscal[S1 + k] / 2;
        r -= (double)dt * hscal[S1 + K + k];
        Ek_out[k] = r;
    }
}

#ifdef HOT_AB_KERNELS
#include "ab_src/force_ab1.hip"
#endif

#ifdef HOT_AB_KERNELS
#include "ab_src/force_ab2.hip"
#endif

template <class T>
__global__ __launch_bounds__(256) void k_inertia_energy(const T* __restrict__ dv, const T* __restrict__ mass, int nn, T g0, T g1, T g2, double* out, GridRed gr, const uint8_t* __restrict__ mask)
{
    __shared__ double red[4];
    double ke = 0, ge = 0;
    const int stride = gridDim.x * 256;
    for (int n0 = blockIdx.x * 256 + threadIdx.x; n0 < nn; n0 += 4 * stride) { // four strided nodes per trip in flight
        T a[4], b[4], c[4], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + u * stride < nn ? n0 + u * stride : n0;
            a[u] = dv[3 * n], b[u] = dv[3 * n + 1], c[u] = dv[3 * n + 2], m[u] = mass[n];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (n0 + u * stride < nn && (!mask || mask[n0 + u * stride])) { // mask: the rows this rank owns (sharded, halo mode)
                ke += (double)((a[u] * a[u] + b[u] * b[u] + c[u] * c[u]) * m[u]);
                ge += (double)((g0 * a[u] + g1 * b[u] + g2 * c[u]) * m[u]);
            }
    }
    double k = block_sum_256<double>(ke, red);
    double gg = block_sum_256<double>(ge, red);
    grid_sum_store(k, gg, 2, gr, out, out + 1, red);
}

// Production force scatter: items (cell segment, node row j, half of the segment) -> the 9 nodes (i, k) of that row with their
// 27 sums in registers, the 1-D weights and derivatives recomputed from x per (item, particle) — 12 staged scalars per particle (stress,
// x) read from LDS once per 9 nodes, where k_force_cells reads 19 of its 27 staged ones per 3 nodes (the item phase of these kernels is
// bound by LDS reads + VALU issue, profiles/r03_sq_counters_C2.json).  25 KB of LDS per 256 fp64 particles: 256-thread workgroups,
// six per CU.
template <class T>
__global__ __launch_bounds__(256) void k_force_cells2(const T* __restrict__ X, const T* __restrict__ stress, int64_t Np, const int32_t* __restrict__ group_first,
    const int32_t* __restrict__ group_origin, const int32_t* __restrict__ group_cell0, const int32_t* __restrict__ cell_first, T* __restrict__ part, T one_over_dx, T scale)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ, THREADS = 256;
    constexpr int CH = sizeof(T) == 4 ? 512 : 256;
    using AT = AccT<T>; // double tile also in fp32 (see k_force_cells)
    __shared__ AT acc[3][TILE];
    __shared__ T sp[12][CH]; // scale * S(9), x(3)
    __shared__ int32_t segs[G::EPB + 2];
    __shared__ int32_t nseg;
    const int g = blockIdx.x, tid = threadIdx.x;
    for (int t = tid; t < 3 * TILE; t += THREADS) (&acc[0][0])[t] = (AT)0;
    const int first = group_first[g], last = group_first[g + 1];
    const int c0 = group_cell0[g], c1 = group_cell0[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    for (int ch = first; ch < last; ch += CH) {
        if (tid == 0) nseg = 0;
        __syncthreads(); // also orders the previous chunk's reads of sp / segs before they are overwritten
        for (int l = tid; l < CH && ch + l < last; l += THREADS) {
            const int p = ch + l;
#pragma unroll
            for (int c = 0; c < 9; ++c) sp[c][l] = scale * stress[(int64_t)c * Np + p];
#pragma unroll
            for (int d = 0; d < 3; ++d) sp[9 + d][l] = X[(int64_t)d * Np + p];
        }
        for (int c = c0 + tid; c < c1; c += THREADS) {
            const int s0 = max(cell_first[c], ch), s1 = min(cell_first[c + 1], min(ch + CH, last));
            if (s1 > s0) segs[atomicAdd(&nseg, 1)] = (s0 - ch) | ((s1 - ch) << 16);
        }
        __syncthreads();
        const int ni = nseg * 6;
        for (int it = tid; it < ni; it += THREADS) {
            const int sd = segs[it / 6], j = (it % 6) >> 1, hf = it & 1, s0 = sd & 0xffff, s1 = sd >> 16;
            const int mid = (s0 + s1 + 1) >> 1, l0 = hf ? mid : s0, l1 = hf ? s1 : mid;
            if (l0 >= l1) continue;
            T a[3][3][3]; // [i][k][component]
#pragma unroll
            for (int e = 0; e < 27; ++e) (&a[0][0][0])[e] = (T)0;
            int b0 = 0, b1 = 0, b2 = 0; // the same for every particle of the cell
            for (int l = l0; l < l1; ++l) {
                T wx[3], dwx[3], wy3[3], dwy3[3], wz[3], dwz[3];
                bspline<T>(one_over_dx, sp[9][l], b0, wx, dwx);
                bspline<T>(one_over_dx, sp[10][l], b1, wy3, dwy3);
                bspline<T>(one_over_dx, sp[11][l], b2, wz, dwz);
                const T wy = j == 0 ? wy3[0] : (j == 1 ? wy3[1] : wy3[2]), dwy = j == 0 ? dwy3[0] : (j == 1 ? dwy3[1] : dwy3[2]);
                T S[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) S[c] = sp[c][l];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const T wi = wx[i], dwi = one_over_dx * dwx[i];
                    const T wij = wi * wy, dwij_i = dwi * wy, dwij_j = wi * one_over_dx * dwy;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const T g0 = dwij_i * wz[k], g1 = dwij_j * wz[k], g2 = wij * one_over_dx * dwz[k];
                        a[i][k][0] += -(S[0] * g0 + S[3] * g1 + S[6] * g2);
                        a[i][k][1] += -(S[1] * g0 + S[4] * g1 + S[7] * g2);
                        a[i][k][2] += -(S[2] * g0 + S[5] * g1 + S[8] * g2);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int t = ((b0 - ox + i) * TY + (b1 - oy + j)) * TZ + (b2 - oz + k);
                    lds_atomic_add(&acc[0][t], (AT)a[i][k][0]), lds_atomic_add(&acc[1][t], (AT)a[i][k][1]), lds_atomic_add(&acc[2][t], (AT)a[i][k][2]);
                }
        }
    }
    __syncthreads();
    T* out = part + (int64_t)g * 3 * TILE; // partial tile, summed per node by k_tile_reduce
    for (int t = tid; t < 3 * TILE; t += THREADS) out[t] = (T)(&acc[0][0])[t];
}

// rasterizeForceToTVStack from the stresses the last k_state left behind (the line search needs it once, at the accepted
// point: lineSearch evaluates only the energy per trial, ImplicitSolver.h:312-333)
template <class T>
void Ctx<T>::force_pass()
{
    int64_t slots = (int64_t)Nb * EPB;
#ifdef HOT_AB_KERNELS
    if (ab_flag("HOT_FORCE_V1")) // one LDS atomic per particle, node and component
        HOT_LAUNCH(this, "force_scatter", k_force_scatter<T>, Ng, 256, 0, pX.p, pStress.p, Np, group_first.p, group_origin.p, group_nb.p, gPart.p, (T)1 / dx, dt);
    else if (ab_flag("HOT_FORCE_CELLS1")) // 27 staged scalars, 3-node items
        HOT_LAUNCH(this, "force_scatter", k_force_cells<T>, Ng, FORCE_THREADS, 0, pX.p, pStress.p, Np, group_first.p, group_origin.p, group_cell0.p, cell_first.p, gPart.p, (T)1 / dx, dt);
    else
#endif
        HOT_LAUNCH(this, "force_scatter", k_force_cells2<T>, Ng, 256, 0, pX.p, pStress.p, Np, group_first.p, group_origin.p, group_cell0.p, cell_first.p, gPart.p, (T)1 / dx, dt);
    reduce_tiles(3, gF.p, gF.p + slots, gF.p + 2 * slots, nullptr, nullptr, "force_reduce");
    if (halo_mode()) {
        T* arr[3] = { gF.p, gF.p + slots, gF.p + 2 * slots };
        tile_exchange(arr, 3); // the ranks that share a block add their partial forces; every block this rank covers is complete afterwards
    }
    else if (sharded())
        allreduce_tiles(gF.p, 3);
}

template <class T>
double Ctx<T>::state_pass(const T* dv_in, bool want_force, bool energy_only)
{
    if (halo_mode()) halo_gather(*levels[0], const_cast<T*>(dv_in)); // dv at the nodes of this rank's particle tiles that other ranks own
    if (energy_only) // a line-search trial: nothing but the energy leaves the kernel (trial F and stresses keep those of the last full pass)
        HOT_LAUNCH(this, "state_energy", (k_state<T, true>), Ng, 256, 0, pX.p, pFn.p, pVol.p, pMu.p, pLam.p, (T*)nullptr, (T*)nullptr, (T*)nullptr, Np, group_first.p, group_origin.p,
            group_nb.p, tileDof.p, vn.p, dv_in, dx, (T)1 / dx, dt, dscal.p, gred(Ng, hscal));
    else
        HOT_LAUNCH(this, "state_update", (k_state<T, false>), Ng, 256, 0, pX.p, pFn.p, pVol.p, pMu.p, pLam.p, pFt.p, pStress.p, keep_debug ? pGradV.p : (T*)nullptr, Np, group_first.p, group_origin.p,
            group_nb.p, tileDof.p, vn.p, dv_in, dx, (T)1 / dx, dt, dscal.p, gred2(Ng, hscal, hscal + 3)); // the sums land in the pinned host slots too: one stream sync, no copy
    if (want_force) force_pass();
    {
        const int grid = std::min(div_up(Nn, 1024), 1024);
        HOT_LAUNCH(this, "inertia_energy", k_inertia_energy<T>, grid, 256, 0, dv_in, mass.p, Nn, (T)cfg.gravity[0], (T)cfg.gravity[1], (T)cfg.gravity[2], dscal.p + 1, gred(grid, hscal + 1, true), vmask);
    }
    wait_ticket();
    if (energy_only) hscal[3] = hscal[0]; // (an energy-only pass has the one sum)
    if (halo_mode())
        c_allreduce(hscal, 4, HOT_COMM_F64, HOT_COMM_SUM, false); // strain energy of the shards (both summation formulas), inertia terms of the rows every rank owns
    else if (sharded()) {
        double two[2] = { hscal[0], hscal[3] };
        c_allreduce(two, 2, HOT_COMM_F64, HOT_COMM_SUM, false); // the shards' strain energies; the inertia terms are computed from replicated vectors
        hscal[0] = two[0], hscal[3] = two[1];
    }
    double result = (double)(T)hscal[0];
    result += hscal[1] / 2;
    result -= (double)dt * hscal[2];
    if (!energy_only) { // the same total with psi summed from the singular values: what the energy-only trials of the next line search are compared with
        Ek_sigma = (double)(T)hscal[3];
        Ek_sigma += hscal[1] / 2;
        Ek_sigma -= (double)dt * hscal[2];
    }
    return result;
}

template <class T>
void Ctx<T>::update_state(const void* dv_in, double* energy)
{
    need(Nn > 0 && dt > 0, "hot_update_state before hot_begin_step");
    if (dv_in) HOT_HIP(hipMemcpyAsync(dv.p, dv_in, 3 * (size_t)Nn * sizeof(T), hipMemcpyDefault, stream));
    Ek = state_pass(dv.p, true);
    if (energy) *energy = Ek;
}

template <class T>
void Ctx<T>::get_particle_state(void* F, void* stress, void* gradV)
{
    need(Np > 0, "no particles");
    int64_t n = Np;
    DBuf<T> tmp;
    tmp.reserve(9 * n);
    auto out = [&](const T* src, void* dst) {
        if (!dst) return;
        HOT_HIP(hipMemcpyAsync(tmp.p, src, 9 * n * sizeof(T), hipMemcpyDeviceToDevice, stream));
        // scatter to original order on the host side of the ABI: reuse get_particles machinery through spare9
        std::vector<T> h(9 * n), o(9 * n);
        std::vector<int32_t> s2o(n);
        HOT_HIP(hipMemcpyAsync(h.data(), tmp.p, 9 * n * sizeof(T), hipMemcpyDeviceToHost, stream));
        HOT_HIP(hipMemcpyAsync(s2o.data(), slot2orig.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        sync();
        for (int64_t p = 0; p < n; ++p)
            for (int c = 0; c < 9; ++c) o[(int64_t)s2o[p] * 9 + c] = h[(int64_t)c * n + p];
        HOT_HIP(hipMemcpy(dst, o.data(), 9 * n * sizeof(T), hipMemcpyDefault));
    };
    out(pFt.p, F), out(pStress.p, stress), out(pGradV.p, gradV);
}

// ------------------------------------------------------------------------------------------------ residual / projection
template <class T>
__device__ __forceinline__ void mat3_vec(const T* __restrict__ M, const T* v, T* o) // column-major
{
    o[0] = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    o[1] = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    o[2] = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
}
// v <- projected v at a collision node (project lambda, MultigridSimulation.h:105-124)
template <class T>
__device__ __forceinline__ void bc_project(int c, const T* __restrict__ P, const uint8_t* __restrict__ slip, bool slipmode, T* v)
{
    if (slipmode) {
        if (slip[c])
            v[0] = (T)0;
        else
            v[0] = v[1] = v[2] = (T)0;
    }
    else {
        T o[3];
        mat3_vec(P + 9 * c, v, o);
        v[0] = o[0], v[1] = o[1], v[2] = o[2];
    }
}

template <class T>
__global__ void k_residual(const T* __restrict__ gF, const int32_t* __restrict__ dofSlot, const T* __restrict__ mass, const T* __restrict__ dv, const int32_t* __restrict__ bcIdx,
    const T* __restrict__ bcP, const T* __restrict__ bcR, const uint8_t* __restrict__ bcSlip, T* r, T* r2 /*a second copy (the right-hand side kept for the exit test)*/, int nn, int64_t slots, T g0, T g1, T g2, T dt, int slipmode)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    int64_t s = dofSlot[n];
    T m = mass[n];
    T v[3] = { g0 * dt * m + gF[s] - dv[3 * n] * m, g1 * dt * m + gF[slots + s] - dv[3 * n + 1] * m, g2 * dt * m + gF[2 * slots + s] - dv[3 * n + 2] * m };
    int c = bcIdx[n];
    if (c >= 0) {
        if (slipmode && bcSlip[c]) { // transformResidual
            T o[3];
            mat3_vec(bcR + 9 * c, v, o);
            v[0] = o[0], v[1] = o[1], v[2] = o[2];
        }
        bc_project(c, bcP, bcSlip, slipmode != 0, v);
    }
    r[3 * n] = v[0], r[3 * n + 1] = v[1], r[3 * n + 2] = v[2];
    r2[3 * n] = v[0], r2[3 * n + 1] = v[1], r2[3 * n + 2] = v[2];
}
template <class T>
__global__ void k_project(const int32_t* __restrict__ bcNode, const T* __restrict__ bcP, const uint8_t* __restrict__ bcSlip, T* v, int nc, int slipmode)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    int n = bcNode[c];
    T x[3] = { v[3 * n], v[3 * n + 1], v[3 * n + 2] };
    bc_project(c, bcP, bcSlip, slipmode != 0, x);
    v[3 * n] = x[0], v[3 * n + 1] = x[1], v[3 * n + 2] = x[2];
}
template <class T>
__global__ void k_transform(const int32_t* __restrict__ bcNode, const T* __restrict__ M, const uint8_t* __restrict__ bcSlip, T* v, int nc)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc || !bcSlip[c]) return;
    int n = bcNode[c];
    T x[3] = { v[3 * n], v[3 * n + 1], v[3 * n + 2] }, o[3];
    mat3_vec(M + 9 * c, x, o);
    v[3 * n] = o[0], v[3 * n + 1] = o[1], v[3 * n + 2] = o[2];
}

template <class T>
void Ctx<T>::residual_dev(T* r)
{
    int slipmode = (cfg.systemBCProject && cfg.boundaryType == 1) ? 1 : 0;
    HOT_LAUNCH(this, "residual", k_residual<T>, div_up(Nn, 256), 256, 0, gF.p, dofSlot.p, mass.p, dv.p, bcIdx.p, bcP.p, bcR.p, bcSlip.p, r, rhs.p, Nn, (int64_t)Nb * EPB, (T)cfg.gravity[0],
        (T)cfg.gravity[1], (T)cfg.gravity[2], dt, slipmode);
}
template <class T>
void Ctx<T>::project_dev(T* v)
{
    if (Nc == 0) return;
    int slipmode = (cfg.systemBCProject && cfg.boundaryType == 1) ? 1 : 0;
    HOT_LAUNCH(this, "project", k_project<T>, div_up(Nc, 256), 256, 0, bcNode.p, bcP.p, bcSlip.p, v, Nc, slipmode);
}
template <class T>
void Ctx<T>::transform_dev(T* v, bool inverse)
{
    if (Nc == 0 || !(cfg.systemBCProject && cfg.boundaryType == 1)) return;
    HOT_LAUNCH(this, "transform", k_transform<T>, div_up(Nc, 256), 256, 0, bcNode.p, inverse ? bcRinv.p : bcR.p, bcSlip.p, v, Nc);
}

template <class T>
void Ctx<T>::residual(void* r)
{
    need(Nn > 0 && dt > 0, "hot_residual before hot_begin_step/hot_update_state");
    residual_dev(work0.p);
    if (halo_mode()) gather_all(*levels[0], work0.p); // the C ABI hands out complete vectors
    download(r, work0.p, 3 * (size_t)Nn);
    sync();
}
template <class T>
void Ctx<T>::project(void* v)
{
    need(Nn > 0, "no grid");
    HOT_HIP(hipMemcpyAsync(work0.p, v, 3 * (size_t)Nn * sizeof(T), hipMemcpyDefault, stream));
    project_dev(work0.p);
    download(v, work0.p, 3 * (size_t)Nn);
    sync();
}

// ------------------------------------------------------------------------------------------------ CN tolerance
template <class T>
__global__ void k_cn_tol(const T* __restrict__ gCN, const int32_t* __restrict__ dofSlot, const T* __restrict__ mass, T* tol, int nn, T scale)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < nn) tol[n] = gCN[dofSlot[n]] * scale / mass[n];
}
template <class T>
__global__ __launch_bounds__(256) void k_max_dpdf(const T* __restrict__ Mu, const T* __restrict__ Lam, int64_t np, unsigned long long* out)
{
    __shared__ double red[4];
    double mx = 0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < np; p += (int64_t)gridDim.x * 256) {
        T mu = Mu[p], la = Lam[p];
        double v = (double)hsqrt((T)3 * ((T)2 * mu + la) * ((T)2 * mu + la) + (T)6 * la * la + (T)12 * mu * mu);
        mx = v > mx ? v : mx;
    }
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_xor(mx, o, 64);
        mx = other > mx ? other : mx;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) mx = red[k] > mx ? red[k] : mx;
        atomicMax(out, (unsigned long long)__double_as_longlong(mx)); // non-negative doubles order like integers
    }
}
template <class T>
void Ctx<T>::cn_tolerance_dev()
{
    T scale = (T)cfg.cneps * 24 * dx * dx * dt;
    HOT_LAUNCH(this, "cn_tolerance", k_cn_tol<T>, div_up(Nn, 256), 256, 0, gCN.p, dofSlot.p, mass.p, cnTol.p, Nn, scale);
    HOT_HIP(hipMemsetAsync(dscal.p + 8, 0, sizeof(double), stream));
    HOT_LAUNCH(this, "max_dpdf_norm", k_max_dpdf<T>, std::min(div_up(Np, 256), 1024), 256, 0, pMu.p, pLam.p, Np, (unsigned long long*)(dscal.p + 8));
    HOT_HIP(hipMemcpyAsync(hscal + 8, dscal.p + 8, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    if (sharded()) c_allreduce(hscal + 8, 1, HOT_COMM_F64, HOT_COMM_MAX, false);
    max_cn_tolerance = (T)cfg.cneps * dt * 24 * (T)std::sqrt((double)Nn) * dx * dx * (T)hscal[8];
}
template <class T>
void Ctx<T>::cn_tolerance(void* tol)
{
    need(Nn > 0 && dt > 0, "hot_cn_tolerance before hot_begin_step");
    need(cfg.useCN, "hot_cn_tolerance needs cfg.useCN (the accumulation is fused into P2G)");
    cn_tolerance_dev();
    if (halo_mode()) gather_all(*levels[0], cnTol.p, 1);
    download(tol, cnTol.p, Nn);
    sync();
}

// ------------------------------------------------------------------------------------------------ matrix-free product
template <class T>
__global__ __launch_bounds__(256) void k_matfree(const T* __restrict__ X, const T* __restrict__ Fn, const T* __restrict__ Ft, const T* __restrict__ Vol, const T* __restrict__ Mu,
    const T* __restrict__ Lam, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin, const int32_t* __restrict__ group_nb,
    const int32_t* __restrict__ gIdx, const T* __restrict__ x, T* __restrict__ part, T dx, T one_over_dx, T dt, int project)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    __shared__ T nv[3][TILE];
    using AT = AccT<T>; // double tile also in fp32 (LDS float atomics are slow, see k_force_cells)
    __shared__ AT acc[3][TILE];
    __shared__ int32_t nb8[8];
    const int g = blockIdx.x;
    if (threadIdx.x < 8) nb8[threadIdx.x] = group_nb[g * 8 + threadIdx.x];
    __syncthreads();
    for (int t = threadIdx.x; t < TILE; t += 256) {
        int idx = gIdx[tile_slot2<T>(t, nb8)];
        T a = 0, b = 0, c = 0;
        if (idx >= 0) a = x[3 * idx], b = x[3 * idx + 1], c = x[3 * idx + 2];
        nv[0][t] = a, nv[1][t] = b, nv[2][t] = c;
        acc[0][t] = acc[1][t] = acc[2][t] = (AT)0;
    }
    __syncthreads();
    const int first = group_first[g], last = group_first[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    for (int p = first + threadIdx.x; p < last; p += 256) {
        T xp[3] = { X[p], X[Np + p], X[2 * Np + p] };
        int base[3];
        T w[3][3], dw[3][3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xp[d], base[d], w[d], dw[d]);
        const int cx = base[0] - ox, cy = base[1] - oy, cz = base[2] - oz;
        Mat3<T> gx;
#pragma unroll
        for (int c = 0; c < 9; ++c) gx.a[c] = (T)0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    T g0 = one_over_dx * dw[0][i] * w[1][j] * w[2][k], g1 = w[0][i] * one_over_dx * dw[1][j] * w[2][k], g2 = w[0][i] * w[1][j] * one_over_dx * dw[2][k];
                    int t = ((cx + i) * TY + (cy + j)) * TZ + (cz + k);
                    T v0 = nv[0][t], v1 = nv[1][t], v2 = nv[2][t];
                    gx.a[0] += v0 * g0, gx.a[1] += v1 * g0, gx.a[2] += v2 * g0;
                    gx.a[3] += v0 * g1, gx.a[4] += v1 * g1, gx.a[5] += v2 * g1;
                    gx.a[6] += v0 * g2, gx.a[7] += v1 * g2, gx.a[8] += v2 * g2;
                }
        Mat3<T> Fo, Fc;
#pragma unroll
        for (int c = 0; c < 9; ++c) Fo.a[c] = Fn[(int64_t)c * Np + p], Fc.a[c] = Ft[(int64_t)c * Np + p];
        HessBlocks<T> h;
        corotated_hessian(Fc, Mu[p], Lam[p], project != 0, h);
        Mat3<T> dP = hess_apply(h, m3_mul(gx, Fo));
        T vol = Vol[p];
        Mat3<T> S;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) S(r, c) = vol * (dP(r, 0) * Fo(c, 0) + dP(r, 1) * Fo(c, 1) + dP(r, 2) * Fo(c, 2));
        T sc = dt * dt; // -(scale) with scale = -dt^2
        for (int n = 0; n < 27; ++n) {
            int i = n / 9, j = (n / 3) % 3, k = n % 3;
            T g0 = one_over_dx * dw[0][i] * w[1][j] * w[2][k], g1 = w[0][i] * one_over_dx * dw[1][j] * w[2][k], g2 = w[0][i] * w[1][j] * one_over_dx * dw[2][k];
            int t = ((cx + i) * TY + (cy + j)) * TZ + (cz + k);
            lds_atomic_add(&acc[0][t], (AT)(sc * (S(0, 0) * g0 + S(0, 1) * g1 + S(0, 2) * g2)));
            lds_atomic_add(&acc[1][t], (AT)(sc * (S(1, 0) * g0 + S(1, 1) * g1 + S(1, 2) * g2)));
            lds_atomic_add(&acc[2][t], (AT)(sc * (S(2, 0) * g0 + S(2, 1) * g1 + S(2, 2) * g2)));
        }
    }
    __syncthreads();
    T* out = part + (int64_t)g * 3 * TILE;
    for (int t = threadIdx.x; t < 3 * TILE; t += 256) out[t] = (T)(&acc[0][0])[t];
}
template <class T>
__global__ void k_matfree_finish(const T* __restrict__ gOut, const int32_t* __restrict__ dofSlot, const T* __restrict__ mass, const T* __restrict__ x, T* y, int nn, int64_t slots)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    int64_t s = dofSlot[n];
    T m = mass[n];
    y[3 * n] = m * x[3 * n] + gOut[s], y[3 * n + 1] = m * x[3 * n + 1] + gOut[slots + s], y[3 * n + 2] = m * x[3 * n + 2] + gOut[2 * slots + s];
}

template <class T>
void Ctx<T>::matfree_dev(const T* x, T* y)
{
    int64_t slots = (int64_t)Nb * EPB;
    DBuf<T>& tile = ap; // scratch tile array (3*slots); `ap` is otherwise only used while building the hierarchy
    tile.reserve(3 * slots);
    if (halo_mode()) halo_gather(*levels[0], const_cast<T*>(x)); // x at the nodes of this rank's particle tiles
    HOT_LAUNCH(this, "matfree_hessian_product", k_matfree<T>, Ng, 256, 0, pX.p, pFn.p, pFt.p, pVol.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_nb.p, gIdx.p, x, gPart.p, dx,
        (T)1 / dx, dt, cfg.project);
    reduce_tiles(3, tile.p, tile.p + slots, tile.p + 2 * slots, nullptr, nullptr, "matfree_reduce");
    if (halo_mode()) {
        T* arr[3] = { tile.p, tile.p + slots, tile.p + 2 * slots };
        tile_exchange(arr, 3);
    }
    else if (sharded())
        allreduce_tiles(tile.p, 3);
    HOT_LAUNCH(this, "matfree_finish", k_matfree_finish<T>, div_up(Nn, 256), 256, 0, tile.p, dofSlot.p, mass.p, x, y, Nn, slots);
}
template <class T>
void Ctx<T>::matfree_multiply(const void* x, void* y)
{
    need(Nn > 0 && dt > 0, "hot_matfree_multiply before hot_update_state");
    HOT_HIP(hipMemcpyAsync(work0.p, x, 3 * (size_t)Nn * sizeof(T), hipMemcpyDefault, stream));
    matfree_dev(work0.p, work1.p);
    if (halo_mode()) gather_all(*levels[0], work1.p);
    download(y, work1.p, 3 * (size_t)Nn);
    sync();
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
