// libhotmi355x — analytic collision objects evaluated per grid node on the device.
//
// Restates (transform x = R s X + b, velocity omega x (x-b) + (ds/dt / s)(x-b) + db/dt):
//   AnalyticCollisionObject::detectAndResolveCollision   Lib/Ziran/Math/Geometry/CollisionObject.cpp:384-447
//   AnalyticCollisionObject::multiObjectCollision (wn)    :107-148
//   HalfSpace / Sphere / AxisAlignedAnalyticBox queries   Lib/Ziran/Math/Geometry/AnalyticLevelSet.cpp:111-118,264-288,353-363,435-452,504-529
//   CappedCylinder (AnalyticLevelSet.h:221-297), Torus (AnalyticLevelSet.cpp:565-608): distance of the y-axis primitive after the
//   level set's own rotation / translation; the torus normal is the gradient the reference gets by automatic differentiation
//   RotationExtractor<T,3>::rotate (Eigen::Quaternion::setFromTwoVectors + toRotationMatrix)   Lib/MPM/MpmSimulationBase.h:271-281
// The box's normal comes from automatic differentiation of a distance that is not differentiable inside the box in the
// reference; only STICKY boxes (no normal needed) are accepted.
#pragma once
#include "hot_svd.h"
#include <algorithm>
#include <cmath>
#include "../../include/hot_mi355x.h"

namespace hot {

template <class T>
struct CollObj { // device copy of hot_collision_object in the simulation's scalar type
    int32_t shape, type;
    T p0[3], p1[3], friction, b[3], dbdt[3];
    T R[9], omega[3], inv_s, dsdt; // R column-major; inv_s = 1 / s
    T Rls[9]; // capped cylinder / torus: rotation of the level set itself (column-major)
    int32_t nmember; // UNION / DIFFERENCE: the member records that follow this one
};

// signed distance and (where the reference has a closed form: half space, sphere, torus) material-space normal of a primitive: the
// signedDistance / normal members of Lib/Ziran/Math/Geometry/AnalyticLevelSet.cpp:272-288 (HalfSpace), :403-421 (Sphere), :504-537 (boxes),
// :580-608 (Torus), AnalyticLevelSet.h:262-287 (CappedCylinder) — what DisjointUnionLevelSet / DifferenceLevelSet call on their members
template <class T>
__device__ __forceinline__ T co_signed_distance(const CollObj<T>& o, const T (&X)[3], T (&N)[3])
{
    N[0] = N[1] = N[2] = (T)0;
    const T t[3] = { X[0] - o.p0[0], X[1] - o.p0[1], X[2] - o.p0[2] };
    if (o.shape == HOT_SHAPE_HALFSPACE) {
        N[0] = o.p1[0], N[1] = o.p1[1], N[2] = o.p1[2];
        return o.p1[0] * t[0] + o.p1[1] * t[1] + o.p1[2] * t[2];
    }
    if (o.shape == HOT_SHAPE_SPHERE) {
        const T d2 = t[0] * t[0] + t[1] * t[1] + t[2] * t[2], dist = hsqrt(d2);
        if (d2 < (T)1e-7)
            N[0] = 1;
        else
            N[0] = t[0] / dist, N[1] = t[1] / dist, N[2] = t[2] / dist;
        return dist - o.p1[0];
    }
    if (o.shape == HOT_SHAPE_BOX) {
        T dd = -(T)3.4e38, q2 = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T c = (o.p0[k] + o.p1[k]) / (T)2, h = (o.p1[k] - o.p0[k]) / (T)2;
            const T d = habs(X[k] - c) - h;
            dd = d > dd ? d : dd;
            const T q = d < (T)0 ? (T)0 : d;
            q2 += q * q;
        }
        return (dd < (T)0 ? dd : (T)0) + hsqrt(q2);
    }
    T P[3]; // primitive space: R_ls^-1 (X - b_ls)
#pragma unroll
    for (int k = 0; k < 3; ++k) P[k] = o.Rls[3 * k] * t[0] + o.Rls[3 * k + 1] * t[1] + o.Rls[3 * k + 2] * t[2];
    if (o.shape == HOT_SHAPE_ROTATED_BOX) {
        T dd = -(T)3.4e38, q2 = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T d = habs(P[k]) - o.p1[k];
            dd = d > dd ? d : dd;
            const T q = d < (T)0 ? (T)0 : d;
            q2 += q * q;
        }
        return (dd < (T)0 ? dd : (T)0) + hsqrt(q2);
    }
    const T rho = hsqrt(P[0] * P[0] + P[2] * P[2]);
    if (o.shape == HOT_SHAPE_TORUS) {
        const T q0 = rho - o.p1[0], L = hsqrt(q0 * q0 + P[1] * P[1]);
        const T gr = q0 / L, G[3] = { gr * P[0] / rho, P[1] / L, gr * P[2] / rho };
#pragma unroll
        for (int k = 0; k < 3; ++k) N[k] = o.Rls[k] * G[0] + o.Rls[3 + k] * G[1] + o.Rls[6 + k] * G[2];
        return L - o.p1[1];
    }
    const T d0 = rho - o.p1[0], d1 = habs(P[1]) - (T)0.5 * o.p1[1]; // capped cylinder
    const T m0 = d0 > (T)0 ? d0 : (T)0, m1 = d1 > (T)0 ? d1 : (T)0, mx = d0 > d1 ? d0 : d1;
    return (mx < (T)0 ? mx : (T)0) + hsqrt(m0 * m0 + m1 * m1);
}

// returns whether node position x collides with o; v is replaced by the resolved velocity, n by the world normal (SLIP / SEPARATE)
template <class T>
__device__ __forceinline__ bool co_detect_resolve(const CollObj<T>* __restrict__ po, const T (&x)[3], T (&v)[3], T (&n)[3])
{
    const CollObj<T>& o = *po; // a composite's members are po[1 .. nmember]
    const T xb[3] = { x[0] - o.b[0], x[1] - o.b[1], x[2] - o.b[2] };
    T X[3], N[3] = { 0, 0, 0 }; // material space: X = R^T (x - b) / s
#pragma unroll
    for (int k = 0; k < 3; ++k) X[k] = (o.R[3 * k] * xb[0] + o.R[3 * k + 1] * xb[1] + o.R[3 * k + 2] * xb[2]) * o.inv_s;
    bool colliding = false;
    if (o.shape == HOT_SHAPE_UNION) { // DisjointUnionLevelSet::signedDistance / normal (AnalyticLevelSet.cpp:148-190)
        T best = (T)3.4e38;
        for (int m = 1; m <= o.nmember; ++m) {
            T Nm[3];
            const T d = co_signed_distance(po[m], X, Nm);
            if (d < best) best = d, N[0] = Nm[0], N[1] = Nm[1], N[2] = Nm[2];
        }
        colliding = best <= (T)0;
    }
    else if (o.shape == HOT_SHAPE_DIFFERENCE) { // DifferenceLevelSet (:220-236): A minus B
        T Na[3], Nb[3];
        const T a = co_signed_distance(po[1], X, Na), nb = -co_signed_distance(po[2], X, Nb);
        if (nb > a)
            N[0] = -Nb[0], N[1] = -Nb[1], N[2] = -Nb[2];
        else
            N[0] = Na[0], N[1] = Na[1], N[2] = Na[2];
        colliding = (a > nb ? a : nb) <= (T)0;
    }
    else if (o.shape == HOT_SHAPE_HALFSPACE) {
        const T phi = o.p1[0] * (X[0] - o.p0[0]) + o.p1[1] * (X[1] - o.p0[1]) + o.p1[2] * (X[2] - o.p0[2]);
        colliding = phi <= (T)0;
        N[0] = o.p1[0], N[1] = o.p1[1], N[2] = o.p1[2];
    }
    else if (o.shape == HOT_SHAPE_SPHERE) {
        const T t0 = X[0] - o.p0[0], t1 = X[1] - o.p0[1], t2 = X[2] - o.p0[2];
        const T d2 = t0 * t0 + t1 * t1 + t2 * t2, r2 = o.p1[0] * o.p1[0];
        if (d2 < r2) {
            colliding = true;
            const T dist = hsqrt(d2);
            if (dist < (T)1e-7)
                N[0] = 1, N[1] = 0, N[2] = 0;
            else {
                const T inv = (T)1 / dist;
                N[0] = inv * t0, N[1] = inv * t1, N[2] = inv * t2;
            }
        }
    }
    else if (o.shape == HOT_SHAPE_CAPPED_CYLINDER || o.shape == HOT_SHAPE_TORUS) {
        const T t[3] = { X[0] - o.p0[0], X[1] - o.p0[1], X[2] - o.p0[2] };
        T P[3]; // primitive space: R_ls^-1 (X - b_ls)
#pragma unroll
        for (int k = 0; k < 3; ++k) P[k] = o.Rls[3 * k] * t[0] + o.Rls[3 * k + 1] * t[1] + o.Rls[3 * k + 2] * t[2];
        const T rho = hsqrt(P[0] * P[0] + P[2] * P[2]);
        if (o.shape == HOT_SHAPE_TORUS) {
            const T q0 = rho - o.p1[0], L = hsqrt(q0 * q0 + P[1] * P[1]);
            colliding = L - o.p1[1] <= (T)0;
            // gradient of sqrt((sqrt(x^2 + z^2) - r0)^2 + y^2) - r1 in primitive space, then R_ls
            const T gr = q0 / L, G[3] = { gr * P[0] / rho, P[1] / L, gr * P[2] / rho };
#pragma unroll
            for (int k = 0; k < 3; ++k) N[k] = o.Rls[k] * G[0] + o.Rls[3 + k] * G[1] + o.Rls[6 + k] * G[2];
        }
        else { // STICKY only: no normal needed
            const T d0 = rho - o.p1[0], d1 = habs(P[1]) - (T)0.5 * o.p1[1];
            const T m0 = d0 > (T)0 ? d0 : (T)0, m1 = d1 > (T)0 ? d1 : (T)0, mx = d0 > d1 ? d0 : d1;
            colliding = (mx < (T)0 ? mx : (T)0) + hsqrt(m0 * m0 + m1 * m1) <= (T)0;
        }
    }
    else if (o.shape == HOT_SHAPE_ROTATED_BOX) { // AnalyticBox (AnalyticLevelSet.cpp:486-529), STICKY only
        const T t[3] = { X[0] - o.p0[0], X[1] - o.p0[1], X[2] - o.p0[2] };
        T dd = -(T)3.4e38, q2 = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T d = habs(o.Rls[3 * k] * t[0] + o.Rls[3 * k + 1] * t[1] + o.Rls[3 * k + 2] * t[2]) - o.p1[k];
            dd = d > dd ? d : dd;
            const T q = d < (T)0 ? (T)0 : d;
            q2 += q * q;
        }
        colliding = (dd < (T)0 ? dd : (T)0) + hsqrt(q2) <= (T)0;
    }
    else { // axis-aligned box (STICKY only): signedDistancePrimitive of the centred box
        T dd = -(T)3.4e38, q2 = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T c = (o.p0[k] + o.p1[k]) / (T)2, h = (o.p1[k] - o.p0[k]) / (T)2;
            const T d = habs(X[k] - c) - h;
            dd = d > dd ? d : dd;
            const T q = d < (T)0 ? (T)0 : d;
            q2 += q * q;
        }
        const T phi = (dd < (T)0 ? dd : (T)0) + hsqrt(q2);
        colliding = phi <= (T)0;
    }
    if (!colliding) return false;
    // v_object = omega x (x - b) + (ds/dt / s)(x - b) + db/dt   (the level sets here have no material velocity)
    const T ss = o.dsdt * o.inv_s;
    const T vo[3] = { o.omega[1] * xb[2] - o.omega[2] * xb[1] + ss * xb[0] + o.dbdt[0], o.omega[2] * xb[0] - o.omega[0] * xb[2] + ss * xb[1] + o.dbdt[1],
        o.omega[0] * xb[1] - o.omega[1] * xb[0] + ss * xb[2] + o.dbdt[2] };
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] -= vo[k];
    if (o.type == HOT_COLLISION_STICKY)
        v[0] = v[1] = v[2] = (T)0;
    else {
#pragma unroll
        for (int k = 0; k < 3; ++k) n[k] = o.R[k] * N[0] + o.R[3 + k] * N[1] + o.R[6 + k] * N[2]; // world normal R N
        const T dot = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
        if (o.type == HOT_COLLISION_SLIP || dot < (T)0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] -= n[k] * dot;
            if (o.friction != (T)0 && dot < (T)0) { // kinematic friction
                const T vn = hsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                if (-dot * o.friction < vn) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) v[k] += (v[k] / vn) * dot * o.friction;
                }
                else
                    v[0] = v[1] = v[2] = (T)0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] += vo[k];
    return true;
}

// Eigen::Quaternion(w, x, y, z).normalized().toRotationMatrix(), column-major
inline void co_quat_to_matrix(const double (&q)[4], double (&R)[9])
{
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(n > 0)) n = 1;
    const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz), R[3] = txy - twz, R[6] = txz + twy;
    R[1] = txy + twz, R[4] = 1 - (txx + tzz), R[7] = tyz - twx;
    R[2] = txz - twy, R[5] = tyz + twx, R[8] = 1 - (txx + tyy);
}

// AnalyticCollisionObject::evalMaxSpeed (CollisionObject.cpp:200-238): the object's largest speed over the corners of
// the particle box (expanded by the caller) and of the level set's bounds that pass the reference's overlap test
// (kept as written there: "any component above the min" and "any component below the max").
inline void co_bounds(const hot_collision_object& o, double (&lo)[3], double (&hi)[3])
{
    for (int d = 0; d < 3; ++d) {
        if (o.shape == HOT_SHAPE_SPHERE)
            lo[d] = o.p0[d] - o.p1[0], hi[d] = o.p0[d] + o.p1[0];
        else if (o.shape == HOT_SHAPE_TORUS) // bounding sphere r0 + r1 (AnalyticLevelSet.cpp:610-617)
            lo[d] = o.p0[d] - (o.p1[0] + o.p1[1]), hi[d] = o.p0[d] + (o.p1[0] + o.p1[1]);
        else if (o.shape == HOT_SHAPE_ROTATED_BOX) { // bounding sphere |half edges| (AnalyticLevelSet.cpp:542-548)
            const double rr = std::sqrt(o.p1[0] * o.p1[0] + o.p1[1] * o.p1[1] + o.p1[2] * o.p1[2]);
            lo[d] = o.p0[d] - rr, hi[d] = o.p0[d] + rr;
        }
        else if (o.shape == HOT_SHAPE_CAPPED_CYLINDER) { // AnalyticLevelSet.h:289-295
            const double rr = std::sqrt(o.p1[0] * o.p1[0] + 0.25 * o.p1[1] * o.p1[1]);
            lo[d] = o.p0[d] - rr, hi[d] = o.p0[d] + rr;
        }
        else
            lo[d] = o.p0[d], hi[d] = o.p1[d];
    }
}
inline double co_max_speed(const hot_collision_object* po, const double (&pmin)[3], const double (&pmax)[3])
{
    const hot_collision_object& o = *po;
    const double wn = std::sqrt(o.omega[0] * o.omega[0] + o.omega[1] * o.omega[1] + o.omega[2] * o.omega[2]);
    if (o.dsdt == 0 && wn == 0) return std::sqrt(o.dbdt[0] * o.dbdt[0] + o.dbdt[1] * o.dbdt[1] + o.dbdt[2] * o.dbdt[2]);
    double lo[3], hi[3]; // ls->getBounds: Sphere (AnalyticLevelSet.cpp:465-469), AxisAlignedAnalyticBox (:371-374)
    if (o.shape == HOT_SHAPE_UNION) { // DisjointUnionLevelSet::getBounds (:157-167): the box around the members' bounds
        for (int d = 0; d < 3; ++d) lo[d] = 1.7e308, hi[d] = -1.7e308;
        for (int m = 1; m <= (int)o.p1[0]; ++m) {
            double l[3], h[3];
            co_bounds(po[m], l, h);
            for (int d = 0; d < 3; ++d) lo[d] = std::min(lo[d], l[d]), hi[d] = std::max(hi[d], h[d]);
        }
    }
    else if (o.shape == HOT_SHAPE_DIFFERENCE) // DifferenceLevelSet::getBounds (:239-242): those of A
        co_bounds(po[1], lo, hi);
    else
        co_bounds(o, lo, hi);
    const double one_over_s = 1 / o.s;
    double best = 0;
    auto speed_at = [&](const double (&x)[3]) {
        const double xb[3] = { x[0] - o.b[0], x[1] - o.b[1], x[2] - o.b[2] }, ss = o.dsdt * one_over_s;
        const double v[3] = { o.omega[1] * xb[2] - o.omega[2] * xb[1] + ss * xb[0] + o.dbdt[0], o.omega[2] * xb[0] - o.omega[0] * xb[2] + ss * xb[1] + o.dbdt[1],
            o.omega[0] * xb[1] - o.omega[1] * xb[0] + ss * xb[2] + o.dbdt[2] };
        best = std::max(best, std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
    };
    for (int i = 0; i < 8; ++i) {
        double x[3], X[3];
        for (int d = 0; d < 3; ++d) x[d] = (i & (1 << d)) ? pmin[d] : pmax[d];
        for (int k = 0; k < 3; ++k) X[k] = (o.R[3 * k] * (x[0] - o.b[0]) + o.R[3 * k + 1] * (x[1] - o.b[1]) + o.R[3 * k + 2] * (x[2] - o.b[2])) * one_over_s;
        const bool above = lo[0] < X[0] || lo[1] < X[1] || lo[2] < X[2], below = X[0] < hi[0] || X[1] < hi[1] || X[2] < hi[2];
        if (above && below) speed_at(x);
    }
    for (int i = 0; i < 8; ++i) {
        double X[3], x[3];
        for (int d = 0; d < 3; ++d) X[d] = (i & (1 << d)) ? lo[d] : hi[d];
        for (int k = 0; k < 3; ++k) x[k] = (o.R[k] * X[0] + o.R[3 + k] * X[1] + o.R[6 + k] * X[2]) * o.s + o.b[k];
        const bool above = pmin[0] < x[0] || pmin[1] < x[1] || pmin[2] < x[2], below = x[0] < pmax[0] || x[1] < pmax[1] || x[2] < pmax[2];
        if (above && below) speed_at(x);
    }
    return best;
}

// multiObjectCollision with wn (CollisionObject.cpp:107-148); nb = normal_basis (column-major 3x3)
template <class T>
__device__ __forceinline__ bool co_multi(const CollObj<T>* __restrict__ objs, int nobj, const T (&x)[3], T (&v)[3], T (&nb)[9], T (&wn)[3])
{
    bool any = false;
    int slip_count = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) nb[k] = (T)0;
    wn[0] = wn[1] = wn[2] = (T)0;
    for (int k = 0; k < nobj; k += 1 + objs[k].nmember) { // a composite's members ride along behind it
        T n[3] = { 0, 0, 0 };
        const bool collide = co_detect_resolve(objs + k, x, v, n);
        any = any || collide;
        if (!collide) continue;
        if (objs[k].type == HOT_COLLISION_STICKY) {
            wn[0] = wn[1] = wn[2] = (T)0;
#pragma unroll
            for (int q = 0; q < 9; ++q) nb[q] = (q % 4 == 0) ? (T)1 : (T)0;
            break;
        }
        for (int c = 0; c < slip_count; ++c) { // Gram-Schmidt against the normals already taken
            const T dot = nb[3 * c] * n[0] + nb[3 * c + 1] * n[1] + nb[3 * c + 2] * n[2];
            n[0] -= dot * nb[3 * c], n[1] -= dot * nb[3 * c + 1], n[2] -= dot * nb[3 * c + 2];
        }
        wn[0] = n[0], wn[1] = n[1], wn[2] = n[2];
        const T len = hsqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (len) {
            nb[3 * slip_count] = n[0] / len, nb[3 * slip_count + 1] = n[1] / len, nb[3 * slip_count + 2] = n[2] / len;
            if (++slip_count == 3) break;
        }
    }
    return any;
}

// rotation taking a to (1,0,0): Eigen::Quaternion::setFromTwoVectors(a, e_x).toRotationMatrix(), column-major out
template <class T>
__device__ __forceinline__ void co_rotate_to_x(const T (&a)[3], T (&R)[9])
{
    const T la = hsqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const T v0[3] = { a[0] / la, a[1] / la, a[2] / la };
    T c = v0[0]; // v1 = (1,0,0)
    T qx, qy, qz, qw;
    const T eps = sizeof(T) == 8 ? (T)1e-12 : (T)1e-5; // NumTraits::dummy_precision
    if (c < (T)-1 + eps) {
        // nearly opposite: any axis orthogonal to both (Eigen takes a singular vector; the choice only turns the
        // tangent plane, which the constrained solve does not see)
        c = c > (T)-1 ? c : (T)-1;
        T ax[3] = { 0, v0[2], -v0[1] }; // v0 x e_x
        T l = hsqrt(ax[1] * ax[1] + ax[2] * ax[2]);
        if (l < (T)1e-30) ax[1] = 1, ax[2] = 0, l = 1;
        const T w2 = ((T)1 + c) * (T)0.5, s = hsqrt((T)1 - w2);
        qw = hsqrt(w2), qx = 0, qy = ax[1] / l * s, qz = ax[2] / l * s;
    }
    else {
        // axis = v0 x v1
        const T ax[3] = { 0, v0[2], -v0[1] };
        const T s = hsqrt(((T)1 + c) * (T)2), invs = (T)1 / s;
        qx = ax[0] * invs, qy = ax[1] * invs, qz = ax[2] * invs, qw = s * (T)0.5;
    }
    const T tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const T twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz), R[3] = txy - twz, R[6] = txz + twy;
    R[1] = txy + twz, R[4] = 1 - (txx + tzz), R[7] = tyz - twx;
    R[2] = txz - twy, R[5] = tyz + twx, R[8] = 1 - (txx + tyy);
}

} // namespace hot
