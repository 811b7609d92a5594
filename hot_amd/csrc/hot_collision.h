// libhotmi355x — analytic collision objects evaluated per grid node on the device.
//
// Restates, for objects whose rotation / scaling are the identity (translation b with velocity dbdt allowed):
//   AnalyticCollisionObject::detectAndResolveCollision   Lib/Ziran/Math/Geometry/CollisionObject.cpp:384-447
//   AnalyticCollisionObject::multiObjectCollision (wn)    :107-148
//   HalfSpace / Sphere / AxisAlignedAnalyticBox queries   Lib/Ziran/Math/Geometry/AnalyticLevelSet.cpp:111-118,264-288,353-363,435-452,504-529
//   RotationExtractor<T,3>::rotate (Eigen::Quaternion::setFromTwoVectors + toRotationMatrix)   Lib/MPM/MpmSimulationBase.h:271-281
// The box's normal comes from automatic differentiation of a distance that is not differentiable inside the box in the
// reference; only STICKY boxes (no normal needed) are accepted.
#pragma once
#include "hot_svd.h"
#include "../../include/hot_mi355x.h"

namespace hot {

template <class T>
struct CollObj { // device copy of hot_collision_object in the simulation's scalar type
    int32_t shape, type;
    T p0[3], p1[3], friction, b[3], dbdt[3];
};

// returns whether node position x collides with o; v is replaced by the resolved velocity, n by the world normal (SLIP / SEPARATE)
template <class T>
__device__ __forceinline__ bool co_detect_resolve(const CollObj<T>& o, const T (&x)[3], T (&v)[3], T (&n)[3])
{
    T X[3] = { x[0] - o.b[0], x[1] - o.b[1], x[2] - o.b[2] };
    T N[3] = { 0, 0, 0 };
    bool colliding = false;
    if (o.shape == HOT_SHAPE_HALFSPACE) {
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
    // v_object = omega x (x - b) + (ds/dt / s)(x - b) + R s V_material + db/dt  with omega = 0, ds/dt = 0, V_material = 0
    const T vo[3] = { o.dbdt[0], o.dbdt[1], o.dbdt[2] };
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] -= vo[k];
    if (o.type == HOT_COLLISION_STICKY)
        v[0] = v[1] = v[2] = (T)0;
    else {
        n[0] = N[0], n[1] = N[1], n[2] = N[2];
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

// multiObjectCollision with wn (CollisionObject.cpp:107-148); nb = normal_basis (column-major 3x3)
template <class T>
__device__ __forceinline__ bool co_multi(const CollObj<T>* __restrict__ objs, int nobj, const T (&x)[3], T (&v)[3], T (&nb)[9], T (&wn)[3])
{
    bool any = false;
    int slip_count = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) nb[k] = (T)0;
    wn[0] = wn[1] = wn[2] = (T)0;
    for (int k = 0; k < nobj; ++k) {
        T n[3] = { 0, 0, 0 };
        const bool collide = co_detect_resolve(objs[k], x, v, n);
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
