// libhotmi355x — per-particle constitutive math (device).
//
//   corotated_state()    CorotatedIsotropic<T,3>::updateScratch + psi + firstPiola
//                        (reference Lib/Ziran/Physics/ConstitutiveModel/CorotatedIsotropic.h:110-144,151-160)
//   corotated_hessian()  the A / B blocks of dP/dF in the SVD frame with optional PSD projection
//                        (SvdBasedIsotropicHelper.h:223-247) — the 9x9 dP/dF of CorotatedIsotropic.h:174-230 is
//                        never materialised as a 25-term sum per entry; callers contract U K V^T directly.
//   von_mises_project / snow_project   PlasticityApplier.cpp:96-131 / :18-50
#pragma once
#include "hot_svd.h"

namespace hot {

template <class T>
__device__ __forceinline__ T clamp_small_magnitude(T x, T eps)
{
    if (x < -eps) return x;
    if (x < (T)0) return -eps;
    if (x < eps) return eps;
    return x;
}

// psi WITHOUT an SVD, for the line search's energy-only trials (round 5; at C4 the trials' SVDs were 30 % of a step).
//   psi = mu |F - R|_F^2 + lambda/2 (J - 1)^2 (CorotatedIsotropic.h:151-155) needs of the polar decomposition only s = tr S = sigma_0 + sigma_1 + sigma_2:
//   |F - R|^2 = |F|^2 - 2 s + 3.  With I1, I2, J the invariants of C = F^T F: a = sigma_0 sigma_1 + ... = (s^2 - I1) / 2 and a^2 = I2 + 2 J s, so s is the
//   largest root of  s^4 - 2 I1 s^2 - 8 J s + I1^2 - 4 I2 = 0  (the other three are s with two signs flipped).
// Near a rotation all of this cancels, so the unknown is u = |F - R|^2 / 2 = e1 - (s - 3) itself and everything is written in the invariants
// e1, e2, e3 of the Green strain E = (C - I) / 2 (I1 = 3 + 2 e1, I2 = 3 + 4 e1 + 4 e2, J^2 = 1 + q, q = 2 e1 + 4 e2 + 8 e3, j = J - 1 = q / (sqrt(1 + q) + 1),
// e1 - j = j^2 / 2 - 2 e2 - 4 e3):  g(u) = u^4 + g3 u^3 + g2 u^2 + g1 u + g0 = 0 with g0 = O(strain^2) free of first-order terms.  u = 0 is the upper bound
// sigma - 1 <= (sigma^2 - 1) / 2 of every term, i.e. a point beyond the quartic's largest root in s where it is convex: Newton from there is monotone, 2 - 4
// steps at the strains of a time step, ~10 at 100 %.  Measured against 50-digit arithmetic (tests/test_gpu_force.py::test_trial_energy_without_svd): relative error of u 8e-9 at strain
// 1e-8, 1e-12 at 1e-4, 2e-15 at 0.1 in fp64 — a factor 3 - 6 BELOW the sigma form's (whose sigma_i - 1 cancels the same way), 4e-6 at 100 % in fp32.
// Not for det F <= 0.1 (the sign convention puts an inverted element's negative singular value last, where the largest root may be a double one) nor
// where Newton has not settled in 12 steps: false, and the caller takes the singular values.
template <class T>
__device__ __forceinline__ bool corotated_psi_invariants(const Mat3<T>& F, T mu, T lambda, T& psi)
{
    // E = (F^T F - I) / 2, the -1 inside the fma chain
    T E00 = (T)0.5 * fma(F(0, 0), F(0, 0), fma(F(1, 0), F(1, 0), fma(F(2, 0), F(2, 0), (T)-1)));
    T E11 = (T)0.5 * fma(F(0, 1), F(0, 1), fma(F(1, 1), F(1, 1), fma(F(2, 1), F(2, 1), (T)-1)));
    T E22 = (T)0.5 * fma(F(0, 2), F(0, 2), fma(F(1, 2), F(1, 2), fma(F(2, 2), F(2, 2), (T)-1)));
    T E01 = (T)0.5 * (F(0, 0) * F(0, 1) + F(1, 0) * F(1, 1) + F(2, 0) * F(2, 1));
    T E02 = (T)0.5 * (F(0, 0) * F(0, 2) + F(1, 0) * F(1, 2) + F(2, 0) * F(2, 2));
    T E12 = (T)0.5 * (F(0, 1) * F(0, 2) + F(1, 1) * F(1, 2) + F(2, 1) * F(2, 2));
    const T e1 = E00 + E11 + E22;
    const T e2 = E00 * E11 + E11 * E22 + E00 * E22 - E01 * E01 - E12 * E12 - E02 * E02;
    const T e3 = E00 * (E11 * E22 - E12 * E12) - E01 * (E01 * E22 - E12 * E02) + E02 * (E01 * E12 - E11 * E02);
    const T q = (T)2 * e1 + (T)4 * e2 + (T)8 * e3; // J^2 - 1
    if (!(q > (T)-0.99) || !(m3_det(F) > (T)0)) return false;
    const T j = q / (hsqrt((T)1 + q) + (T)1);
    const T g0 = ((e1 + (T)8) * e1 + (T)28) * e1 * e1 - (T)8 * j * e1 + (T)12 * j * j - (T)64 * e2 - (T)96 * e3;
    const T g1 = -((((T)4 * e1 + (T)28) * e1 + (T)72) * e1 + (T)64 - (T)8 * j);
    const T g2 = ((T)6 * e1 + (T)32) * e1 + (T)48;
    const T g3 = -((T)4 * e1 + (T)12);
    const T tol = sizeof(T) == 8 ? (T)8.9e-16 : (T)4.8e-7; // 4 eps
    T u = (T)0;
    bool settled = false;
    for (int it = 0; it < 12 && !settled; ++it) {
        const T g = (((u + g3) * u + g2) * u + g1) * u + g0;
        const T gp = (((T)4 * u + (T)3 * g3) * u + (T)2 * g2) * u + g1;
        const T du = -g / gp;
        u += du;
        settled = !(du > tol * u);
    }
    u = u > (T)0 ? u : (T)0; // (round-off of g0 at F = a rotation)
    psi = (T)2 * mu * u + (T)0.5 * lambda * j * j;
    return settled;
}

// psi and P = 2 mu (F - R) + lambda (J - 1) J F^-T from one SVD
template <class T>
__device__ inline void corotated_state(const Mat3<T>& F, T mu, T lambda, T& psi, Mat3<T>& P, T* psi_sigma = nullptr /*psi evaluated as corotated_psi_sigma does: what an energy-only trial of the same F would return*/)
{
    // (the trial form first: nothing of it but two registers lives across the SVD)
    T pt = (T)0;
    const bool have_pt = psi_sigma && corotated_psi_invariants(F, mu, lambda, pt);
    Mat3<T> U, V;
    T sg[3];
    svd3(F, U, sg, V);
    Mat3<T> R = m3_mul_bt(U, V);
    Mat3<T> JFinvT = m3_cofactor(F);
    T J = sg[0] * sg[1] * sg[2];
    T fr = (T)0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        T d = F.a[i] - R.a[i];
        fr += d * d;
        P.a[i] = (T)2 * mu * d + lambda * (J - (T)1) * JFinvT.a[i];
    }
    T Jm1 = J - (T)1;
    psi = mu * fr + (T)0.5 * lambda * Jm1 * Jm1;
    if (psi_sigma) {
        if (!have_pt) {
            const T d0 = sg[0] - (T)1, d1 = sg[1] - (T)1, d2 = sg[2] - (T)1;
            pt = mu * (d0 * d0 + d1 * d1 + d2 * d2) + (T)0.5 * lambda * Jm1 * Jm1;
        }
        *psi_sigma = pt;
    }
}

// psi alone: from the invariants (above); where those decline, from the singular values: |F - R|_F^2 = sum (sigma_i - 1)^2.  U and V are never used,
// so their rotations are dead code after inlining; the bidiagonal and with it sigma are bit-identical to corotated_state's — either way the value is
// the one corotated_state returns in *psi_sigma for the same F.
template <class T>
__device__ inline T corotated_psi_sigma(const Mat3<T>& F, T mu, T lambda)
{
    T pt;
    if (corotated_psi_invariants(F, mu, lambda, pt)) return pt;
    Mat3<T> U, V;
    T sg[3];
    svd3(F, U, sg, V);
    T J = sg[0] * sg[1] * sg[2];
    T d0 = sg[0] - (T)1, d1 = sg[1] - (T)1, d2 = sg[2] - (T)1;
    T Jm1 = J - (T)1;
    return mu * (d0 * d0 + d1 * d1 + d2 * d2) + (T)0.5 * lambda * Jm1 * Jm1;
}

// dP/dF in the SVD frame: symmetric 3x3 A (diagonal-diagonal couplings) and three symmetric 2x2 blocks
template <class T>
struct HessBlocks {
    Mat3<T> U, V;
    Mat3<T> A;
    T B01[3], B12[3], B20[3]; // (00, 01, 11)
};
template <class T>
__device__ inline void corotated_hessian(const Mat3<T>& F, T mu, T lambda, bool project, HessBlocks<T>& h)
{
    const T eps = (T)1e-6;
    T s[3];
    svd3(F, h.U, s, h.V);
    T J = s[0] * s[1] * s[2];
    T _2mu = mu * (T)2;
    T _lambda = lambda * (J - (T)1);
    T Sprod[3] = { s[1] * s[2], s[0] * s[2], s[0] * s[1] };
    T psi0 = _2mu * (s[0] - (T)1) + _lambda * Sprod[0];
    T psi1 = _2mu * (s[1] - (T)1) + _lambda * Sprod[1];
    T psi2 = _2mu * (s[2] - (T)1) + _lambda * Sprod[2];
    h.A(0, 0) = _2mu + lambda * Sprod[0] * Sprod[0];
    h.A(1, 1) = _2mu + lambda * Sprod[1] * Sprod[1];
    h.A(2, 2) = _2mu + lambda * Sprod[2] * Sprod[2];
    h.A(0, 1) = h.A(1, 0) = _lambda * s[2] + lambda * Sprod[0] * Sprod[1];
    h.A(0, 2) = h.A(2, 0) = _lambda * s[1] + lambda * Sprod[0] * Sprod[2];
    h.A(1, 2) = h.A(2, 1) = _lambda * s[0] + lambda * Sprod[1] * Sprod[2];
    T m01 = _2mu - _lambda * s[2], m02 = _2mu - _lambda * s[1], m12 = _2mu - _lambda * s[0];
    T p01 = (psi0 + psi1) / clamp_small_magnitude(s[0] + s[1], eps);
    T p02 = (psi0 + psi2) / clamp_small_magnitude(s[0] + s[2], eps);
    T p12 = (psi1 + psi2) / clamp_small_magnitude(s[1] + s[2], eps);
    h.B01[0] = h.B01[2] = (m01 + p01) * (T)0.5, h.B01[1] = (m01 - p01) * (T)0.5;
    h.B12[0] = h.B12[2] = (m12 + p12) * (T)0.5, h.B12[1] = (m12 - p12) * (T)0.5;
    h.B20[0] = h.B20[2] = (m02 + p02) * (T)0.5, h.B20[1] = (m02 - p02) * (T)0.5;
    if (project) {
        make_pd3(h.A);
        make_pd2(h.B01[0], h.B01[1], h.B01[2]);
        make_pd2(h.B12[0], h.B12[1], h.B12[2]);
        make_pd2(h.B20[0], h.B20[1], h.B20[2]);
    }
}
// K = (dPhat/dFhat) : D in the SVD frame (SvdBasedIsotropicHelper.h:257-282)
template <class T>
__device__ __forceinline__ Mat3<T> hess_contract(const HessBlocks<T>& h, const Mat3<T>& D)
{
    Mat3<T> B;
    B(0, 0) = h.A(0, 0) * D(0, 0) + h.A(0, 1) * D(1, 1) + h.A(0, 2) * D(2, 2);
    B(1, 1) = h.A(1, 0) * D(0, 0) + h.A(1, 1) * D(1, 1) + h.A(1, 2) * D(2, 2);
    B(2, 2) = h.A(2, 0) * D(0, 0) + h.A(2, 1) * D(1, 1) + h.A(2, 2) * D(2, 2);
    B(0, 1) = h.B01[0] * D(0, 1) + h.B01[1] * D(1, 0);
    B(1, 0) = h.B01[1] * D(0, 1) + h.B01[2] * D(1, 0);
    B(0, 2) = h.B20[0] * D(0, 2) + h.B20[1] * D(2, 0);
    B(2, 0) = h.B20[1] * D(0, 2) + h.B20[2] * D(2, 0);
    B(1, 2) = h.B12[0] * D(1, 2) + h.B12[1] * D(2, 1);
    B(2, 1) = h.B12[1] * D(1, 2) + h.B12[2] * D(2, 1);
    return B;
}
// dP = U (K : (U^T dF V)) V^T   (CorotatedIsotropic.h:162-171)
template <class T>
__device__ __forceinline__ Mat3<T> hess_apply(const HessBlocks<T>& h, const Mat3<T>& dF)
{
    Mat3<T> D = m3_mul(m3_mul_at(h.U, dF), h.V);
    Mat3<T> K = hess_contract(h, D);
    return m3_mul_bt(m3_mul(h.U, K), h.V);
}

template <class T>
__device__ inline bool von_mises_project(Mat3<T>& strain, T mu, T lambda, T yield_stress)
{
    Mat3<T> U, V;
    T s[3];
    svd3(strain, U, s, V);
#pragma unroll
    for (int d = 0; d < 3; ++d) s[d] = s[d] > (T)1e-4 ? s[d] : (T)1e-4;
    T J = s[0] * s[1] * s[2];
    T tau[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) tau[d] = (T)2 * mu * (s[d] - (T)1) * s[d] + lambda * (J - (T)1) * J;
    T tr = tau[0] + tau[1] + tau[2];
    T st[3] = { tau[0] - tr / (T)3, tau[1] - tr / (T)3, tau[2] - tr / (T)3 };
    T s_norm = hsqrt(st[0] * st[0] + st[1] * st[1] + st[2] * st[2]);
    T scaled_tauy = hsqrt((T)2 / (T)3) * yield_stress;
    if (s_norm - scaled_tauy <= (T)0) return false;
    T alpha = scaled_tauy / s_norm;
    T sn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        T tau_new = alpha * st[d] + tr / (T)3;
        T b2m4ac = mu * mu - (T)2 * mu * (lambda * (J - (T)1) * J - tau_new);
        sn[d] = (mu + hsqrt(b2m4ac)) / ((T)2 * mu);
    }
    Mat3<T> US;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) US(r, c) = U(r, c) * sn[c];
    strain = m3_mul_bt(US, V);
    return true;
}

template <class T>
__device__ __forceinline__ T hexp(T x);
template <>
__device__ __forceinline__ float hexp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double hexp<double>(double x) { return exp(x); }

template <class T>
__device__ inline void snow_project(Mat3<T>& strain, T& mu, T& lambda, T& Jp, T psi, T theta_c, T theta_s, T min_Jp, T max_Jp)
{
    Mat3<T> U, V;
    T s[3];
    svd3(strain, U, s, V);
    T Fe_det = (T)1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T v = s[i] < (T)1 + theta_s ? s[i] : (T)1 + theta_s;
        s[i] = v > (T)1 - theta_c ? v : (T)1 - theta_c;
        Fe_det *= s[i];
    }
    Mat3<T> US;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) US(r, c) = U(r, c) * s[c];
    T Jp_new = Jp * m3_det(strain) / Fe_det;
    if (!(Jp_new <= max_Jp)) Jp_new = max_Jp;
    if (!(Jp_new >= min_Jp)) Jp_new = min_Jp;
    strain = m3_mul_bt(US, V);
    T hard = hexp(psi * (Jp - Jp_new));
    mu *= hard;
    lambda *= hard;
    Jp = Jp_new;
}

} // namespace hot
