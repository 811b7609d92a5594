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

// psi and P = 2 mu (F - R) + lambda (J - 1) J F^-T from one SVD
template <class T>
__device__ inline void corotated_state(const Mat3<T>& F, T mu, T lambda, T& psi, Mat3<T>& P, T* psi_sigma = nullptr /*psi evaluated as corotated_psi_sigma does, from the same singular values*/)
{
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
        const T d0 = sg[0] - (T)1, d1 = sg[1] - (T)1, d2 = sg[2] - (T)1;
        *psi_sigma = mu * (d0 * d0 + d1 * d1 + d2 * d2) + (T)0.5 * lambda * Jm1 * Jm1;
    }
}

// psi alone, from the singular values: |F - R|_F^2 = sum (sigma_i - 1)^2.  U and V are never used, so their rotations are dead code after
// inlining; the bidiagonal and with it sigma are bit-identical to corotated_state's.
template <class T>
__device__ inline T corotated_psi_sigma(const Mat3<T>& F, T mu, T lambda)
{
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
