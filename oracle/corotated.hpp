// ORACLE (test infrastructure only — never linked into the product library).
//
// Fixed-corotated ("CorotatedIsotropic") hyperelasticity, 3-D, restated from
//   reference Lib/Ziran/Physics/ConstitutiveModel/CorotatedIsotropic.h:69-73   (Lame parameters)
//   reference ...CorotatedIsotropic.h:75-80,110-144                          (updateScratch, 3-D)
//   reference ...CorotatedIsotropic.h:151-160                                (psi, firstPiola)
//   reference ...CorotatedIsotropic.h:162-171                                (firstPiolaDifferential)
//   reference ...CorotatedIsotropic.h:174-230                                (firstPiolaDerivative, 9x9)
//   reference Lib/Ziran/Physics/ConstitutiveModel/SvdBasedIsotropicHelper.h:223-247 (buildMatrixBlock, projectABBlock)
//   reference ...SvdBasedIsotropicHelper.h:257-282                           (dPdFOfSigmaContract[Projected])
//   reference Lib/Ziran/Math/MathTools.h:162-174                             (clamp_small_magnitude)
// and the two plasticity return mappings
//   reference Lib/Ziran/Physics/PlasticityApplier.cpp:18-50   (SnowPlasticity::projectStrain)
//   reference Lib/Ziran/Physics/PlasticityApplier.cpp:96-131  (VonMisesFixedCorotated::projectStrain)
#pragma once
#include "mat3.hpp"

namespace hot_oracle {

template <class T>
inline T clamp_small_magnitude(T x, T eps)
{
    if (x < -eps) return x;
    if (x < 0) return -eps;
    if (x < eps) return eps;
    return x;
}

template <class T>
inline void lame(T E, T nu, T& mu, T& lambda)
{
    lambda = E * nu / (((T)1 + nu) * ((T)1 - (T)2 * nu));
    mu = E / ((T)2 * ((T)1 + nu));
}

template <class T>
struct CorotatedScratch {
    M3<T> F, U, V, R, JFinvT;
    V3<T> sigma;
    T J;
    T psi0, psi1, psi2, psi00, psi11, psi22, psi01, psi02, psi12;
    T m01, p01, m02, p02, m12, p12;
    M3<T> Aij;
    T B01[3], B12[3], B20[3]; // symmetric 2x2 stored as (00, 01, 11)
};

template <class T>
inline void corotated_update_scratch(const M3<T>& newF, T mu, T lambda, bool project, CorotatedScratch<T>& s)
{
    const T eps = (T)1e-6;
    s.F = newF;
    svd3(s.F, s.U, s.sigma, s.V);
    s.R = s.U * s.V.transpose();
    s.JFinvT = cofactor(s.F);
    s.J = s.sigma(0) * s.sigma(1) * s.sigma(2);
    T _2mu = mu * 2;
    T _lambda = lambda * (s.J - 1);
    T Sprod[3] = { s.sigma(1) * s.sigma(2), s.sigma(0) * s.sigma(2), s.sigma(0) * s.sigma(1) };
    s.psi0 = _2mu * (s.sigma(0) - 1) + _lambda * Sprod[0];
    s.psi1 = _2mu * (s.sigma(1) - 1) + _lambda * Sprod[1];
    s.psi2 = _2mu * (s.sigma(2) - 1) + _lambda * Sprod[2];
    s.psi00 = _2mu + lambda * Sprod[0] * Sprod[0];
    s.psi11 = _2mu + lambda * Sprod[1] * Sprod[1];
    s.psi22 = _2mu + lambda * Sprod[2] * Sprod[2];
    s.psi01 = _lambda * s.sigma(2) + lambda * Sprod[0] * Sprod[1];
    s.psi02 = _lambda * s.sigma(1) + lambda * Sprod[0] * Sprod[2];
    s.psi12 = _lambda * s.sigma(0) + lambda * Sprod[1] * Sprod[2];
    s.m01 = _2mu - _lambda * s.sigma(2);
    s.m02 = _2mu - _lambda * s.sigma(1);
    s.m12 = _2mu - _lambda * s.sigma(0);
    s.p01 = (s.psi0 + s.psi1) / clamp_small_magnitude(s.sigma(0) + s.sigma(1), eps);
    s.p02 = (s.psi0 + s.psi2) / clamp_small_magnitude(s.sigma(0) + s.sigma(2), eps);
    s.p12 = (s.psi1 + s.psi2) / clamp_small_magnitude(s.sigma(1) + s.sigma(2), eps);
    // buildMatrixBlock (always built here so that the un-projected derivative can share the code path)
    s.Aij(0, 0) = s.psi00, s.Aij(1, 1) = s.psi11, s.Aij(2, 2) = s.psi22;
    s.Aij(0, 1) = s.Aij(1, 0) = s.psi01;
    s.Aij(0, 2) = s.Aij(2, 0) = s.psi02;
    s.Aij(1, 2) = s.Aij(2, 1) = s.psi12;
    s.B01[0] = s.B01[2] = (s.m01 + s.p01) * (T)0.5, s.B01[1] = (s.m01 - s.p01) * (T)0.5;
    s.B12[0] = s.B12[2] = (s.m12 + s.p12) * (T)0.5, s.B12[1] = (s.m12 - s.p12) * (T)0.5;
    s.B20[0] = s.B20[2] = (s.m02 + s.p02) * (T)0.5, s.B20[1] = (s.m02 - s.p02) * (T)0.5;
    if (project) {
        make_pd3(s.Aij);
        make_pd2(s.B01[0], s.B01[1], s.B01[2]);
        make_pd2(s.B12[0], s.B12[1], s.B12[2]);
        make_pd2(s.B20[0], s.B20[1], s.B20[2]);
    }
}

template <class T>
inline T corotated_psi(const CorotatedScratch<T>& s, T mu, T lambda)
{
    T Jm1 = s.J - 1;
    return mu * (s.F - s.R).squaredNorm() + (T).5 * lambda * Jm1 * Jm1;
}

// NOT a restatement of the reference: the PRODUCT's form of the same psi for the line search's trial energies (hot_amd/csrc/hot_constitutive.h
// corotated_psi_invariants / corotated_psi_sigma — of the polar decomposition psi needs tr S alone, the largest root of a quartic in the invariants of
// F^T F, and where that declines |F - R|^2 = sum (sigma_i - 1)^2), restated here so that the two forms can be compared where it matters: on the
// accept / reject decisions `Ek <= Ek0` of lineSearch (ImplicitSolver.h:312-333).  Selected for every energy evaluation of a context by
// HOT_ORACLE_PSI_INVARIANTS=1 at hoto_create (tests/oracle_lib.py psi_invariants()); the default is corotated_psi above.
inline bool& psi_invariants_flag()
{
    static bool f = false;
    return f;
}
template <class T>
inline T corotated_psi_product_form(const CorotatedScratch<T>& s, T mu, T lambda)
{
    const M3<T>& F = s.F;
    const T E00 = (T)0.5 * std::fma(F(0, 0), F(0, 0), std::fma(F(1, 0), F(1, 0), std::fma(F(2, 0), F(2, 0), (T)-1)));
    const T E11 = (T)0.5 * std::fma(F(0, 1), F(0, 1), std::fma(F(1, 1), F(1, 1), std::fma(F(2, 1), F(2, 1), (T)-1)));
    const T E22 = (T)0.5 * std::fma(F(0, 2), F(0, 2), std::fma(F(1, 2), F(1, 2), std::fma(F(2, 2), F(2, 2), (T)-1)));
    const T E01 = (T)0.5 * (F(0, 0) * F(0, 1) + F(1, 0) * F(1, 1) + F(2, 0) * F(2, 1));
    const T E02 = (T)0.5 * (F(0, 0) * F(0, 2) + F(1, 0) * F(1, 2) + F(2, 0) * F(2, 2));
    const T E12 = (T)0.5 * (F(0, 1) * F(0, 2) + F(1, 1) * F(1, 2) + F(2, 1) * F(2, 2));
    const T e1 = E00 + E11 + E22;
    const T e2 = E00 * E11 + E11 * E22 + E00 * E22 - E01 * E01 - E12 * E12 - E02 * E02;
    const T e3 = E00 * (E11 * E22 - E12 * E12) - E01 * (E01 * E22 - E12 * E02) + E02 * (E01 * E12 - E11 * E02);
    const T q = (T)2 * e1 + (T)4 * e2 + (T)8 * e3; // J^2 - 1
    bool settled = false;
    T psi = 0;
    if (q > (T)-0.99 && F.determinant() > (T)0) {
        const T j = q / (std::sqrt((T)1 + q) + (T)1);
        const T g0 = ((e1 + (T)8) * e1 + (T)28) * e1 * e1 - (T)8 * j * e1 + (T)12 * j * j - (T)64 * e2 - (T)96 * e3;
        const T g1 = -((((T)4 * e1 + (T)28) * e1 + (T)72) * e1 + (T)64 - (T)8 * j);
        const T g2 = ((T)6 * e1 + (T)32) * e1 + (T)48;
        const T g3 = -((T)4 * e1 + (T)12);
        const T tol = sizeof(T) == 8 ? (T)8.9e-16 : (T)4.8e-7;
        T u = 0;
        for (int it = 0; it < 12 && !settled; ++it) {
            const T g = (((u + g3) * u + g2) * u + g1) * u + g0;
            const T gp = (((T)4 * u + (T)3 * g3) * u + (T)2 * g2) * u + g1;
            const T du = -g / gp;
            u += du;
            settled = !(du > tol * u);
        }
        u = u > (T)0 ? u : (T)0;
        psi = (T)2 * mu * u + (T)0.5 * lambda * j * j;
    }
    if (settled) return psi;
    const T d0 = s.sigma(0) - 1, d1 = s.sigma(1) - 1, d2 = s.sigma(2) - 1, Jm1 = s.J - 1;
    return mu * (d0 * d0 + d1 * d1 + d2 * d2) + (T)0.5 * lambda * Jm1 * Jm1;
}

template <class T>
inline M3<T> corotated_first_piola(const CorotatedScratch<T>& s, T mu, T lambda)
{
    return (s.F - s.R) * ((T)2 * mu) + s.JFinvT * (lambda * (s.J - 1));
}

// K = dPhat/dFhat : D in the SVD frame (works for both projected and un-projected blocks since the
// blocks were (re)built in the scratch): reference SvdBasedIsotropicHelper.h:257-282
template <class T>
inline M3<T> corotated_contract(const CorotatedScratch<T>& s, const M3<T>& A)
{
    M3<T> B;
    B(0, 0) = s.Aij(0, 0) * A(0, 0) + s.Aij(0, 1) * A(1, 1) + s.Aij(0, 2) * A(2, 2);
    B(1, 1) = s.Aij(1, 0) * A(0, 0) + s.Aij(1, 1) * A(1, 1) + s.Aij(1, 2) * A(2, 2);
    B(2, 2) = s.Aij(2, 0) * A(0, 0) + s.Aij(2, 1) * A(1, 1) + s.Aij(2, 2) * A(2, 2);
    B(0, 1) = s.B01[0] * A(0, 1) + s.B01[1] * A(1, 0);
    B(1, 0) = s.B01[1] * A(0, 1) + s.B01[2] * A(1, 0);
    B(0, 2) = s.B20[0] * A(0, 2) + s.B20[1] * A(2, 0);
    B(2, 0) = s.B20[1] * A(0, 2) + s.B20[2] * A(2, 0);
    B(1, 2) = s.B12[0] * A(1, 2) + s.B12[1] * A(2, 1);
    B(2, 1) = s.B12[1] * A(1, 2) + s.B12[2] * A(2, 1);
    return B;
}

template <class T>
inline M3<T> corotated_first_piola_differential(const CorotatedScratch<T>& s, const M3<T>& dF)
{
    M3<T> D = s.U.transpose() * dF * s.V;
    M3<T> K = corotated_contract(s, D);
    return s.U * K * s.V.transpose();
}

// 9x9 dP/dF, row/col index = i + 3*j for entry (i,j) (column-major vec), dPdF[ij + 9*rs] (symmetric).
// reference CorotatedIsotropic.h:174-230 (the 25-term sum per entry).
template <class T>
inline void corotated_first_piola_derivative(const CorotatedScratch<T>& ss, T* dPdF /*81, col-major*/)
{
    const M3<T>& U = ss.U;
    const M3<T>& V = ss.V;
    const M3<T>& A = ss.Aij;
    auto B01 = [&](int a, int b) { return ss.B01[a + b]; };
    auto B12 = [&](int a, int b) { return ss.B12[a + b]; };
    auto B20 = [&](int a, int b) { return ss.B20[a + b]; };
    for (int ij = 0; ij < 9; ++ij) {
        int j = ij / 3, i = ij - j * 3;
        for (int rs = 0; rs <= ij; ++rs) {
            int s = rs / 3, r = rs - s * 3;
            T v = A(0, 0) * U(i, 0) * V(j, 0) * U(r, 0) * V(s, 0) + A(0, 1) * U(i, 0) * V(j, 0) * U(r, 1) * V(s, 1) + A(0, 2) * U(i, 0) * V(j, 0) * U(r, 2) * V(s, 2)
                + A(0, 1) * U(i, 1) * V(j, 1) * U(r, 0) * V(s, 0) + A(1, 1) * U(i, 1) * V(j, 1) * U(r, 1) * V(s, 1) + A(1, 2) * U(i, 1) * V(j, 1) * U(r, 2) * V(s, 2)
                + A(0, 2) * U(i, 2) * V(j, 2) * U(r, 0) * V(s, 0) + A(1, 2) * U(i, 2) * V(j, 2) * U(r, 1) * V(s, 1) + A(2, 2) * U(i, 2) * V(j, 2) * U(r, 2) * V(s, 2)
                + B01(0, 0) * U(i, 0) * V(j, 1) * U(r, 0) * V(s, 1) + B01(0, 1) * U(i, 0) * V(j, 1) * U(r, 1) * V(s, 0) + B01(1, 0) * U(i, 1) * V(j, 0) * U(r, 0) * V(s, 1) + B01(1, 1) * U(i, 1) * V(j, 0) * U(r, 1) * V(s, 0)
                + B12(0, 0) * U(i, 1) * V(j, 2) * U(r, 1) * V(s, 2) + B12(0, 1) * U(i, 1) * V(j, 2) * U(r, 2) * V(s, 1) + B12(1, 0) * U(i, 2) * V(j, 1) * U(r, 1) * V(s, 2) + B12(1, 1) * U(i, 2) * V(j, 1) * U(r, 2) * V(s, 1)
                + B20(1, 1) * U(i, 0) * V(j, 2) * U(r, 0) * V(s, 2) + B20(1, 0) * U(i, 0) * V(j, 2) * U(r, 2) * V(s, 0) + B20(0, 1) * U(i, 2) * V(j, 0) * U(r, 0) * V(s, 2) + B20(0, 0) * U(i, 2) * V(j, 0) * U(r, 2) * V(s, 0);
            dPdF[ij + 9 * rs] = dPdF[rs + 9 * ij] = v;
        }
    }
}

// reference PlasticityApplier.cpp:96-131.  Returns true when the strain was projected.
template <class T>
inline bool von_mises_project(M3<T>& strain, T mu, T lambda, T yield_stress)
{
    M3<T> U, V;
    V3<T> sigma;
    svd3(strain, U, sigma, V);
    for (int d = 0; d < 3; ++d) sigma(d) = std::max((T)1e-4, sigma(d));
    T J = sigma(0) * sigma(1) * sigma(2);
    V3<T> tau_trial;
    for (int d = 0; d < 3; ++d) tau_trial(d) = 2 * mu * (sigma(d) - 1) * sigma(d) + lambda * (J - 1) * J;
    T trace_tau = tau_trial(0) + tau_trial(1) + tau_trial(2);
    V3<T> s_trial;
    for (int d = 0; d < 3; ++d) s_trial(d) = tau_trial(d) - trace_tau / (T)3;
    T s_norm = std::sqrt(s_trial.squaredNorm());
    T scaled_tauy = std::sqrt((T)2 / ((T)6 - 3)) * yield_stress;
    if (s_norm - scaled_tauy <= 0) return false;
    T alpha = scaled_tauy / s_norm;
    V3<T> sigma_new;
    for (int d = 0; d < 3; ++d) {
        T tau_new = alpha * s_trial(d) + trace_tau / (T)3;
        T b2m4ac = mu * mu - 2 * mu * (lambda * (J - 1) * J - tau_new);
        T sqrtb2m4ac = std::sqrt(b2m4ac);
        sigma_new(d) = (mu + sqrtb2m4ac) / (2 * mu);
    }
    M3<T> S = M3<T>::zero();
    S(0, 0) = sigma_new(0), S(1, 1) = sigma_new(1), S(2, 2) = sigma_new(2);
    strain = U * S * V.transpose();
    return true;
}

// reference PlasticityApplier.cpp:18-50.  Hardens (mu, lambda) in place and updates Jp.
template <class T>
inline void snow_project(M3<T>& strain, T& mu, T& lambda, T& Jp, T psi, T theta_c, T theta_s, T min_Jp, T max_Jp)
{
    M3<T> U, V;
    V3<T> sigma;
    svd3(strain, U, sigma, V);
    T Fe_det = 1;
    for (int i = 0; i < 3; ++i) {
        sigma(i) = std::max(std::min(sigma(i), (T)1 + theta_s), (T)1 - theta_c);
        Fe_det *= sigma(i);
    }
    M3<T> S = M3<T>::zero();
    S(0, 0) = sigma(0), S(1, 1) = sigma(1), S(2, 2) = sigma(2);
    M3<T> Fe = U * S * V.transpose();
    T Jp_new = Jp * strain.determinant() / Fe_det;
    if (!(Jp_new <= max_Jp)) Jp_new = max_Jp;
    if (!(Jp_new >= min_Jp)) Jp_new = min_Jp;
    strain = Fe;
    mu *= std::exp(psi * (Jp - Jp_new));
    lambda *= std::exp(psi * (Jp - Jp_new));
    Jp = Jp_new;
}

} // namespace hot_oracle
