// libhotmi355x — the constitutive model and the plastic return mappings evaluated for caller-supplied deformation gradients.
//
//   hot_constitutive_eval   CorotatedIsotropic<T,3>::updateScratch + psi + firstPiola + firstPiolaDerivative
//                           (reference Lib/Ziran/Physics/ConstitutiveModel/CorotatedIsotropic.h:110-230): the same device
//                           functions the particle kernels call (hot_constitutive.h, hot_svd.h), one sample per thread.
//   hot_plasticity_eval     VonMisesFixedCorotated / SnowPlasticity::projectStrain (Lib/Ziran/Physics/PlasticityApplier.cpp:96-131, :18-50).
// The reference exposes these as public members of the model / applier classes; here they also give the tests a way to pin
// the device SVD and constitutive code against numpy fixtures directly (tests/golden/fp_golden.npz).
#include "hot_impl.h"
#include "hot_constitutive.h"

namespace hot {

__host__ __device__ constexpr int sym45_(int a, int b) { return a <= b ? (a * 9 - (a * (a - 1)) / 2 + (b - a)) : (b * 9 - (b * (b - 1)) / 2 + (a - b)); }

// one sample per thread, array-of-structs column-major in and out
template <class T>
__global__ __launch_bounds__(256) void k_constitutive_eval(const T* __restrict__ F, const T* __restrict__ Mu, const T* __restrict__ Lam, int n, int project, T* __restrict__ psi,
    T* __restrict__ P, T* __restrict__ dPdF)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    Mat3<T> Fc;
#pragma unroll
    for (int c = 0; c < 9; ++c) Fc.a[c] = F[9 * (int64_t)p + c];
    const T mu = Mu[p], la = Lam[p];
    if (project == 2) { // the line search's energy-only evaluation (k_state<T, true>): psi alone, no SVD where the invariants serve
        if (psi) psi[p] = corotated_psi_sigma(Fc, mu, la);
        return;
    }
    if (psi || P) {
        T e;
        Mat3<T> Pm;
        corotated_state(Fc, mu, la, e, Pm);
        if (psi) psi[p] = e;
        if (P)
#pragma unroll
            for (int c = 0; c < 9; ++c) P[9 * (int64_t)p + c] = Pm.a[c];
    }
    if (dPdF) {
        HessBlocks<T> h;
        corotated_hessian(Fc, mu, la, project != 0, h);
        // column rs of dP/dF = vec(U (K : (U^T E_rs V)) V^T) for the unit matrix E_rs (CorotatedIsotropic.h:162-171)
        for (int rs = 0; rs < 9; ++rs) {
            Mat3<T> E;
#pragma unroll
            for (int c = 0; c < 9; ++c) E.a[c] = c == rs ? (T)1 : (T)0;
            const Mat3<T> dP = hess_apply(h, E);
#pragma unroll
            for (int c = 0; c < 9; ++c) dPdF[81 * (int64_t)p + 9 * rs + c] = dP.a[c];
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_plasticity_eval(int kind, int n, T* __restrict__ F, T* __restrict__ Mu, T* __restrict__ Lam, T* __restrict__ Jp, T yield_stress, T s0, T s1, T s2,
    T s3, T s4)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    Mat3<T> Fc;
#pragma unroll
    for (int c = 0; c < 9; ++c) Fc.a[c] = F[9 * (int64_t)p + c];
    if (kind == 1)
        von_mises_project(Fc, Mu[p], Lam[p], yield_stress);
    else {
        T mu = Mu[p], la = Lam[p], jp = Jp[p];
        snow_project(Fc, mu, la, jp, s0, s1, s2, s3, s4);
        Mu[p] = mu, Lam[p] = la, Jp[p] = jp;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) F[9 * (int64_t)p + c] = Fc.a[c];
}

template <class T>
void Ctx<T>::constitutive_eval(int32_t n, const void* F, const void* mu, const void* lambda, int32_t project, void* psi, void* P, void* dPdF)
{
    need(n > 0 && F && mu && lambda, "hot_constitutive_eval: n > 0, F, mu and lambda are required");
    if (project != 0 && project != 2) project = 1; // any other non-zero value: the PSD projection, as before the psi-only mode (2) existed
    if (project == 2) P = nullptr, dPdF = nullptr;
    DBuf<T> dF, dMu, dLam, dPsi, dP, dD;
    dF.reserve(9 * (size_t)n), dMu.reserve(n), dLam.reserve(n), dPsi.reserve(n), dP.reserve(9 * (size_t)n), dD.reserve(dPdF ? 81 * (size_t)n : 1);
    HOT_HIP(hipMemcpyAsync(dF.p, F, 9 * (size_t)n * sizeof(T), hipMemcpyDefault, stream));
    HOT_HIP(hipMemcpyAsync(dMu.p, mu, (size_t)n * sizeof(T), hipMemcpyDefault, stream));
    HOT_HIP(hipMemcpyAsync(dLam.p, lambda, (size_t)n * sizeof(T), hipMemcpyDefault, stream));
    HOT_LAUNCH(this, "constitutive_eval", k_constitutive_eval<T>, div_up(n, 256), 256, 0, dF.p, dMu.p, dLam.p, n, project, psi ? dPsi.p : (T*)nullptr, P ? dP.p : (T*)nullptr, dPdF ? dD.p : (T*)nullptr);
    download(psi, dPsi.p, n), download(P, dP.p, 9 * (size_t)n), download(dPdF, dD.p, 81 * (size_t)n);
    sync();
}

template <class T>
void Ctx<T>::plasticity_eval(int32_t kind, int32_t n, void* F, void* mu, void* lambda, void* Jp)
{
    need(kind == 1 || kind == 2, "hot_plasticity_eval: kind must be 1 (von Mises) or 2 (snow)");
    need(n > 0 && F && mu && lambda && (kind == 1 || Jp), "hot_plasticity_eval: F, mu, lambda (and Jp for snow) are required");
    DBuf<T> dF, dMu, dLam, dJp;
    dF.reserve(9 * (size_t)n), dMu.reserve(n), dLam.reserve(n), dJp.reserve(n);
    HOT_HIP(hipMemcpyAsync(dF.p, F, 9 * (size_t)n * sizeof(T), hipMemcpyDefault, stream));
    HOT_HIP(hipMemcpyAsync(dMu.p, mu, (size_t)n * sizeof(T), hipMemcpyDefault, stream));
    HOT_HIP(hipMemcpyAsync(dLam.p, lambda, (size_t)n * sizeof(T), hipMemcpyDefault, stream));
    if (Jp) HOT_HIP(hipMemcpyAsync(dJp.p, Jp, (size_t)n * sizeof(T), hipMemcpyDefault, stream));
    HOT_LAUNCH(this, "plasticity_eval", k_plasticity_eval<T>, div_up(n, 256), 256, 0, kind, n, dF.p, dMu.p, dLam.p, dJp.p, (T)cfg.yield_stress, (T)cfg.snow[0], (T)cfg.snow[1], (T)cfg.snow[2],
        (T)cfg.snow[3], (T)cfg.snow[4]);
    download(F, dF.p, 9 * (size_t)n), download(mu, dMu.p, n), download(lambda, dLam.p, n);
    if (kind == 2) download(Jp, dJp.p, n);
    sync();
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
