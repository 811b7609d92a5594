// libhotmi355x — register-resident 3x3 algebra for the per-particle constitutive kernels.
//
// svd3(): implicit-shift QR SVD of a 3x3 matrix with the conventions of the reference's
// singularValueDecomposition (Lib/Ziran/Math/Linear/ImplicitQRSVD.h:355-516,518-533; algorithm of Gast et al.
// 2016): U, V are rotations, sigma sorted by magnitude with any negative sign on sigma_2, tolerances
// 128 eps (float) / 1024 eps (double).  P, psi and the PSD-projected dP/dF depend on that convention (they are
// invariant to the remaining sign freedom).  Written for the GPU: every Givens rotation has compile-time row/
// column indices so the 3x3 operands stay in VGPRs (no scratch), branches are the data-dependent deflation cases.
#pragma once
#include "hot_common.h"

namespace hot {

template <class T>
struct Mat3 {
    T a[9]; // column-major: (r,c) -> a[c*3+r]
    __device__ __forceinline__ T& operator()(int r, int c) { return a[c * 3 + r]; }
    __device__ __forceinline__ const T& operator()(int r, int c) const { return a[c * 3 + r]; }
};
template <class T>
__device__ __forceinline__ Mat3<T> m3_identity()
{
    Mat3<T> m;
#pragma unroll
    for (int i = 0; i < 9; ++i) m.a[i] = (i % 4 == 0) ? (T)1 : (T)0;
    return m;
}
template <class T>
__device__ __forceinline__ Mat3<T> m3_mul(const Mat3<T>& A, const Mat3<T>& B)
{
    Mat3<T> C;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) C(r, c) = A(r, 0) * B(0, c) + A(r, 1) * B(1, c) + A(r, 2) * B(2, c);
    return C;
}
// A * B^T
template <class T>
__device__ __forceinline__ Mat3<T> m3_mul_bt(const Mat3<T>& A, const Mat3<T>& B)
{
    Mat3<T> C;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) C(r, c) = A(r, 0) * B(c, 0) + A(r, 1) * B(c, 1) + A(r, 2) * B(c, 2);
    return C;
}
// A^T * B
template <class T>
__device__ __forceinline__ Mat3<T> m3_mul_at(const Mat3<T>& A, const Mat3<T>& B)
{
    Mat3<T> C;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) C(r, c) = A(0, r) * B(0, c) + A(1, r) * B(1, c) + A(2, r) * B(2, c);
    return C;
}
template <class T>
__device__ __forceinline__ T m3_det(const Mat3<T>& F)
{
    return F(0, 0) * (F(1, 1) * F(2, 2) - F(1, 2) * F(2, 1)) - F(0, 1) * (F(1, 0) * F(2, 2) - F(1, 2) * F(2, 0)) + F(0, 2) * (F(1, 0) * F(2, 1) - F(1, 1) * F(2, 0));
}
// J F^{-T} (reference Lib/Ziran/Math/Linear/DenseExt.h:240-252)
template <class T>
__device__ __forceinline__ Mat3<T> m3_cofactor(const Mat3<T>& F)
{
    Mat3<T> A;
    A(0, 0) = F(1, 1) * F(2, 2) - F(1, 2) * F(2, 1);
    A(0, 1) = F(1, 2) * F(2, 0) - F(1, 0) * F(2, 2);
    A(0, 2) = F(1, 0) * F(2, 1) - F(1, 1) * F(2, 0);
    A(1, 0) = F(0, 2) * F(2, 1) - F(0, 1) * F(2, 2);
    A(1, 1) = F(0, 0) * F(2, 2) - F(0, 2) * F(2, 0);
    A(1, 2) = F(0, 1) * F(2, 0) - F(0, 0) * F(2, 1);
    A(2, 0) = F(0, 1) * F(1, 2) - F(0, 2) * F(1, 1);
    A(2, 1) = F(0, 2) * F(1, 0) - F(0, 0) * F(1, 2);
    A(2, 2) = F(0, 0) * F(1, 1) - F(0, 1) * F(1, 0);
    return A;
}
template <class T>
__device__ __forceinline__ Mat3<T> m3_inverse(const Mat3<T>& F)
{
    Mat3<T> c = m3_cofactor(F);
    T inv = (T)1 / m3_det(F);
    Mat3<T> r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r(i, j) = c(j, i) * inv;
    return r;
}

template <class T>
__device__ __forceinline__ T hsqrt(T x);
template <>
__device__ __forceinline__ float hsqrt<float>(float x) { return sqrtf(x); }
template <>
__device__ __forceinline__ double hsqrt<double>(double x) { return sqrt(x); }
template <class T>
__device__ __forceinline__ T habs(T x) { return x < 0 ? -x : x; }

// sqrt(d) and 1 / sqrt(d) together, for the rotations (c, s) = (a, b) / sqrt(a^2 + b^2).  Written as sqrt followed by a division, the
// compiler emits its correctly rounded sequences for both (range scaling, v_rsq + Goldschmidt, v_div_scale / v_rcp / v_div_fmas /
// v_div_fixup): ~28 of the per-particle kernels' instructions per rotation, and FP64 instructions issue at half rate — k_state is bound
// by them (SQ counters: 21 M of 36 M VALU instructions are FP64 FMAs at C2).  Here: one v_rsq, two coupled Newton (Goldschmidt) steps
// for both values and a last correction of the root: 11 instructions, both results within 1-2 ulp.  Outside [1e-280, 1e280] (and for 0)
// the plain forms are used.
template <class T>
__device__ __forceinline__ void hrsqrt2(T d, T& sq, T& inv);
template <>
__device__ __forceinline__ void hrsqrt2<double>(double d, double& sq, double& inv)
{
    if (d > 1e-280 && d < 1e280) {
        const double y = __builtin_amdgcn_rsq(d);
        double g = d * y, h = 0.5 * y;
        double r = fma(-h, g, 0.5);
        g = fma(g, r, g), h = fma(h, r, h);
        r = fma(-h, g, 0.5);
        g = fma(g, r, g), h = fma(h, r, h);
        g = fma(fma(-g, g, d), h, g);
        sq = g, inv = h + h;
    }
    else {
        sq = sqrt(d);
        inv = sq != 0.0 ? 1.0 / sq : 0.0;
    }
}
template <>
__device__ __forceinline__ void hrsqrt2<float>(float d, float& sq, float& inv)
{
    if (d > 1e-30f && d < 1e30f) {
        float y = __builtin_amdgcn_rsqf(d); // 1 ulp
        y = fmaf(y * 0.5f, fmaf(-d * y, y, 1.0f), y); // one Newton step
        inv = y;
        const float g = d * y;
        sq = fmaf(fmaf(-g, g, d), 0.5f * y, g);
    }
    else {
        sq = sqrtf(d);
        inv = sq != 0.0f ? 1.0f / sq : 0.0f;
    }
}

// Givens pair (c,s): [c -s; s c] applied to rows (I,K) / columns (I,K)
template <class T>
struct Giv {
    T c, s;
};
template <class T>
__device__ __forceinline__ Giv<T> giv_compute(T a, T b) // (c -s; s c)(a;b) = (*;0)
{
    Giv<T> g{ (T)1, (T)0 };
    T d = a * a + b * b;
    T sq, t;
    hrsqrt2(d, sq, t);
    if (sq != (T)0) {
        g.c = a * t;
        g.s = -b * t;
    }
    return g;
}
template <class T>
__device__ __forceinline__ Giv<T> giv_unconventional(T a, T b) // (c -s; s c)(a;b) = (0;*)
{
    Giv<T> g{ (T)0, (T)1 };
    T d = a * a + b * b;
    T sq, t;
    hrsqrt2(d, sq, t);
    if (sq != (T)0) {
        g.s = a * t;
        g.c = b * t;
    }
    return g;
}
template <int I, int K, class T>
__device__ __forceinline__ void giv_rows(const Giv<T>& g, Mat3<T>& A)
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T t1 = A(I, j), t2 = A(K, j);
        A(I, j) = g.c * t1 - g.s * t2;
        A(K, j) = g.s * t1 + g.c * t2;
    }
}
template <int I, int K, class T>
__device__ __forceinline__ void giv_cols(const Giv<T>& g, Mat3<T>& A)
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T t1 = A(j, I), t2 = A(j, K);
        A(j, I) = g.c * t1 - g.s * t2;
        A(j, K) = g.s * t1 + g.c * t2;
    }
}

template <class T>
__device__ __forceinline__ void zero_chase(Mat3<T>& H, Mat3<T>& U, Mat3<T>& V)
{
    Giv<T> r1 = giv_compute(H(0, 0), H(1, 0));
    Giv<T> r2;
    if (H(1, 0) != (T)0)
        r2 = giv_compute(H(0, 0) * H(0, 1) + H(1, 0) * H(1, 1), H(0, 0) * H(0, 2) + H(1, 0) * H(1, 2));
    else
        r2 = giv_compute(H(0, 1), H(0, 2));
    giv_rows<0, 1>(r1, H);
    giv_cols<1, 2>(r2, H);
    giv_cols<1, 2>(r2, V);
    Giv<T> r3 = giv_compute(H(1, 1), H(2, 1));
    giv_rows<1, 2>(r3, H);
    giv_cols<0, 1>(r1, U);
    giv_cols<1, 2>(r3, U);
}

// 2x2 polar + SVD in Givens form (reference ImplicitQRSVD.h:45-68,104-163)
template <class T>
__device__ __forceinline__ void svd2(T a00, T a01, T a10, T a11, Giv<T>& U, T& s0, T& s1, Giv<T>& V)
{
    T x0 = a00 + a11, x1 = a10 - a01;
    T den, iden;
    hrsqrt2(x0 * x0 + x1 * x1, den, iden);
    U.c = (T)1, U.s = (T)0;
    if (den != (T)0) {
        U.c = x0 * iden;
        U.s = -x1 * iden;
    }
    T x = U.c * a00 - U.s * a10, y = U.c * a01 - U.s * a11, z = U.s * a01 + U.c * a11;
    T cosine, sine;
    T y2 = y * y;
    if (y2 == (T)0) {
        cosine = (T)1, sine = (T)0;
        s0 = x, s1 = z;
    }
    else {
        T tau = (T)0.5 * (x - z);
        T w = hsqrt(tau * tau + y2);
        T t = (tau > (T)0) ? y / (tau + w) : y / (tau - w);
        T sq1;
        hrsqrt2(t * t + (T)1, sq1, cosine);
        sine = -t * cosine;
        T c2 = cosine * cosine, csy = (T)2 * cosine * sine * y, s2 = sine * sine;
        s0 = c2 * x - csy + s2 * z;
        s1 = s2 * x + csy + c2 * z;
    }
    if (s0 < s1) {
        T tmp = s0;
        s0 = s1, s1 = tmp;
        V.c = -sine, V.s = cosine;
    }
    else {
        V.c = cosine, V.s = sine;
    }
    T nc = U.c * V.c - U.s * V.s, ns = U.s * V.c + U.c * V.s;
    U.c = nc, U.s = ns;
}

template <int I, class T>
__device__ __forceinline__ void neg_col(Mat3<T>& A)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) A(r, I) = -A(r, I);
}
template <int I, int J, class T>
__device__ __forceinline__ void swap_cols(Mat3<T>& A)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T t = A(r, I);
        A(r, I) = A(r, J);
        A(r, J) = t;
    }
}
template <class T>
__device__ __forceinline__ void hswap(T& a, T& b)
{
    T t = a;
    a = b, b = t;
}

template <class T>
__device__ __forceinline__ void svd_tail0(Mat3<T>& B, Mat3<T>& U, T (&sg)[3], Mat3<T>& V) // process<0> + sort<0>
{
    Giv<T> u, v;
    sg[2] = B(2, 2);
    svd2(B(0, 0), B(0, 1), B(1, 0), B(1, 1), u, sg[0], sg[1], v);
    giv_cols<0, 1>(u, U);
    giv_cols<0, 1>(v, V);
    if (habs(sg[1]) >= habs(sg[2])) {
        if (sg[1] < (T)0) {
            sg[1] = -sg[1], sg[2] = -sg[2];
            neg_col<1>(U), neg_col<2>(U);
        }
        return;
    }
    if (sg[2] < (T)0) {
        sg[1] = -sg[1], sg[2] = -sg[2];
        neg_col<1>(U), neg_col<2>(U);
    }
    hswap(sg[1], sg[2]);
    swap_cols<1, 2>(U), swap_cols<1, 2>(V);
    if (sg[1] > sg[0]) {
        hswap(sg[0], sg[1]);
        swap_cols<0, 1>(U), swap_cols<0, 1>(V);
    }
    else {
        neg_col<2>(U), neg_col<2>(V);
    }
}
template <class T>
__device__ __forceinline__ void svd_tail1(Mat3<T>& B, Mat3<T>& U, T (&sg)[3], Mat3<T>& V) // process<1> + sort<1>
{
    Giv<T> u, v;
    sg[0] = B(0, 0);
    svd2(B(1, 1), B(1, 2), B(2, 1), B(2, 2), u, sg[1], sg[2], v);
    giv_cols<1, 2>(u, U);
    giv_cols<1, 2>(v, V);
    if (habs(sg[0]) >= sg[1]) {
        if (sg[0] < (T)0) {
            sg[0] = -sg[0], sg[2] = -sg[2];
            neg_col<0>(U), neg_col<2>(U);
        }
        return;
    }
    hswap(sg[0], sg[1]);
    swap_cols<0, 1>(U), swap_cols<0, 1>(V);
    if (habs(sg[1]) < habs(sg[2])) {
        hswap(sg[1], sg[2]);
        swap_cols<1, 2>(U), swap_cols<1, 2>(V);
    }
    else {
        neg_col<1>(U), neg_col<1>(V);
    }
    if (sg[1] < (T)0) {
        sg[1] = -sg[1], sg[2] = -sg[2];
        neg_col<1>(U), neg_col<2>(U);
    }
}

template <class T>
__device__ inline void svd3(const Mat3<T>& A, Mat3<T>& U, T (&sg)[3], Mat3<T>& V)
{
    constexpr T eps = sizeof(T) == 4 ? (T)1.1920928955078125e-07 : (T)2.220446049250313e-16;
    T tol = (sizeof(T) == 4 ? (T)128 : (T)1024) * eps;
    Mat3<T> B = A;
    U = m3_identity<T>();
    V = m3_identity<T>();
    { // makeUpperBidiag
        Giv<T> r = giv_compute(B(1, 0), B(2, 0));
        giv_rows<1, 2>(r, B);
        giv_cols<1, 2>(r, U);
        zero_chase(B, U, V);
    }
    T alpha_1 = B(0, 0), beta_1 = B(0, 1), alpha_2 = B(1, 1), alpha_3 = B(2, 2), beta_2 = B(1, 2);
    T gamma_1 = alpha_1 * beta_1, gamma_2 = alpha_2 * beta_2;
    T nrm = (T)0.5 * hsqrt(alpha_1 * alpha_1 + alpha_2 * alpha_2 + alpha_3 * alpha_3 + beta_1 * beta_1 + beta_2 * beta_2);
    tol *= (nrm > (T)1 ? nrm : (T)1);
    int guard = 0;
    while (habs(beta_2) > tol && habs(beta_1) > tol && habs(alpha_1) > tol && habs(alpha_2) > tol && habs(alpha_3) > tol && guard++ < 64) {
        // Wilkinson shift
        T a1 = alpha_2 * alpha_2 + beta_1 * beta_1, b1 = gamma_2, a2 = alpha_3 * alpha_3 + beta_2 * beta_2;
        T d = (T)0.5 * (a1 - a2);
        T bs = b1 * b1;
        T q = bs / (habs(d) + hsqrt(d * d + bs));
        T mu = a2 - (d < (T)0 || (d == (T)0 && (1 / d) < 0) ? -q : q); // copysign(q, d)
        Giv<T> r = giv_compute(alpha_1 * alpha_1 - mu, gamma_1);
        giv_cols<0, 1>(r, B);
        giv_cols<0, 1>(r, V);
        zero_chase(B, U, V);
        alpha_1 = B(0, 0), beta_1 = B(0, 1), alpha_2 = B(1, 1), alpha_3 = B(2, 2), beta_2 = B(1, 2);
        gamma_1 = alpha_1 * beta_1, gamma_2 = alpha_2 * beta_2;
    }
    if (habs(beta_2) <= tol) {
        svd_tail0(B, U, sg, V);
    }
    else if (habs(beta_1) <= tol) {
        svd_tail1(B, U, sg, V);
    }
    else if (habs(alpha_2) <= tol) {
        Giv<T> r1 = giv_unconventional(B(1, 2), B(2, 2));
        giv_rows<1, 2>(r1, B);
        giv_cols<1, 2>(r1, U);
        svd_tail0(B, U, sg, V);
    }
    else if (habs(alpha_3) <= tol) {
        Giv<T> r1 = giv_compute(B(1, 1), B(1, 2));
        giv_cols<1, 2>(r1, B);
        giv_cols<1, 2>(r1, V);
        Giv<T> r2 = giv_compute(B(0, 0), B(0, 2));
        giv_cols<0, 2>(r2, B);
        giv_cols<0, 2>(r2, V);
        svd_tail0(B, U, sg, V);
    }
    else {
        Giv<T> r1 = giv_unconventional(B(0, 1), B(1, 1));
        giv_rows<0, 1>(r1, B);
        giv_cols<0, 1>(r1, U);
        Giv<T> r2 = giv_unconventional(B(0, 2), B(2, 2));
        giv_rows<0, 2>(r2, B);
        giv_cols<0, 2>(r2, U);
        svd_tail1(B, U, sg, V);
    }
}

// PSD projection of a symmetric 3x3 (EigenDecomposition.h:126-135) by cyclic Jacobi; S is overwritten
template <class T>
__device__ inline void make_pd3(Mat3<T>& S)
{
    Mat3<T> A = S, Q = m3_identity<T>();
    constexpr T eps = sizeof(T) == 4 ? (T)1.1920928955078125e-07 : (T)2.220446049250313e-16;
    for (int sweep = 0; sweep < 12; ++sweep) {
        T off = A(0, 1) * A(0, 1) + A(0, 2) * A(0, 2) + A(1, 2) * A(1, 2);
        T dg = A(0, 0) * A(0, 0) + A(1, 1) * A(1, 1) + A(2, 2) * A(2, 2);
        if (off <= eps * eps * dg || off == (T)0) break;
#define HOT_JACOBI(P_, Q_)                                                              \
    {                                                                                   \
        T apq = A(P_, Q_);                                                              \
        if (apq != (T)0) {                                                              \
            T theta = (A(Q_, Q_) - A(P_, P_)) / ((T)2 * apq);                           \
            T t = (theta >= (T)0 ? (T)1 : (T)-1) / (habs(theta) + hsqrt(theta * theta + (T)1)); \
            Giv<T> g;                                                                   \
            T sq1_;                                                                     \
            hrsqrt2(t * t + (T)1, sq1_, g.c);                                           \
            g.s = t * g.c;                                                              \
            giv_cols<P_, Q_>(g, A);                                                     \
            giv_rows<P_, Q_>(g, A);                                                     \
            giv_cols<P_, Q_>(g, Q);                                                     \
        }                                                                               \
    }
        HOT_JACOBI(0, 1)
        HOT_JACOBI(0, 2)
        HOT_JACOBI(1, 2)
#undef HOT_JACOBI
    }
    T d0 = A(0, 0) < (T)0 ? (T)0 : A(0, 0), d1 = A(1, 1) < (T)0 ? (T)0 : A(1, 1), d2 = A(2, 2) < (T)0 ? (T)0 : A(2, 2);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) S(r, c) = Q(r, 0) * d0 * Q(c, 0) + Q(r, 1) * d1 * Q(c, 1) + Q(r, 2) * d2 * Q(c, 2);
}
// symmetric 2x2 [[a,b],[b,d]]
template <class T>
__device__ __forceinline__ void make_pd2(T& a, T& b, T& d)
{
    if (b == (T)0) {
        if (a < (T)0) a = (T)0;
        if (d < (T)0) d = (T)0;
        return;
    }
    T theta = (d - a) / ((T)2 * b);
    T t = (theta >= (T)0 ? (T)1 : (T)-1) / (habs(theta) + hsqrt(theta * theta + (T)1));
    T sq1, c;
    hrsqrt2(t * t + (T)1, sq1, c);
    T s = t * c;
    T l0 = a - t * b, l1 = d + t * b;
    if (l0 < (T)0) l0 = (T)0;
    if (l1 < (T)0) l1 = (T)0;
    a = c * c * l0 + s * s * l1;
    d = s * s * l0 + c * c * l1;
    b = -c * s * l0 + s * c * l1;
}

} // namespace hot
