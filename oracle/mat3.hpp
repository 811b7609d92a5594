// ORACLE (test infrastructure only — never linked into the product library).
//
// Small dense 3x3 algebra for the CPU restatement.  Matrices are column-major like the reference's
// Eigen::Matrix<T,3,3> (element (r,c) at a[c*3+r]), which is also the layout of the C ABI's
// 9-scalar particle attributes (F, C).
//
// Restates, function by function:
//   svd3()            reference Lib/Ziran/Math/Linear/ImplicitQRSVD.h:355-516,518-533 (3x3 implicit-shift QR
//                     SVD, "rotation variant": U,V in SO(3), |s0|>=|s1|>=|s2|, s0>=s1>=0, sign on s2),
//                     helpers :233-245 (wilkinsonShift), :250-266 (process), :271-276 (flipSign),
//                     :281-353 (sort<0>/<1>), 2x2 SVD :104-163, 2x2 polar :45-68;
//                     Givens rotations reference Lib/Ziran/Math/Linear/Givens.h:30-183, zeroChase :196-240,
//                     makeUpperBidiag :253-271.
//   cofactor()        reference Lib/Ziran/Math/Linear/DenseExt.h:240-252.
//   make_pd3/2()      reference Lib/Ziran/Math/Linear/EigenDecomposition.h:126-135 (clamp negative eigenvalues of
//                     a symmetric matrix to 0).  The reference calls Eigen::SelfAdjointEigenSolver (Eigen is
//                     un-vendored and un-pinned, reference Lib/Ziran/CMakeLists.txt:1); the projected matrix
//                     V max(D,0) V^T is unique whatever eigen-solver is used, so a cyclic Jacobi iteration
//                     (Golub & Van Loan, Alg. 8.5.2) run to machine precision is used here.
#pragma once
#include <cmath>
#include <algorithm>
#include <limits>

namespace hot_oracle {

template <class T>
struct V3 {
    T a[3];
    T& operator()(int i) { return a[i]; }
    const T& operator()(int i) const { return a[i]; }
    static V3 zero() { return V3{ { 0, 0, 0 } }; }
    V3 operator+(const V3& o) const { return V3{ { a[0] + o.a[0], a[1] + o.a[1], a[2] + o.a[2] } }; }
    V3 operator-(const V3& o) const { return V3{ { a[0] - o.a[0], a[1] - o.a[1], a[2] - o.a[2] } }; }
    V3 operator*(T s) const { return V3{ { a[0] * s, a[1] * s, a[2] * s } }; }
    V3& operator+=(const V3& o)
    {
        a[0] += o.a[0], a[1] += o.a[1], a[2] += o.a[2];
        return *this;
    }
    V3& operator-=(const V3& o)
    {
        a[0] -= o.a[0], a[1] -= o.a[1], a[2] -= o.a[2];
        return *this;
    }
    T dot(const V3& o) const { return a[0] * o.a[0] + a[1] * o.a[1] + a[2] * o.a[2]; }
    T squaredNorm() const { return dot(*this); }
};

template <class T>
struct M3 {
    T a[9]; // column-major
    T& operator()(int r, int c) { return a[c * 3 + r]; }
    const T& operator()(int r, int c) const { return a[c * 3 + r]; }
    static M3 zero()
    {
        M3 m;
        for (int i = 0; i < 9; ++i) m.a[i] = 0;
        return m;
    }
    static M3 identity()
    {
        M3 m = zero();
        m(0, 0) = m(1, 1) = m(2, 2) = 1;
        return m;
    }
    M3 operator+(const M3& o) const
    {
        M3 m;
        for (int i = 0; i < 9; ++i) m.a[i] = a[i] + o.a[i];
        return m;
    }
    M3 operator-(const M3& o) const
    {
        M3 m;
        for (int i = 0; i < 9; ++i) m.a[i] = a[i] - o.a[i];
        return m;
    }
    M3 operator*(T s) const
    {
        M3 m;
        for (int i = 0; i < 9; ++i) m.a[i] = a[i] * s;
        return m;
    }
    M3& operator+=(const M3& o)
    {
        for (int i = 0; i < 9; ++i) a[i] += o.a[i];
        return *this;
    }
    M3& operator-=(const M3& o)
    {
        for (int i = 0; i < 9; ++i) a[i] -= o.a[i];
        return *this;
    }
    M3 operator*(const M3& o) const
    {
        M3 m;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) m(r, c) = (*this)(r, 0) * o(0, c) + (*this)(r, 1) * o(1, c) + (*this)(r, 2) * o(2, c);
        return m;
    }
    V3<T> operator*(const V3<T>& v) const
    {
        V3<T> r;
        for (int i = 0; i < 3; ++i) r(i) = (*this)(i, 0) * v(0) + (*this)(i, 1) * v(1) + (*this)(i, 2) * v(2);
        return r;
    }
    M3 transpose() const
    {
        M3 m;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) m(r, c) = (*this)(c, r);
        return m;
    }
    T squaredNorm() const
    {
        T s = 0;
        for (int i = 0; i < 9; ++i) s += a[i] * a[i];
        return s;
    }
    T determinant() const
    {
        const M3& F = *this;
        return F(0, 0) * (F(1, 1) * F(2, 2) - F(1, 2) * F(2, 1)) - F(0, 1) * (F(1, 0) * F(2, 2) - F(1, 2) * F(2, 0)) + F(0, 2) * (F(1, 0) * F(2, 1) - F(1, 1) * F(2, 0));
    }
};

template <class T>
inline M3<T> outer(const V3<T>& a, const V3<T>& b)
{
    M3<T> m;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) m(r, c) = a(r) * b(c);
    return m;
}

// J F^{-T}: reference DenseExt.h:240-252
template <class T>
inline M3<T> cofactor(const M3<T>& F)
{
    M3<T> A;
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
inline M3<T> inverse(const M3<T>& F)
{
    // A^{-1} = cof(A)^T / det
    M3<T> c = cofactor(F);
    T d = F.determinant();
    return c.transpose() * ((T)1 / d);
}

// ---------------------------------------------------------------- Givens (reference Givens.h:30-183)
template <class T>
struct Givens {
    int rowi, rowk;
    T c, s;
    Givens(int i, int k)
        : rowi(i), rowk(k), c(1), s(0) {}
    Givens(T a, T b, int i, int k)
        : rowi(i), rowk(k) { compute(a, b); }
    void compute(T a, T b)
    {
        T d = a * a + b * b;
        c = 1;
        s = 0;
        T sqrtd = std::sqrt(d);
        if (sqrtd) {
            T t = 1 / sqrtd;
            c = a * t;
            s = -b * t;
        }
    }
    void computeUnconventional(T a, T b)
    {
        T d = a * a + b * b;
        c = 0;
        s = 1;
        T sqrtd = std::sqrt(d);
        if (sqrtd) {
            T t = 1 / sqrtd;
            s = a * t;
            c = b * t;
        }
    }
    void rowRotation(M3<T>& A) const
    {
        for (int j = 0; j < 3; ++j) {
            T tau1 = A(rowi, j), tau2 = A(rowk, j);
            A(rowi, j) = c * tau1 - s * tau2;
            A(rowk, j) = s * tau1 + c * tau2;
        }
    }
    void columnRotation(M3<T>& A) const
    {
        for (int j = 0; j < 3; ++j) {
            T tau1 = A(j, rowi), tau2 = A(j, rowk);
            A(j, rowi) = c * tau1 - s * tau2;
            A(j, rowk) = s * tau1 + c * tau2;
        }
    }
    void operator*=(const Givens& A)
    {
        T new_c = c * A.c - s * A.s;
        T new_s = s * A.c + c * A.s;
        c = new_c;
        s = new_s;
    }
};

// reference Givens.h:196-240
template <class T>
inline void zeroChase(M3<T>& H, M3<T>& U, M3<T>& V)
{
    Givens<T> r1(H(0, 0), H(1, 0), 0, 1);
    Givens<T> r2(1, 2);
    if (H(1, 0) != 0)
        r2.compute(H(0, 0) * H(0, 1) + H(1, 0) * H(1, 1), H(0, 0) * H(0, 2) + H(1, 0) * H(1, 2));
    else
        r2.compute(H(0, 1), H(0, 2));
    r1.rowRotation(H);
    r2.columnRotation(H);
    r2.columnRotation(V);
    Givens<T> r3(H(1, 1), H(2, 1), 1, 2);
    r3.rowRotation(H);
    r1.columnRotation(U);
    r3.columnRotation(U);
}

// reference Givens.h:253-271
template <class T>
inline void makeUpperBidiag(M3<T>& H, M3<T>& U, M3<T>& V)
{
    U = M3<T>::identity();
    V = M3<T>::identity();
    Givens<T> r(H(1, 0), H(2, 0), 1, 2);
    r.rowRotation(H);
    r.columnRotation(U);
    zeroChase(H, U, V);
}

// 2x2 polar + SVD of the sub-block B(t..t+1, t..t+1) in Givens form: reference ImplicitQRSVD.h:45-68,104-163
template <class T>
inline void svd2_givens(T a00, T a01, T a10, T a11, Givens<T>& U, T& sig0, T& sig1, Givens<T>& V)
{
    // polar
    T x0 = a00 + a11, x1 = a10 - a01;
    T denominator = std::sqrt(x0 * x0 + x1 * x1);
    U.c = 1;
    U.s = 0;
    if (denominator != 0) {
        U.c = x0 / denominator;
        U.s = -x1 / denominator;
    }
    // S = R^T-rotated A (rowRotation with rowi=0,rowk=1 on the 2x2)
    T s00 = U.c * a00 - U.s * a10, s01 = U.c * a01 - U.s * a11;
    T /*s10 = U.s * a00 + U.c * a10,*/ s11 = U.s * a01 + U.c * a11;
    T cosine, sine;
    T x = s00, y = s01, z = s11;
    T y2 = y * y;
    if (y2 == 0) {
        cosine = 1;
        sine = 0;
        sig0 = x;
        sig1 = z;
    }
    else {
        T tau = (T)0.5 * (x - z);
        T w = std::sqrt(tau * tau + y2);
        T t;
        if (tau > 0)
            t = y / (tau + w);
        else
            t = y / (tau - w);
        cosine = (T)1 / std::sqrt(t * t + (T)1);
        sine = -t * cosine;
        T c2 = cosine * cosine;
        T csy = 2 * cosine * sine * y;
        T s2 = sine * sine;
        sig0 = c2 * x - csy + s2 * z;
        sig1 = s2 * x + csy + c2 * z;
    }
    if (sig0 < sig1) {
        std::swap(sig0, sig1);
        V.c = -sine;
        V.s = cosine;
    }
    else {
        V.c = cosine;
        V.s = sine;
    }
    U *= V;
}

template <class T>
inline T wilkinsonShift(T a1, T b1, T a2)
{
    T d = (T)0.5 * (a1 - a2);
    T bs = b1 * b1;
    return a2 - std::copysign(bs / (std::fabs(d) + std::sqrt(d * d + bs)), d);
}

template <class T>
inline void svd_process(int t, M3<T>& B, M3<T>& U, V3<T>& sigma, M3<T>& V)
{
    int other = (t == 1) ? 0 : 2;
    Givens<T> u(0, 1), v(0, 1);
    sigma(other) = B(other, other);
    T s0, s1;
    svd2_givens(B(t, t), B(t, t + 1), B(t + 1, t), B(t + 1, t + 1), u, s0, s1, v);
    sigma(t) = s0;
    sigma(t + 1) = s1;
    u.rowi += t, u.rowk += t, v.rowi += t, v.rowk += t;
    u.columnRotation(U);
    v.columnRotation(V);
}

template <class T>
inline void svd_flipSign(int i, M3<T>& U, V3<T>& sigma)
{
    sigma(i) = -sigma(i);
    for (int r = 0; r < 3; ++r) U(r, i) = -U(r, i);
}
template <class T>
inline void swapCols(M3<T>& A, int i, int j)
{
    for (int r = 0; r < 3; ++r) std::swap(A(r, i), A(r, j));
}
template <class T>
inline void negCol(M3<T>& A, int i)
{
    for (int r = 0; r < 3; ++r) A(r, i) = -A(r, i);
}

template <class T>
inline void svd_sort0(M3<T>& U, V3<T>& sigma, M3<T>& V)
{
    if (std::fabs(sigma(1)) >= std::fabs(sigma(2))) {
        if (sigma(1) < 0) {
            svd_flipSign(1, U, sigma);
            svd_flipSign(2, U, sigma);
        }
        return;
    }
    if (sigma(2) < 0) {
        svd_flipSign(1, U, sigma);
        svd_flipSign(2, U, sigma);
    }
    std::swap(sigma(1), sigma(2));
    swapCols(U, 1, 2);
    swapCols(V, 1, 2);
    if (sigma(1) > sigma(0)) {
        std::swap(sigma(0), sigma(1));
        swapCols(U, 0, 1);
        swapCols(V, 0, 1);
    }
    else {
        negCol(U, 2);
        negCol(V, 2);
    }
}

template <class T>
inline void svd_sort1(M3<T>& U, V3<T>& sigma, M3<T>& V)
{
    if (std::fabs(sigma(0)) >= sigma(1)) {
        if (sigma(0) < 0) {
            svd_flipSign(0, U, sigma);
            svd_flipSign(2, U, sigma);
        }
        return;
    }
    std::swap(sigma(0), sigma(1));
    swapCols(U, 0, 1);
    swapCols(V, 0, 1);
    if (std::fabs(sigma(1)) < std::fabs(sigma(2))) {
        std::swap(sigma(1), sigma(2));
        swapCols(U, 1, 2);
        swapCols(V, 1, 2);
    }
    else {
        negCol(U, 1);
        negCol(V, 1);
    }
    if (sigma(1) < 0) {
        svd_flipSign(1, U, sigma);
        svd_flipSign(2, U, sigma);
    }
}

// reference ImplicitQRSVD.h:355-516 ; default tolerances :518-533 (128 eps float, 1024 eps double)
template <class T>
inline int svd3(const M3<T>& A, M3<T>& U, V3<T>& sigma, M3<T>& V)
{
    T tol = (sizeof(T) == 4 ? (T)128 : (T)1024) * std::numeric_limits<T>::epsilon();
    M3<T> B = A;
    U = M3<T>::identity();
    V = M3<T>::identity();
    makeUpperBidiag(B, U, V);
    int count = 0;
    T mu = 0;
    Givens<T> r(0, 1);
    T alpha_1 = B(0, 0), beta_1 = B(0, 1), alpha_2 = B(1, 1), alpha_3 = B(2, 2), beta_2 = B(1, 2);
    T gamma_1 = alpha_1 * beta_1, gamma_2 = alpha_2 * beta_2;
    tol *= std::max((T)0.5 * std::sqrt(alpha_1 * alpha_1 + alpha_2 * alpha_2 + alpha_3 * alpha_3 + beta_1 * beta_1 + beta_2 * beta_2), (T)1);
    while (std::fabs(beta_2) > tol && std::fabs(beta_1) > tol && std::fabs(alpha_1) > tol && std::fabs(alpha_2) > tol && std::fabs(alpha_3) > tol) {
        mu = wilkinsonShift(alpha_2 * alpha_2 + beta_1 * beta_1, gamma_2, alpha_3 * alpha_3 + beta_2 * beta_2);
        r.compute(alpha_1 * alpha_1 - mu, gamma_1);
        r.columnRotation(B);
        r.columnRotation(V);
        zeroChase(B, U, V);
        alpha_1 = B(0, 0), beta_1 = B(0, 1), alpha_2 = B(1, 1), alpha_3 = B(2, 2), beta_2 = B(1, 2);
        gamma_1 = alpha_1 * beta_1, gamma_2 = alpha_2 * beta_2;
        count++;
    }
    if (std::fabs(beta_2) <= tol) {
        svd_process(0, B, U, sigma, V);
        svd_sort0(U, sigma, V);
    }
    else if (std::fabs(beta_1) <= tol) {
        svd_process(1, B, U, sigma, V);
        svd_sort1(U, sigma, V);
    }
    else if (std::fabs(alpha_2) <= tol) {
        Givens<T> r1(1, 2);
        r1.computeUnconventional(B(1, 2), B(2, 2));
        r1.rowRotation(B);
        r1.columnRotation(U);
        svd_process(0, B, U, sigma, V);
        svd_sort0(U, sigma, V);
    }
    else if (std::fabs(alpha_3) <= tol) {
        Givens<T> r1(1, 2);
        r1.compute(B(1, 1), B(1, 2));
        r1.columnRotation(B);
        r1.columnRotation(V);
        Givens<T> r2(0, 2);
        r2.compute(B(0, 0), B(0, 2));
        r2.columnRotation(B);
        r2.columnRotation(V);
        svd_process(0, B, U, sigma, V);
        svd_sort0(U, sigma, V);
    }
    else if (std::fabs(alpha_1) <= tol) {
        Givens<T> r1(0, 1);
        r1.computeUnconventional(B(0, 1), B(1, 1));
        r1.rowRotation(B);
        r1.columnRotation(U);
        Givens<T> r2(0, 2);
        r2.computeUnconventional(B(0, 2), B(2, 2));
        r2.rowRotation(B);
        r2.columnRotation(U);
        svd_process(1, B, U, sigma, V);
        svd_sort1(U, sigma, V);
    }
    return count;
}

// ---------------------------------------------------------------- PSD projection (EigenDecomposition.h:126-135)
// symmetric 3x3: cyclic Jacobi to convergence, clamp eigenvalues at 0, rebuild.
template <class T>
inline void make_pd3(M3<T>& S)
{
    M3<T> A = S, Q = M3<T>::identity();
    for (int sweep = 0; sweep < 64; ++sweep) {
        T off = A(0, 1) * A(0, 1) + A(0, 2) * A(0, 2) + A(1, 2) * A(1, 2);
        T diag = A(0, 0) * A(0, 0) + A(1, 1) * A(1, 1) + A(2, 2) * A(2, 2);
        if (off <= std::numeric_limits<T>::epsilon() * std::numeric_limits<T>::epsilon() * diag || off == 0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                T apq = A(p, q);
                if (apq == 0) continue;
                T theta = (A(q, q) - A(p, p)) / (2 * apq);
                T t = (theta >= 0 ? (T)1 : (T)-1) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                T c = 1 / std::sqrt(t * t + 1), s = t * c;
                // A <- J^T A J with J = [[c, s],[-s, c]] on (p,q)
                for (int k = 0; k < 3; ++k) {
                    T akp = A(k, p), akq = A(k, q);
                    A(k, p) = c * akp - s * akq;
                    A(k, q) = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    T apk = A(p, k), aqk = A(q, k);
                    A(p, k) = c * apk - s * aqk;
                    A(q, k) = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    T qkp = Q(k, p), qkq = Q(k, q);
                    Q(k, p) = c * qkp - s * qkq;
                    Q(k, q) = s * qkp + c * qkq;
                }
            }
    }
    T d[3] = { A(0, 0), A(1, 1), A(2, 2) };
    for (int i = 0; i < 3; ++i)
        if (d[i] < 0) d[i] = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S(r, c) = Q(r, 0) * d[0] * Q(c, 0) + Q(r, 1) * d[1] * Q(c, 1) + Q(r, 2) * d[2] * Q(c, 2);
}

// symmetric 2x2 [[a,b],[b,d]] -> PSD projection (closed-form eigen-decomposition)
template <class T>
inline void make_pd2(T& a, T& b, T& d)
{
    if (b == 0) {
        if (a < 0) a = 0;
        if (d < 0) d = 0;
        return;
    }
    T theta = (d - a) / (2 * b);
    T t = (theta >= 0 ? (T)1 : (T)-1) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
    T c = 1 / std::sqrt(t * t + 1), s = t * c;
    T l0 = a - t * b, l1 = d + t * b; // eigenvalues; eigenvectors (c,-s), (s,c)
    if (l0 < 0) l0 = 0;
    if (l1 < 0) l1 = 0;
    a = c * c * l0 + s * s * l1;
    d = s * s * l0 + c * c * l1;
    b = -c * s * l0 + s * c * l1;
}

} // namespace hot_oracle
