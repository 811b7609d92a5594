// libhotmi355x — operator applications: block-ELL SpMV, restriction / prolongation, smoothers, V-cycle.
//
//   k_spmv          SquareMatrix::multiply (reference Projects/multigrid/SquareMatrix.h:477-487) — THE bandwidth consumer
//                   (SURVEY §8a row 17/21).  One wavefront per block row: a row is 125 contiguous 3x3 blocks (9000 B fp64),
//                   lane l owns slots l and l+64, so the 64 lanes stream the row in two fully coalesced sweeps; x is
//                   gathered per slot, the three row sums are reduced with __shfl_xor.
//   k_gs_color      MultigridOperator::gs_smooth (Projects/multigrid/MultigridPreconditioner.h:266-318): symmetric coloured
//                   block Gauss–Seidel in the reference's exact node order (colour, first-touch block, id).  One wavefront
//                   per 4^3-node block walks its nodes sequentially; the row sweep is the SpMV sweep with the ordering
//                   predicate on the packed colour key, values of the block's own earlier nodes come from LDS.
//   restrict/prolong SparseMPMMatrix::transposeMultiply / multiply on the transfer matrices (MPMMultigridMatrix.h:63-70) as
//                   pure gathers over the child / parent tables with scalar weights (the reference stores 3x3 w*I blocks).
//   smooth_dev      jacobi_smooth :160-173, optimal_jacobi_smooth :174-189, cg_smooth :190-226, gs_smooth :266-318
//   vcycle_dev      MultigridOperator::operator() :362-421 with setup_parameters :525-551
#include "hot_impl.h"
#include "hot_svd.h"
#include <cstdlib>

namespace hot {

// ------------------------------------------------------------------------------------------------ vector ops
template <class T>
__global__ void k_axpy(size_t n, T a, const T* __restrict__ x, T* y)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
template <class T>
__global__ void k_axpy_dev(size_t n, const double* a, double sign, const T* __restrict__ x, T* y)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    T s = (T)(sign * (*a));
    if (i < n) y[i] += s * x[i];
}
// y = x + (*a) * y
template <class T>
__global__ void k_xpay_dev(size_t n, const double* a, const T* __restrict__ x, T* y)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    T s = (T)(*a);
    if (i < n) y[i] = x[i] + s * y[i];
}
template <class T>
__global__ __launch_bounds__(256) void k_dot(size_t n, const T* __restrict__ x, const T* __restrict__ y, double* out)
{
    __shared__ double red[4];
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (double)(x[i] * y[i]);
    double t = block_sum_256<double>(s, red);
    if (threadIdx.x == 0) atomic_add(out, t);
}
template <class T>
void Ctx<T>::axpy(size_t n, T a, const T* x, T* y)
{
    HOT_LAUNCH(this, "axpy", k_axpy<T>, div_up(n, 256), 256, 0, n, a, x, y);
}
template <class T>
void Ctx<T>::axpy_dev(size_t n, const double* a, double sign, const T* x, T* y)
{
    HOT_LAUNCH(this, "axpy", k_axpy_dev<T>, div_up(n, 256), 256, 0, n, a, sign, x, y);
}
template <class T>
void Ctx<T>::copy(size_t n, const T* x, T* y)
{
    HOT_HIP(hipMemcpyAsync(y, x, n * sizeof(T), hipMemcpyDeviceToDevice, stream));
}
template <class T>
void Ctx<T>::zero(size_t n, T* y)
{
    HOT_HIP(hipMemsetAsync(y, 0, n * sizeof(T), stream));
}
template <class T>
void Ctx<T>::dot_to(size_t n, const T* x, const T* y, double* out)
{
    HOT_HIP(hipMemsetAsync(out, 0, sizeof(double), stream));
    HOT_LAUNCH(this, "dot", k_dot<T>, std::min(div_up(n, 1024), 256), 256, 0, n, x, y, out); // <= 256 same-address atomics
}
template <class T>
double Ctx<T>::dot_host(size_t n, const T* x, const T* y)
{
    dot_to(n, x, y, dscal.p + 100);
    HOT_HIP(hipMemcpyAsync(hscal + 100, dscal.p + 100, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    return hscal[100];
}

// ------------------------------------------------------------------------------------------------ SpMV
template <class T>
__global__ __launch_bounds__(256) void k_spmv(const int32_t* __restrict__ col, const T* __restrict__ val, const T* __restrict__ x, T* __restrict__ y, int n)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int32_t* c = col + (int64_t)row * 125;
    const T* v = val + (int64_t)row * 1125;
    T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int k = lane + 64 * r;
        if (k < 125) {
            int j = c[k];
            const T* b = v + k * 9;
            T x0 = x[3 * (int64_t)j], x1 = x[3 * (int64_t)j + 1], x2 = x[3 * (int64_t)j + 2];
            s0 += b[0] * x0 + b[3] * x1 + b[6] * x2;
            s1 += b[1] * x0 + b[4] * x1 + b[7] * x2;
            s2 += b[2] * x0 + b[5] * x1 + b[8] * x2;
        }
    }
    s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
    if (lane == 0) y[3 * (int64_t)row] = s0, y[3 * (int64_t)row + 1] = s1, y[3 * (int64_t)row + 2] = s2;
}
template <class T>
__global__ void k_scal_v(size_t n, T a, T* x)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= a;
}
template <class T>
void Ctx<T>::scal(size_t n, T a, T* x)
{
    HOT_LAUNCH(this, "scal", k_scal_v<T>, div_up(n, 256), 256, 0, n, a, x);
}
template <class T>
void Ctx<T>::spmv_dev(Level<T>& L, const T* x, T* y)
{
    HOT_LAUNCH(this, lname("spmv", L.id).c_str(), k_spmv<T>, div_up(L.n, 4), 256, 0, L.col.p, L.val.p, x, y, L.n);
}

// ------------------------------------------------------------------------------------------------ transfers
template <class T>
__global__ void k_restrict(const int32_t* __restrict__ child, const T* __restrict__ fine, T* coarse, int nc)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * nc) return;
    int I = e / 3, d = e - 3 * I;
    T s = 0;
    for (int q = 0; q < 27; ++q) {
        int ci = child[I * 27 + q];
        if (ci < 0) continue;
        T w = ((q / 9 != 1) ? (T)0.5 : (T)1) * (((q / 3) % 3 != 1) ? (T)0.5 : (T)1) * ((q % 3 != 1) ? (T)0.5 : (T)1);
        s += w * fine[3 * (int64_t)ci + d];
    }
    coarse[e] = s;
}
template <class T>
__global__ void k_prolong(const int32_t* __restrict__ pcol, const T* __restrict__ pw, const T* __restrict__ coarse, T* fine, int n)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * n) return;
    int i = e / 3, d = e - 3 * i;
    T s = 0;
    for (int l = 0; l < 8; ++l) s += pw[8 * (int64_t)i + l] * coarse[3 * (int64_t)pcol[8 * (int64_t)i + l] + d];
    fine[e] = s;
}
template <class T>
void Ctx<T>::restrict_dev(int level, const T* fine, T* coarse)
{
    Level<T>& C = *levels[level + 1];
    HOT_LAUNCH(this, "restrict", k_restrict<T>, div_up(3 * (size_t)C.n, 256), 256, 0, C.child.p, fine, coarse, C.n);
}
template <class T>
void Ctx<T>::prolong_dev(int level, const T* coarse, T* fine)
{
    Level<T>& F = *levels[level];
    HOT_LAUNCH(this, "prolong", k_prolong<T>, div_up(3 * (size_t)F.n, 256), 256, 0, F.pcol.p, F.pw.p, coarse, fine, F.n);
}

// ------------------------------------------------------------------------------------------------ smoothers
// mr_i = Dinv_i r_i (scale_diagonal_{entry,block}_inverse, MultigridPreconditioner.h:143-154)
template <class T>
__global__ void k_scale(const T* __restrict__ D, const T* __restrict__ r, T* mr, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T* d = D + 9 * (int64_t)i;
    T a = r[3 * (int64_t)i], b = r[3 * (int64_t)i + 1], c = r[3 * (int64_t)i + 2];
    mr[3 * (int64_t)i] = d[0] * a + d[3] * b + d[6] * c;
    mr[3 * (int64_t)i + 1] = d[1] * a + d[4] * b + d[7] * c;
    mr[3 * (int64_t)i + 2] = d[2] * a + d[5] * b + d[8] * c;
}

template <class T>
void Ctx<T>::scale_dev(Level<T>& L, const T* in, T* out)
{
    HOT_LAUNCH(this, "diag_scale", k_scale<T>, div_up(L.n, 256), 256, 0, L.diagInv.p, in, out, L.n);
}

// One colour of one half-sweep of symmetric block GS.  FWD: h_i = Dinv (rhs_i - sum_{j<i} A_ij h_j), also writes
// hD_i = D_i h_i ; BWD: du_i = Dinv (rhs_i - sum_{j>i} A_ij du_j).  "<" is the packed (colour, block, index) key.
template <class T, bool FWD>
__global__ __launch_bounds__(64) void k_gs_color(const int32_t* __restrict__ col, const T* __restrict__ val, const uint32_t* __restrict__ ckey, const int32_t* __restrict__ gs_order,
    const int32_t* __restrict__ block_start, const T* __restrict__ diagVal, const T* __restrict__ diagBlockInv, const T* __restrict__ rhs, T* x, T* hD, int block0, int nblk)
{
    __shared__ T xl[64][3];
    const int lane = threadIdx.x;
    const int b = block0 + blockIdx.x;
    if (blockIdx.x >= nblk) return;
    const int start = block_start[b], cnt = block_start[b + 1] - start;
    for (int s = 0; s < cnt; ++s) {
        const int ii = FWD ? s : cnt - 1 - s;
        const int i = gs_order[start + ii];
        const uint32_t keyi = ckey[i];
        const int32_t* c = col + (int64_t)i * 125;
        const T* v = val + (int64_t)i * 1125;
        T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int k = lane + 64 * r;
            if (k < 125) {
                int j = c[k];
                uint32_t keyj = ckey[j];
                bool take = FWD ? (keyj < keyi) : (keyj > keyi);
                if (take) {
                    T x0, x1, x2;
                    if ((keyj >> 7) == (keyi >> 7)) {
                        int lj = (int)(keyj & 127u) - 1;
                        x0 = xl[lj][0], x1 = xl[lj][1], x2 = xl[lj][2];
                    }
                    else {
                        x0 = x[3 * (int64_t)j], x1 = x[3 * (int64_t)j + 1], x2 = x[3 * (int64_t)j + 2];
                    }
                    const T* bb = v + k * 9;
                    s0 += bb[0] * x0 + bb[3] * x1 + bb[6] * x2;
                    s1 += bb[1] * x0 + bb[4] * x1 + bb[7] * x2;
                    s2 += bb[2] * x0 + bb[5] * x1 + bb[8] * x2;
                }
            }
        }
        s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
        T r0 = rhs[3 * (int64_t)i] - s0, r1 = rhs[3 * (int64_t)i + 1] - s1, r2 = rhs[3 * (int64_t)i + 2] - s2;
        const T* di = diagBlockInv + 9 * (int64_t)i;
        T h0 = di[0] * r0 + di[3] * r1 + di[6] * r2, h1 = di[1] * r0 + di[4] * r1 + di[7] * r2, h2 = di[2] * r0 + di[5] * r1 + di[8] * r2;
        if (lane == 0) {
            xl[ii][0] = h0, xl[ii][1] = h1, xl[ii][2] = h2;
            x[3 * (int64_t)i] = h0, x[3 * (int64_t)i + 1] = h1, x[3 * (int64_t)i + 2] = h2;
            if (FWD) {
                const T* d = diagVal + 9 * (int64_t)i;
                hD[3 * (int64_t)i] = d[0] * h0 + d[3] * h1 + d[6] * h2;
                hD[3 * (int64_t)i + 1] = d[1] * h0 + d[4] * h1 + d[7] * h2;
                hD[3 * (int64_t)i + 2] = d[2] * h0 + d[5] * h1 + d[8] * h2;
            }
        }
        __syncthreads(); // single-wave workgroup: orders the LDS write before the next node's reads
    }
}

// Two-phase block GS (the production path; k_gs_color above is the simple reference kernel kept for A/B checks).
// One 512-thread workgroup per 4^3-node block of the current colour:
//   phase A (8 waves, bandwidth-bound): every wave streams whole matrix rows of the block (lane = stencil slot).
//           Couplings to nodes OUTSIDE the block that precede the row in the sweep order are folded into
//           s_i = rhs_i - sum A_ij x_j ; couplings INSIDE the block that precede it are copied into an LDS
//           triangular array laid out by (column, row) so that phase B reads it conflict-free.
//   phase B (1 wave, latency-bound but LDS/register only): right-looking block substitution, lane = row:
//           step c: lane c finalises h_c = Dinv_c s_c, broadcasts it, every later row subtracts L[row][c] h_c.
// The node order inside a block, the colour order and the predicate are exactly those of k_gs_color, i.e. the
// reference's gs_smooth (MultigridPreconditioner.h:266-318); only the association order of the row sums differs.
template <class T>
struct GsLds {
    static constexpr int TRI = 2017; // 64*63/2 ordered pairs + one always-zero entry (index 2016) for masked lanes
    static constexpr size_t bytes = (size_t)9 * TRI * sizeof(T) + 64 * 3 * sizeof(T) + 64 * sizeof(int32_t);
};
__device__ __forceinline__ int gs_tri_fwd(int row, int colm) { return 63 * colm - (colm * (colm - 1)) / 2 + (row - colm - 1); } // row > colm
__device__ __forceinline__ int gs_tri_bwd(int row, int colm) { return (colm * (colm - 1)) / 2 + row; } // row < colm

template <class T, bool FWD>
__global__ __launch_bounds__(1024) void k_gs_block(const int32_t* __restrict__ col, const T* __restrict__ val, const uint32_t* __restrict__ ckey, const int32_t* __restrict__ gs_order,
    const int32_t* __restrict__ block_start, const T* __restrict__ diagVal, const T* __restrict__ diagBlockInv, const T* __restrict__ rhs, T* x, T* hD, int block0, int dbg,
    const int32_t* __restrict__ rowcnt, const int32_t* __restrict__ meta)
{
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    constexpr int TRI = GsLds<T>::TRI;
    T* tri = (T*)gs_smem; // [9][TRI]
    T* sv = tri + 9 * TRI; // [64][3]
    int32_t* nodes = (int32_t*)(sv + 192);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = block0 + blockIdx.x;
    const int start = block_start[b], cnt = block_start[b + 1] - start;
    const int nthreads = blockDim.x, nwaves = blockDim.x >> 6;
    if (!(dbg & 2)) for (int e = tid; e < 9 * TRI; e += nthreads) tri[e] = (T)0;
    if (tid < 64) nodes[tid] = tid < cnt ? gs_order[start + tid] : -1;
    __syncthreads();
    // ---------------- phase A, fast path for regrouped rows: every load of the (<= 4) rows of this wave is issued
    // before any is used, and the precomputed slot descriptor (meta >= 0: column j outside the block, meta <= -2:
    // local index -2-meta inside the block) removes the col -> ckey -> x dependent-load chain
    if (rowcnt != nullptr && meta != nullptr) {
        constexpr int RQ = 4;
        for (int q0 = 0; q0 * nwaves < cnt; q0 += RQ) {
            T bv[RQ][9];
            int mm[RQ], kb[RQ], ke[RQ];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const int ii = w + nwaves * (q0 + q);
                mm[q] = -1, kb[q] = 0, ke[q] = 0;
#pragma unroll
                for (int e = 0; e < 9; ++e) bv[q][e] = (T)0;
                if (ii < cnt) {
                    const int i = nodes[ii];
                    const int nl = rowcnt[2 * i], nu = rowcnt[2 * i + 1];
                    kb[q] = FWD ? 0 : nl + 1, ke[q] = FWD ? nl : nl + 1 + nu;
                    const int k = kb[q] + lane;
                    if (k < ke[q]) {
                        mm[q] = meta[(int64_t)i * 125 + k];
                        const T* bb = val + ((int64_t)i * 125 + k) * 9;
#pragma unroll
                        for (int e = 0; e < 9; ++e) bv[q][e] = bb[e];
                    }
                }
            }
            T xs[RQ][3];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const int m = mm[q];
                xs[q][0] = xs[q][1] = xs[q][2] = (T)0;
                if (m >= 0) xs[q][0] = x[3 * (int64_t)m], xs[q][1] = x[3 * (int64_t)m + 1], xs[q][2] = x[3 * (int64_t)m + 2];
            }
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const int ii = w + nwaves * (q0 + q);
                if (ii >= cnt) continue; // wave-uniform
                const int i = nodes[ii];
                T s0 = bv[q][0] * xs[q][0] + bv[q][3] * xs[q][1] + bv[q][6] * xs[q][2];
                T s1 = bv[q][1] * xs[q][0] + bv[q][4] * xs[q][1] + bv[q][7] * xs[q][2];
                T s2 = bv[q][2] * xs[q][0] + bv[q][5] * xs[q][1] + bv[q][8] * xs[q][2];
                if (mm[q] <= -2) {
                    const int lj = -2 - mm[q];
                    const int idx = FWD ? gs_tri_fwd(ii, lj) : gs_tri_bwd(ii, lj);
#pragma unroll
                    for (int e = 0; e < 9; ++e) tri[e * TRI + idx] = bv[q][e];
                }
                // half rows longer than one wave (cannot happen for interior 4^3 blocks): plain strided tail
                for (int k = kb[q] + 64 + lane; k < ke[q]; k += 64) {
                    const int m = meta[(int64_t)i * 125 + k];
                    const T* bb = val + ((int64_t)i * 125 + k) * 9;
                    if (m >= 0) {
                        T x0 = x[3 * (int64_t)m], x1 = x[3 * (int64_t)m + 1], x2 = x[3 * (int64_t)m + 2];
                        s0 += bb[0] * x0 + bb[3] * x1 + bb[6] * x2;
                        s1 += bb[1] * x0 + bb[4] * x1 + bb[7] * x2;
                        s2 += bb[2] * x0 + bb[5] * x1 + bb[8] * x2;
                    }
                    else if (m <= -2) {
                        const int lj = -2 - m;
                        const int idx = FWD ? gs_tri_fwd(ii, lj) : gs_tri_bwd(ii, lj);
#pragma unroll
                        for (int e = 0; e < 9; ++e) tri[e * TRI + idx] = bb[e];
                    }
                }
                s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
                if (lane == 0) sv[ii * 3] = rhs[3 * (int64_t)i] - s0, sv[ii * 3 + 1] = rhs[3 * (int64_t)i + 1] - s1, sv[ii * 3 + 2] = rhs[3 * (int64_t)i + 2] - s2;
            }
        }
    }
    else
    // ---------------- phase A, generic path (rows in any slot order, predicate on the packed key)
    for (int ii = w; ii < cnt; ii += nwaves) {
        const int i = nodes[ii];
        const uint32_t keyi = ckey[i];
        const int32_t* c = col + (int64_t)i * 125;
        const T* v = val + (int64_t)i * 1125;
        T s0 = 0, s1 = 0, s2 = 0;
        // issue every load of the row up front (both slot rounds): the 3x3 blocks do not depend on the
        // col -> ckey -> x chain, so the whole row (9 KB per wave) is in flight at once
        T bv[2][9];
        int jj[2];
        // rows regrouped by k_gs_split_rows: the forward sweep needs slots [0, nl), the backward sweep
        // [nl + 1, nl + 1 + nu); without the split every slot is visited and filtered by the key predicate
        int kbeg = 0, kend = 125;
        if (rowcnt) {
            int nl = rowcnt[2 * i], nu = rowcnt[2 * i + 1];
            kbeg = FWD ? 0 : nl + 1, kend = FWD ? nl : nl + 1 + nu;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int k = kbeg + lane + 64 * r;
            jj[r] = k < kend ? c[k] : -1;
#pragma unroll
            for (int e = 0; e < 9; ++e) bv[r][e] = k < kend ? v[k * 9 + e] : (T)0;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int j = jj[r];
            if (j >= 0) {
                uint32_t keyj = ckey[j];
                bool take = FWD ? (keyj < keyi) : (keyj > keyi);
                if (take) {
                    if ((keyj >> 7) == (keyi >> 7)) {
                        int lj = (int)(keyj & 127u) - 1;
                        int idx = FWD ? gs_tri_fwd(ii, lj) : gs_tri_bwd(ii, lj);
                        // padded slots alias column 0/1 with an all-zero block (SquareMatrix.h:563-566): they must not
                        // overwrite the real (row, column) entry, so only non-zero blocks are stored
                        bool nz = false;
#pragma unroll
                        for (int e = 0; e < 9; ++e) nz = nz || bv[r][e] != (T)0;
                        if (nz) {
#pragma unroll
                            for (int e = 0; e < 9; ++e) tri[e * TRI + idx] = bv[r][e];
                        }
                    }
                    else {
                        T x0 = x[3 * (int64_t)j], x1 = x[3 * (int64_t)j + 1], x2 = x[3 * (int64_t)j + 2];
                        s0 += bv[r][0] * x0 + bv[r][3] * x1 + bv[r][6] * x2;
                        s1 += bv[r][1] * x0 + bv[r][4] * x1 + bv[r][7] * x2;
                        s2 += bv[r][2] * x0 + bv[r][5] * x1 + bv[r][8] * x2;
                    }
                }
            }
        }
        s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
        if (lane == 0) {
            sv[ii * 3] = rhs[3 * (int64_t)i] - s0, sv[ii * 3 + 1] = rhs[3 * (int64_t)i + 1] - s1, sv[ii * 3 + 2] = rhs[3 * (int64_t)i + 2] - s2;
        }
    }
    __syncthreads();
    if (w != 0 || (dbg & 1)) return;
    // ---------------- phase B: lane = row
    const int me = lane;
    const int i = me < cnt ? nodes[me] : -1;
    T d[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) d[e] = i >= 0 ? diagBlockInv[9 * (int64_t)i + e] : (T)0;
    T a0 = me < cnt ? sv[me * 3] : (T)0, a1 = me < cnt ? sv[me * 3 + 1] : (T)0, a2 = me < cnt ? sv[me * 3 + 2] : (T)0;
    T h0 = 0, h1 = 0, h2 = 0;
    // column `cidx` of the in-block triangle for this lane's row (zero where the row does not follow the column);
    // the next column is fetched from LDS while the current step's dependent arithmetic runs
    auto load_col = [&](int cidx, T (&L)[9]) {
        bool act = FWD ? (me > cidx && me < cnt) : (me < cidx);
        int idx = act ? (FWD ? gs_tri_fwd(me, cidx) : gs_tri_bwd(me, cidx)) : TRI - 1; // masked lanes read the zero entry
#pragma unroll
        for (int e = 0; e < 9; ++e) L[e] = tri[e * TRI + idx];
    };
    T Lc[9], Ln[9];
    if (cnt > 0) load_col(FWD ? 0 : cnt - 1, Lc);
    for (int s = 0; s < cnt; ++s) {
        const int cidx = FWD ? s : cnt - 1 - s;
        if (s + 1 < cnt) load_col(FWD ? s + 1 : cnt - 2 - s, Ln);
        // candidate solution of every row from its current partial sum; only lane cidx's is final
        T c0 = d[0] * a0 + d[3] * a1 + d[6] * a2, c1 = d[1] * a0 + d[4] * a1 + d[7] * a2, c2 = d[2] * a0 + d[5] * a1 + d[8] * a2;
        if (me == cidx) h0 = c0, h1 = c1, h2 = c2;
        T b0 = lane_bcast(c0, cidx), b1 = lane_bcast(c1, cidx), b2 = lane_bcast(c2, cidx); // v_readlane: cidx is wave-uniform
        a0 -= Lc[0] * b0 + Lc[3] * b1 + Lc[6] * b2;
        a1 -= Lc[1] * b0 + Lc[4] * b1 + Lc[7] * b2;
        a2 -= Lc[2] * b0 + Lc[5] * b1 + Lc[8] * b2;
#pragma unroll
        for (int e = 0; e < 9; ++e) Lc[e] = Ln[e];
    }
    if (i >= 0) {
        x[3 * (int64_t)i] = h0, x[3 * (int64_t)i + 1] = h1, x[3 * (int64_t)i + 2] = h2;
        if (FWD) {
            const T* dd = diagVal + 9 * (int64_t)i;
            hD[3 * (int64_t)i] = dd[0] * h0 + dd[3] * h1 + dd[6] * h2;
            hD[3 * (int64_t)i + 1] = dd[1] * h0 + dd[4] * h1 + dd[7] * h2;
            hD[3 * (int64_t)i + 2] = dd[2] * h0 + dd[5] * h1 + dd[8] * h2;
        }
    }
}

// r_i = sum over the nl slots preceding row i of A_ik (h - du)_k   (rows regrouped by k_gs_split_rows)
template <class T>
__global__ __launch_bounds__(256) void k_gs_residual(const int32_t* __restrict__ col, const T* __restrict__ val, const int32_t* __restrict__ rowcnt, const T* __restrict__ h,
    const T* __restrict__ du, T* __restrict__ r, int n)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int nl = rowcnt[2 * row];
    const int32_t* c = col + (int64_t)row * 125;
    const T* v = val + (int64_t)row * 1125;
    T s0 = 0, s1 = 0, s2 = 0;
    for (int k = lane; k < nl; k += 64) {
        int j = c[k];
        const T* b = v + k * 9;
        T x0 = h[3 * (int64_t)j] - du[3 * (int64_t)j], x1 = h[3 * (int64_t)j + 1] - du[3 * (int64_t)j + 1], x2 = h[3 * (int64_t)j + 2] - du[3 * (int64_t)j + 2];
        s0 += b[0] * x0 + b[3] * x1 + b[6] * x2;
        s1 += b[1] * x0 + b[4] * x1 + b[7] * x2;
        s2 += b[2] * x0 + b[5] * x1 + b[8] * x2;
    }
    s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
    if (lane == 0) r[3 * (int64_t)row] = s0, r[3 * (int64_t)row + 1] = s1, r[3 * (int64_t)row + 2] = s2;
}

template <class T>
__global__ void k_cg_scalars(double* s, int what)
{
    // s[0]=zTrk  s[1]=dAu.du  s[2]=omega  s[3]=-omega  s[4]=zTrk_new  s[5]=beta
    if (what == 0) {
        s[2] = s[0] / s[1];
        s[3] = -s[2];
    }
    else {
        s[5] = s[4] / s[0];
        s[0] = s[4];
    }
}

template <class T>
void Ctx<T>::smooth_dev(int level, int kind, int iterations, T tolerance, T* u, T* r, T* du, T* dAu)
{
    Level<T>& L = *levels[level];
    size_t n3 = 3 * (size_t)L.n;
    auto Aproject = [&](T* v) {
        if (level == 0 && !cfg.systemBCProject) project_dev(v);
    };
    auto scaler = [&](const T* in, T* out) { scale_dev(L, in, out); };
    if (kind == 0) {
        for (; iterations--;) {
            scaler(r, du);
            scal(n3, (T)cfg.topomega, du);
            axpy(n3, (T)1, du, u);
            spmv_dev(L, du, dAu);
            Aproject(dAu);
            axpy(n3, (T)-1, dAu, r);
        }
    }
    else if (kind == 1) {
        for (; iterations--;) {
            double rr = dot_host(n3, r, r);
            if (std::sqrt(rr) < (double)tolerance) break;
            scaler(r, du);
            spmv_dev(L, du, dAu);
            Aproject(dAu);
            double a = dot_host(n3, du, r), b = dot_host(n3, du, dAu);
            T omega = (T)(a / b);
            axpy(n3, omega, du, u);
            axpy(n3, -omega, dAu, r);
        }
    }
    else if (kind == 2) {
        T* z = L.tmp.p;
        double* s = dscal.p + 40;
        scaler(L.initialResidual.p, z);
        double zTrk0 = dot_host(n3, z, L.initialResidual.p);
        scaler(r, z);
        copy(n3, z, du);
        double zTrk = dot_host(n3, z, r);
        double tol = (double)(T)(zTrk0 * 0.25); // cgratio = 0.5 hard-wired (:203-209)
        HOT_HIP(hipMemcpyAsync(s, &zTrk, sizeof(double), hipMemcpyHostToDevice, stream));
        int cnt = 0;
        for (; iterations--;) {
            if (zTrk < tol) break;
            spmv_dev(L, du, dAu);
            Aproject(dAu);
            dot_to(n3, dAu, du, s + 1);
            HOT_LAUNCH(this, "cg_scalars", k_cg_scalars<T>, 1, 1, 0, s, 0);
            axpy_dev(n3, s + 2, 1.0, du, u);
            axpy_dev(n3, s + 3, 1.0, dAu, r);
            scaler(r, z);
            dot_to(n3, z, r, s + 4);
            HOT_LAUNCH(this, "cg_scalars", k_cg_scalars<T>, 1, 1, 0, s, 1);
            HOT_LAUNCH(this, "xpay", k_xpay_dev<T>, div_up(n3, 256), 256, 0, n3, s + 5, z, du);
            HOT_HIP(hipMemcpyAsync(hscal + 40, s, sizeof(double), hipMemcpyDeviceToHost, stream));
            sync();
            zTrk = hscal[40];
            ++cnt;
        }
        stats.linear_iterations += cnt;
    }
    else if (kind == 5) {
        HOT_CHECK(L.nblocks > 0, HOT_ERR_INVALID, "GS smoother requested but the level was built without colouring");
        T* hdu = L.tmp.p;
        static const bool simple_gs = getenv("HOT_SIMPLE_GS") != nullptr; // A/B switch: one-wave-per-block reference kernel
        static const int gs_threads = getenv("HOT_GS_THREADS") ? atoi(getenv("HOT_GS_THREADS")) : 1024;
        static const bool no_meta = getenv("HOT_GS_NO_META") != nullptr; // A/B switch: generic phase A on split rows
        static const bool no_lres = getenv("HOT_GS_FULL_RESIDUAL") != nullptr; // A/B switch: r -= A du by a full SpMV
        static const int gs_dbg = getenv("HOT_GS_DBG") ? atoi(getenv("HOT_GS_DBG")) : 0; // timing experiments only (wrong results)
        static bool attr_set = false;
        if (!attr_set) {
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_block<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GsLds<T>::bytes));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_block<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GsLds<T>::bytes));
            attr_set = true;
        }
        iterations = ((iterations + 1) >> 1);
        for (; iterations--;) {
            zero(n3, hdu);
            for (int c = 0; c < 8; ++c) {
                int b0 = L.color_block_begin[c], nb = L.color_block_begin[c + 1] - b0;
                if (nb > 0) {
                    if (simple_gs)
                        HOT_LAUNCH(this, lname("gs_forward", L.id).c_str(), (k_gs_color<T, true>), nb, 64, 0, L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, L.diagVal.p, L.diagBlockInv.p, r, hdu, dAu, b0, nb);
                    else
                        HOT_LAUNCH(this, lname("gs_forward", L.id).c_str(), (k_gs_block<T, true>), nb, gs_threads, GsLds<T>::bytes, L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, L.diagVal.p, L.diagBlockInv.p, r, hdu, dAu, b0, gs_dbg, L.split ? L.rowcnt.p : (const int32_t*)nullptr, (L.split && !no_meta) ? L.gsmeta.p : (const int32_t*)nullptr);
                }
            }
            // dAu now holds D h ; du = backward solve
            zero(n3, du);
            for (int c = 7; c >= 0; --c) {
                int b0 = L.color_block_begin[c], nb = L.color_block_begin[c + 1] - b0;
                if (nb > 0) {
                    if (simple_gs)
                        HOT_LAUNCH(this, lname("gs_backward", L.id).c_str(), (k_gs_color<T, false>), nb, 64, 0, L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, L.diagVal.p, L.diagBlockInv.p, dAu, du, (T*)nullptr, b0, nb);
                    else
                        HOT_LAUNCH(this, lname("gs_backward", L.id).c_str(), (k_gs_block<T, false>), nb, gs_threads, GsLds<T>::bytes, L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, L.diagVal.p, L.diagBlockInv.p, dAu, du, (T*)nullptr, b0, gs_dbg, L.split ? L.rowcnt.p : (const int32_t*)nullptr, (L.split && !no_meta) ? L.gsmeta.p : (const int32_t*)nullptr);
                }
            }
            axpy(n3, (T)1, du, u);
            if (L.split && !simple_gs && !(level == 0 && !cfg.systemBCProject) && !no_lres) {
                // r - A du = L (h - du): with (D+L) h = r and (D+U) du = D h the full product A du collapses to the
                // strictly-preceding half of the matrix applied to (h - du) (same value, half the bytes of an SpMV)
                HOT_LAUNCH(this, lname("gs_residual", L.id).c_str(), k_gs_residual<T>, div_up(L.n, 4), 256, 0, L.col.p, L.val.p, L.rowcnt.p, hdu, du, r, L.n);
            }
            else {
                spmv_dev(L, du, dAu);
                Aproject(dAu);
                axpy(n3, (T)-1, dAu, r);
            }
        }
    }
    else
        HOT_CHECK(false, HOT_ERR_INVALID, "unsupported smoother kind");
}

template <class T>
void Ctx<T>::vcycle_dev(const T* in, T* out)
{
    int levelCnt = (int)levels.size();
    int times = cfg.times, levelscale = cfg.levelscale;
    int splitLevel;
    auto downIter = [&](int level) { return times + level * levelscale; };
    auto upIter = [&](int level) { return cfg.topDownMGS ? 0 : times + level * levelscale; };
    auto topIter = [&](int level) {
        if (cfg.topDownMGS) return 10000;
        if (cfg.levelCnt == 1) return times + level * levelscale;
        if (!(cfg.coarseSolver == 2 || cfg.coarseSolver == 6)) return (times + level * levelscale) * 3;
        return 10000;
    };
    splitLevel = cfg.topDownMGS ? 1 : cfg.levelCnt - 1;
    T tolTop = (T)(cfg.cneps * cfg.cneps);
    auto run = [&](bool regular, int level, T* sol, int its) {
        Level<T>& L = *levels[level];
        smooth_dev(level, regular ? cfg.smoother : cfg.coarseSolver, its, regular ? (T)0 : tolTop, sol, L.residual.p, L.du.p, L.dAu.p);
    };
    stats.vcycles++;
    Level<T>& L0 = *levels[0];
    size_t n0 = 3 * (size_t)L0.n;
    copy(n0, in, L0.residual.p); // dRhs == 0 (ImplicitSolver.h:483-484,579), correctResidualProjection is the identity
    zero(n0, out);
    if (levelCnt > 1)
        restrict_dev(0, L0.residual.p, levels[1]->initialResidual.p);
    else
        copy(n0, L0.residual.p, L0.initialResidual.p);
    for (int l = 1; l < levelCnt - 1; ++l) restrict_dev(l, levels[l]->initialResidual.p, levels[l + 1]->initialResidual.p);
    int level;
    for (level = 0; level < levelCnt - 1; ++level) {
        T* sol = level == 0 ? out : levels[level]->sol.p;
        run(level < splitLevel, level, sol, level < splitLevel ? upIter(level) : topIter(level));
        restrict_dev(level, levels[level]->residual.p, levels[level + 1]->residual.p);
        zero(3 * (size_t)levels[level + 1]->n, levels[level + 1]->sol.p);
    }
    run(false, level, level == 0 ? out : levels[level]->sol.p, topIter(level));
    for (--level; level >= 0; --level) {
        Level<T>& L = *levels[level];
        T* sol = level == 0 ? out : L.sol.p;
        size_t n3 = 3 * (size_t)L.n;
        prolong_dev(level, levels[level + 1]->sol.p, L.du.p);
        axpy(n3, (T)1, L.du.p, sol);
        spmv_dev(L, L.du.p, L.dAu.p);
        axpy(n3, (T)-1, L.dAu.p, L.residual.p);
        run(level < splitLevel, level, sol, level < splitLevel ? downIter(level) : topIter(level));
    }
}

template <class T>
void Ctx<T>::precondition_dev(const T* in, T* out)
{
    HOT_CHECK(!levels.empty() && levels[0]->built, HOT_ERR_INVALID, "preconditioner used before hot_build_mg");
    vcycle_dev(in, out);
}

// ------------------------------------------------------------------------------------------------ C ABI wrappers
template <class T>
void Ctx<T>::spmv(int32_t level, const void* x, void* y)
{
    need(level >= 0 && level < (int)levels.size(), "level out of range");
    Level<T>& L = *levels[level];
    size_t n3 = 3 * (size_t)L.n;
    DBuf<T> a, b;
    a.reserve(n3), b.reserve(n3);
    HOT_HIP(hipMemcpyAsync(a.p, x, n3 * sizeof(T), hipMemcpyDefault, stream));
    spmv_dev(L, a.p, b.p);
    download(y, b.p, n3);
    sync();
}
template <class T>
void Ctx<T>::restrict_(int32_t level, const void* fine, void* coarse)
{
    need(level >= 0 && level + 1 < (int)levels.size(), "level out of range");
    size_t nf = 3 * (size_t)levels[level]->n, nc = 3 * (size_t)levels[level + 1]->n;
    DBuf<T> a, b;
    a.reserve(nf), b.reserve(nc);
    HOT_HIP(hipMemcpyAsync(a.p, fine, nf * sizeof(T), hipMemcpyDefault, stream));
    restrict_dev(level, a.p, b.p);
    download(coarse, b.p, nc);
    sync();
}
template <class T>
void Ctx<T>::prolong(int32_t level, const void* coarse, void* fine)
{
    need(level >= 0 && level + 1 < (int)levels.size(), "level out of range");
    size_t nf = 3 * (size_t)levels[level]->n, nc = 3 * (size_t)levels[level + 1]->n;
    DBuf<T> a, b;
    a.reserve(nc), b.reserve(nf);
    HOT_HIP(hipMemcpyAsync(a.p, coarse, nc * sizeof(T), hipMemcpyDefault, stream));
    prolong_dev(level, a.p, b.p);
    download(fine, b.p, nf);
    sync();
}
template <class T>
void Ctx<T>::smooth(int32_t level, int32_t kind, int32_t iterations, double tol, void* u, void* r, const void* r0)
{
    need(level >= 0 && level < (int)levels.size() && levels[level]->built, "hot_smooth: level not built (hot_build_mg)");
    Level<T>& L = *levels[level];
    size_t n3 = 3 * (size_t)L.n;
    DBuf<T> du_, dr_;
    du_.reserve(n3), dr_.reserve(n3);
    HOT_HIP(hipMemcpyAsync(du_.p, u, n3 * sizeof(T), hipMemcpyDefault, stream));
    HOT_HIP(hipMemcpyAsync(dr_.p, r, n3 * sizeof(T), hipMemcpyDefault, stream));
    HOT_HIP(hipMemcpyAsync(L.initialResidual.p, r0 ? r0 : r, n3 * sizeof(T), hipMemcpyDefault, stream));
    smooth_dev(level, kind, iterations, (T)tol, du_.p, dr_.p, L.du.p, L.dAu.p);
    download(u, du_.p, n3);
    download(r, dr_.p, n3);
    sync();
}
template <class T>
void Ctx<T>::vcycle(const void* in, void* out)
{
    need(!levels.empty() && levels[0]->built, "hot_vcycle before hot_build_mg");
    size_t n3 = 3 * (size_t)Nn;
    HOT_HIP(hipMemcpyAsync(work0.p, in, n3 * sizeof(T), hipMemcpyDefault, stream));
    vcycle_dev(work0.p, work1.p);
    download(out, work1.p, n3);
    sync();
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
