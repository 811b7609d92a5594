// libhotmi355x — assembled Hessian  H = M + dt^2 sum_p V_p sum_ij (dP/dF : (Fn^T grad w_i)(Fn^T grad w_j))
// in the reference's padded ELL-125 layout, BC projection of the system, block diagonal and its inverse.
//
//   k_fill_cols     entryCol of level 0 straight from the grid: slot linearOffset(node_i - node_j) holds dof_j
//                   (reference Projects/multigrid/ImplicitSolver.h:465-468,548-551); absent neighbours get the reference's
//                   padding column (0, or 1 for row 0; :556-559).
//   k_hessian       buildMatrix<true>'s particle loop (ImplicitSolver.h:498-552) with
//                   FBasedMpmForceHelper::runLambdaWithDifferential (Lib/MPM/Force/FBasedMpmForceHelper.h:63-121) and
//                   CorotatedIsotropic::firstPiolaDerivative (CorotatedIsotropic.h:174-230).
//                   MI355X design: one workgroup per particle group; per particle the 9x9 dP/dF is built once in LDS
//                   by rotating the 21 non-zero SVD-frame couplings (A: 9, B01/B12/B20: 4 each), then T_i = dPdF . g_i
//                   (27 x 27 values) is staged in LDS and each thread owns up to two of the 378 unordered node pairs,
//                   accumulating its 3x3 block IN REGISTERS across all consecutive particles that share a base cell
//                   (they share the 27 nodes); blocks are flushed with global atomics once per cell — ppc times fewer
//                   atomics than the per-particle scatter, and no colour passes.  The mirrored block is the exact
//                   transpose (as in the reference, :545-551), so the assembled matrix is exactly symmetric.
//   k_bc_project    the per-row BC projection (:554-593); k_diag buildDiagonal (Projects/multigrid/SquareMatrix.h:301-324).
#include "hot_impl.h"
#include "hot_constitutive.h"

namespace hot {

// dof of the grid node at integer coords (or -1): block hash -> tile -> idx
template <class T>
__device__ __forceinline__ int32_t node_dof(const HashMap& bm, const int32_t* __restrict__ gIdx, int x, int y, int z)
{
    using G = Geo<T>;
    if ((x | y | z) < 0) return -1;
    uint64_t off = G::linear_offset(x, y, z);
    int32_t b = hash_find_id(bm, off >> 12);
    if (b < 0) return -1;
    return gIdx[(int64_t)b * G::EPB + (int)((off & 0xfff) >> G::data_bits)];
}

template <class T>
__global__ void k_fill_cols(HashMap bm, const int32_t* __restrict__ gIdx, const int32_t* __restrict__ id2coord, const T* __restrict__ mass, int32_t* col, T* val, int nn, int init_val)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)nn * 125) return;
    int n = (int)(e / 125), k = (int)(e - (int64_t)n * 125);
    int dx = k / 25 - 2, dy = (k / 5) % 5 - 2, dz = k % 5 - 2;
    int32_t j = node_dof<T>(bm, gIdx, id2coord[3 * n] - dx, id2coord[3 * n + 1] - dy, id2coord[3 * n + 2] - dz);
    col[e] = j >= 0 ? j : (n > 0 ? 0 : 1);
    if (!init_val) return;
    T m = (k == 62) ? mass[n] : (T)0;
    T* v = val + e * 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) v[c] = (c % 4 == 0) ? m : (T)0;
}

#ifdef HOT_AB_KERNELS
// pair index q in [0,378) -> (i <= j) over 27 nodes
__device__ __forceinline__ void pair_ij(int q, int& i, int& j)
{
    // row i has 27 - i entries; solve by scanning (27 steps max, done once per thread)
    int base = 0;
    for (i = 0; i < 27; ++i) {
        int len = 27 - i;
        if (q < base + len) break;
        base += len;
    }
    j = i + (q - base);
}

template <class T>
__global__ __launch_bounds__(256) void k_hessian(const T* __restrict__ X, const T* __restrict__ Fn, const T* __restrict__ Ft, const T* __restrict__ Vol, const T* __restrict__ Mu,
    const T* __restrict__ Lam, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin, const int32_t* __restrict__ group_nb,
    const int32_t* __restrict__ gIdx, T* val, T dx, T one_over_dx, T dt, int project)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    constexpr int CH = 64; // particles per SVD chunk
    __shared__ int32_t tidx[TILE];
    __shared__ int32_t nb8[8];
    __shared__ T hb[CH][34]; // per particle: U(9) V(9) A(6: 00 11 22 01 02 12) B01(3) B12(3) B20(3) + vol*dt^2
    __shared__ T dP[81]; // dP/dF of the current particle, [(a + 3 v) + 9 (b + 3 q)]
    __shared__ T gvec[27][3]; // Fn^T grad w_i
    __shared__ T Ti[27][27]; // T_i[a + 3*(b + 3*q)]
    __shared__ int32_t cell[3]; // base node of the current particle
    __shared__ int32_t rowdof[27];
    const int g = blockIdx.x, tid = threadIdx.x;
    if (tid < 8) nb8[tid] = group_nb[g * 8 + tid];
    __syncthreads();
    for (int t = tid; t < TILE; t += 256) {
        int tz = t % TZ, ty = (t / TZ) % TY, tx = t / (TZ * TY);
        int ox = tx >> G::xb, oy = ty >> G::yb, oz = tz >> G::zb;
        int elem = ((tx & (G::BX - 1)) << (G::yb + G::zb)) | ((ty & (G::BY - 1)) << G::zb) | (tz & (G::BZ - 1));
        tidx[t] = gIdx[(int64_t)nb8[ox * 4 + oy * 2 + oz] * G::EPB + elem];
    }
    const int first = group_first[g], last = group_first[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    // the (up to) two node pairs owned by this thread
    int pi[2], pj[2];
    pair_ij(tid, pi[0], pj[0]);
    bool has2 = tid + 256 < 378;
    pair_ij(has2 ? tid + 256 : 0, pi[1], pj[1]);
    T accm[2][9];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 9; ++c) accm[s][c] = (T)0;
    int cur[3] = { -(1 << 30), 0, 0 };
    bool have_cell = false;

    auto flush = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 1 && !has2) continue;
            int i = pi[s], j = pj[s];
            int di = rowdof[i], dj = rowdof[j];
            if (di >= 0 && dj >= 0) {
                // node offsets inside the 3x3x3 kernel
                int ix = i / 9, iy = (i / 3) % 3, iz = i % 3, jx = j / 9, jy = (j / 3) % 3, jz = j % 3;
                int sij = (ix - jx + 2) * 25 + (iy - jy + 2) * 5 + (iz - jz + 2);
                T* a = val + ((int64_t)di * 125 + sij) * 9;
#pragma unroll
                for (int c = 0; c < 9; ++c) atomic_add(a + c, accm[s][c]);
                if (i != j) {
                    int sji = (jx - ix + 2) * 25 + (jy - iy + 2) * 5 + (jz - iz + 2);
                    T* b = val + ((int64_t)dj * 125 + sji) * 9;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int r = 0; r < 3; ++r) atomic_add(b + (c * 3 + r), accm[s][r * 3 + c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) accm[s][c] = (T)0;
        }
    };

    for (int chunk = first; chunk < last; chunk += CH) {
        __syncthreads();
        // ---- per-particle SVD-frame blocks for this chunk
        if (tid < CH && chunk + tid < last) {
            int p = chunk + tid;
            Mat3<T> Fc;
#pragma unroll
            for (int c = 0; c < 9; ++c) Fc.a[c] = Ft[(int64_t)c * Np + p];
            HessBlocks<T> h;
            corotated_hessian(Fc, Mu[p], Lam[p], project != 0, h);
#pragma unroll
            for (int c = 0; c < 9; ++c) hb[tid][c] = h.U.a[c], hb[tid][9 + c] = h.V.a[c];
            hb[tid][18] = h.A(0, 0), hb[tid][19] = h.A(1, 1), hb[tid][20] = h.A(2, 2), hb[tid][21] = h.A(0, 1), hb[tid][22] = h.A(0, 2), hb[tid][23] = h.A(1, 2);
#pragma unroll
            for (int c = 0; c < 3; ++c) hb[tid][24 + c] = h.B01[c], hb[tid][27 + c] = h.B12[c], hb[tid][30 + c] = h.B20[c];
            hb[tid][33] = Vol[p] * dt * dt;
        }
        __syncthreads();
        const int cnt = min(CH, last - chunk);
        for (int l = 0; l < cnt; ++l) {
            const int p = chunk + l;
            // ---- stage: kernel gradients (lanes 0..26), base cell (lane 0), dPdF (lanes 64..144)
            if (tid < 27) {
                T xp[3] = { X[p], X[Np + p], X[2 * Np + p] };
                int base[3];
                T w[3][3], dw[3][3];
#pragma unroll
                for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xp[d], base[d], w[d], dw[d]);
                int i = tid / 9, j = (tid / 3) % 3, k = tid % 3;
                T g0 = one_over_dx * dw[0][i] * w[1][j] * w[2][k], g1 = w[0][i] * one_over_dx * dw[1][j] * w[2][k], g2 = w[0][i] * w[1][j] * one_over_dx * dw[2][k];
                // Fn^T g
#pragma unroll
                for (int c = 0; c < 3; ++c) gvec[tid][c] = Fn[(int64_t)(c * 3 + 0) * Np + p] * g0 + Fn[(int64_t)(c * 3 + 1) * Np + p] * g1 + Fn[(int64_t)(c * 3 + 2) * Np + p] * g2;
                if (tid == 0) cell[0] = base[0], cell[1] = base[1], cell[2] = base[2];
            }
            else if (tid >= 64 && tid < 64 + 81) {
                int e = tid - 64;
                int ij = e % 9, rs = e / 9;
                int jj = ij / 3, ii = ij - jj * 3, ss = rs / 3, rr = rs - ss * 3;
                const T* H = hb[l];
                auto U = [&](int r, int c) { return H[c * 3 + r]; };
                auto V = [&](int r, int c) { return H[9 + c * 3 + r]; };
                T A00 = H[18], A11 = H[19], A22 = H[20], A01 = H[21], A02 = H[22], A12 = H[23];
                T v = A00 * U(ii, 0) * V(jj, 0) * U(rr, 0) * V(ss, 0) + A01 * U(ii, 0) * V(jj, 0) * U(rr, 1) * V(ss, 1) + A02 * U(ii, 0) * V(jj, 0) * U(rr, 2) * V(ss, 2)
                    + A01 * U(ii, 1) * V(jj, 1) * U(rr, 0) * V(ss, 0) + A11 * U(ii, 1) * V(jj, 1) * U(rr, 1) * V(ss, 1) + A12 * U(ii, 1) * V(jj, 1) * U(rr, 2) * V(ss, 2)
                    + A02 * U(ii, 2) * V(jj, 2) * U(rr, 0) * V(ss, 0) + A12 * U(ii, 2) * V(jj, 2) * U(rr, 1) * V(ss, 1) + A22 * U(ii, 2) * V(jj, 2) * U(rr, 2) * V(ss, 2)
                    + H[24] * U(ii, 0) * V(jj, 1) * U(rr, 0) * V(ss, 1) + H[25] * U(ii, 0) * V(jj, 1) * U(rr, 1) * V(ss, 0) + H[25] * U(ii, 1) * V(jj, 0) * U(rr, 0) * V(ss, 1) + H[26] * U(ii, 1) * V(jj, 0) * U(rr, 1) * V(ss, 0)
                    + H[27] * U(ii, 1) * V(jj, 2) * U(rr, 1) * V(ss, 2) + H[28] * U(ii, 1) * V(jj, 2) * U(rr, 2) * V(ss, 1) + H[28] * U(ii, 2) * V(jj, 1) * U(rr, 1) * V(ss, 2) + H[29] * U(ii, 2) * V(jj, 1) * U(rr, 2) * V(ss, 1)
                    + H[32] * U(ii, 0) * V(jj, 2) * U(rr, 0) * V(ss, 2) + H[31] * U(ii, 0) * V(jj, 2) * U(rr, 2) * V(ss, 0) + H[31] * U(ii, 2) * V(jj, 0) * U(rr, 0) * V(ss, 2) + H[30] * U(ii, 2) * V(jj, 0) * U(rr, 2) * V(ss, 0);
                dP[e] = v * H[33];
            }
            __syncthreads();
            // ---- new base cell?  flush the register accumulators against the OLD rows, then load the new rows
            bool changed = !have_cell || cell[0] != cur[0] || cell[1] != cur[1] || cell[2] != cur[2];
            if (changed) {
                if (have_cell) flush();
                __syncthreads();
                cur[0] = cell[0], cur[1] = cell[1], cur[2] = cell[2];
                have_cell = true;
                if (tid < 27) {
                    int i = tid / 9, j = (tid / 3) % 3, k = tid % 3;
                    rowdof[tid] = tidx[((cur[0] - ox + i) * TY + (cur[1] - oy + j)) * TZ + (cur[2] - oz + k)];
                }
            }
            // ---- T_i[a + 3*(b + 3 q)] = sum_v dP[(a + 3 v) + 9 (b + 3 q)] g_i[v]
            for (int e = tid; e < 729; e += 256) {
                int i = e / 27, abq = e - i * 27;
                int a = abq % 3, bq = abq / 3;
                Ti[i][abq] = dP[(a + 0) + 9 * bq] * gvec[i][0] + dP[(a + 3) + 9 * bq] * gvec[i][1] + dP[(a + 6) + 9 * bq] * gvec[i][2];
            }
            __syncthreads();
            // ---- pair blocks: delta[a][b] = sum_q T_i[a,(b,q)] g_j[q]
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s == 1 && !has2) continue;
                int i = pi[s], j = pj[s];
                T g0 = gvec[j][0], g1 = gvec[j][1], g2 = gvec[j][2];
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int a = 0; a < 3; ++a) accm[s][b * 3 + a] += Ti[i][a + 3 * (b + 0)] * g0 + Ti[i][a + 3 * (b + 3)] * g1 + Ti[i][a + 3 * (b + 6)] * g2;
            }
            __syncthreads();
        }
    }
    if (have_cell) flush();
}

#endif

// BC projection of the assembled system (ImplicitSolver.h:554-593)
template <class T>
__global__ void k_bc_project_matrix(const int32_t* __restrict__ col, T* val, const int32_t* __restrict__ bcIdx, const T* __restrict__ bcR, const T* __restrict__ bcRinv,
    const uint8_t* __restrict__ bcSlip, int nn, const uint8_t* __restrict__ own)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)nn * 125) return;
    int i = (int)(e / 125);
    if (own && !own[i]) return; // sharded: rows of other ranks
    int j = col[e];
    int ic = bcIdx[i], jc = bcIdx[j];
    if (ic < 0 && jc < 0) return;
    T* v = val + e * 9;
    bool zero_pad = true;
#pragma unroll
    for (int c = 0; c < 9; ++c) zero_pad = zero_pad && v[c] == (T)0;
    bool iSlip = ic >= 0 && bcSlip[ic], jSlip = jc >= 0 && bcSlip[jc];
    if ((ic >= 0 && !iSlip) || (jc >= 0 && !jSlip)) {
        bool diag = (j == i) && ((e - (int64_t)i * 125) == 62);
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c] = (diag && c % 4 == 0) ? (T)1 : (T)0;
        return;
    }
    if (zero_pad) return; // padded slot (column 0/1 alias): nothing to rotate
    Mat3<T> M;
#pragma unroll
    for (int c = 0; c < 9; ++c) M.a[c] = v[c];
    if (iSlip) {
        Mat3<T> R;
#pragma unroll
        for (int c = 0; c < 9; ++c) R.a[c] = bcR[9 * ic + c];
        M = m3_mul(R, M);
    }
    if (jSlip) {
        Mat3<T> R;
#pragma unroll
        for (int c = 0; c < 9; ++c) R.a[c] = bcRinv[9 * jc + c];
        M = m3_mul(M, R);
    }
    if (iSlip) M(0, 0) = 0, M(0, 1) = 0, M(0, 2) = 0;
    if (jSlip) M(0, 0) = 0, M(1, 0) = 0, M(2, 0) = 0;
    if (i == j && (e - (int64_t)i * 125) == 62) M(0, 0) = 1;
#pragma unroll
    for (int c = 0; c < 9; ++c) v[c] = M.a[c];
}

// buildDiagonal (SquareMatrix.h:301-324): D_i = sum of entries whose column is i; scaler by Ainv; block inverse
template <class T>
__global__ void k_diag(const int32_t* __restrict__ col, const T* __restrict__ val, T* diagVal, T* diagInv, T* diagBlockInv, int n, int Ainv, int stencil_order, const uint8_t* __restrict__ own)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (own && !own[i])) return;
    Mat3<T> D;
#pragma unroll
    for (int c = 0; c < 9; ++c) D.a[c] = (T)0;
    // rows still in stencil-slot order: the only slot whose column is i is the centre one (62); padded slots name column 0
    // (or 1 in row 0) and hold zeros (SquareMatrix.h:563-566, ImplicitSolver.h:594-602)
    for (int k = stencil_order ? 62 : 0; k < (stencil_order ? 63 : 125); ++k) {
        if (col[(int64_t)i * 125 + k] == i) {
            const T* v = val + ((int64_t)i * 125 + k) * 9;
#pragma unroll
            for (int c = 0; c < 9; ++c) D.a[c] += v[c];
        }
    }
    // block inverse by cofactors like Eigen's 3x3 inverse(), on the block scaled by an exact power of two: bit-identical
    // in the normal range, and the determinant of a very light node (m ~ 1e-14 => det ~ 1e-42) no longer underflows in
    // fp32
    T amax = (T)0;
#pragma unroll
    for (int c = 0; c < 9; ++c) amax = fmax(amax, habs(D.a[c]));
    const int ex = (amax > (T)0 && amax < (T)INFINITY) ? ilogb(amax) : 0;
    Mat3<T> Ds;
#pragma unroll
    for (int c = 0; c < 9; ++c) Ds.a[c] = scalbn(D.a[c], -ex);
    Mat3<T> Bi = m3_inverse(Ds);
#pragma unroll
    for (int c = 0; c < 9; ++c) Bi.a[c] = scalbn(Bi.a[c], -ex);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        diagVal[9 * (int64_t)i + c] = D.a[c];
        diagBlockInv[9 * (int64_t)i + c] = Bi.a[c];
        diagInv[9 * (int64_t)i + c] = Ainv == 0 ? ((c % 4 == 0) ? (T)1 / D.a[c] : (T)0) : Bi.a[c];
    }
}

template <class T>
void Ctx<T>::build_diagonal(Level<T>& L)
{
    L.diagVal.reserve(9 * (size_t)L.n), L.diagInv.reserve(9 * (size_t)L.n), L.diagBlockInv.reserve(9 * (size_t)L.n);
    HOT_LAUNCH(this, "build_diagonal", k_diag<T>, div_up(L.n, 256), 256, 0, L.col.p, L.val.p, L.diagVal.p, L.diagInv.p, L.diagBlockInv.p, L.n, cfg.Ainv, L.split ? 0 : 1, L.mask());
    if (L.part && !halo_mode()) // first-generation sharding: the diagonal blocks are used by the replicated vector algebra of the smoothers (scalers) as well: every rank gets all of them
        exchange(L, L.diagVal.p, -1, 9), exchange(L, L.diagInv.p, -1, 9), exchange(L, L.diagBlockInv.p, -1, 9);
}

template <class T>
__global__ __launch_bounds__(256) void k_count_nnzb(const T* __restrict__ val, int64_t nblocks, unsigned long long* out, const uint8_t* __restrict__ own)
{
    __shared__ double red[4];
    double c = 0;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nblocks; b += (int64_t)gridDim.x * 256) {
        if (own && !own[b / 125]) continue;
        const T* v = val + b * 9;
        bool nz = false;
#pragma unroll
        for (int k = 0; k < 9; ++k) nz = nz || v[k] != (T)0;
        c += nz ? 1.0 : 0.0;
    }
    double t = block_sum_256<double>(c, red);
    if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)t);
}
template <class T>
void Ctx<T>::count_nnzb(Level<T>& L)
{
    unsigned long long* d = (unsigned long long*)(dscal.p + 120);
    HOT_HIP(hipMemsetAsync(d, 0, 8, stream));
    HOT_LAUNCH(this, "count_nnzb", k_count_nnzb<T>, std::min(div_up((size_t)L.n * 125, 256), 2048), 256, 0, L.val.p, (int64_t)L.n * 125, d, L.mask());
    unsigned long long h = 0;
    HOT_HIP(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, stream));
    sync();
    int64_t total = (int64_t)h;
    if (L.part) c_allreduce(&total, 1, HOT_COMM_I64, HOT_COMM_SUM, false); // every rank counted the rows it owns
    L.nnzb = (long long)total;
}

template <class T>
void Ctx<T>::build_hessian()
{
    need(Nn > 0 && dt > 0, "hot_build_hessian before hot_update_state");
    double t0 = wall_ms();
    const bool keep0 = halo_mode() && !levels.empty() && levels[0]->halo.built && levels[0]->n == Nn; // halo mode: level 0 was set up by hot_p2g
    release_levels(keep0 ? 1 : 0);
    Level<T>* L = keep0 ? levels[0] : acquire_level(0);
    if (!keep0) levels.push_back(L);
    L->n = Nn;
    L->built = false, L->split = false;
    size_t ne = (size_t)Nn * 125;
    L->col.reserve(ne), L->val.reserve(ne * 9), L->coord.reserve(3 * (size_t)Nn);
    if (!keep0) {
        HOT_HIP(hipMemcpyAsync(L->coord.p, id2coord.p, 3 * (size_t)Nn * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        L->part = false, L->colored = false;
    }
    if (sharded() && !keep0) { // row ownership of level 0 (always partitioned): needs the colouring, which only reads the coordinates
        L->nstart = nstart0;
        color_level(*L);
        level_ownership(*L);
    }
    const bool v1 = ab_flag("HOT_HESSIAN_V1"); // A/B build only: per-cell global-atomic scatter kernel
    HOT_LAUNCH(this, "hessian_fill_cols", k_fill_cols<T>, div_up(ne, 256), 256, 0, block_map, gIdx.p, id2coord.p, mass.p, L->col.p, L->val.p, Nn, v1 ? 1 : 0);
#ifdef HOT_AB_KERNELS
    if (v1)
        HOT_LAUNCH(this, "hessian_assemble_v1", k_hessian<T>, Ng, 256, 0, pX.p, pFn.p, pFt.p, pVol.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_nb.p, gIdx.p, L->val.p, dx, (T)1 / dx,
            dt, cfg.project);
    else if (ab_flag("HOT_HESSIAN_TILES") || ab_flag("HOT_HESSIAN_TILES_V1") || ab_flag("HOT_HESSIAN_MFMA"))
        assemble_tiles(*L); // rounds 2 - 4: particle chunks staged in LDS
    else
#endif
        assemble_rows(*L);
    if (L->part) exchange_rows(*L, written.p); // rows near the shard boundary: the other side's particles contribute as well
    if (cfg.systemBCProject && Nc > 0)
        HOT_LAUNCH(this, "hessian_bc_project", k_bc_project_matrix<T>, div_up(ne, 256), 256, 0, L->col.p, L->val.p, bcIdx.p, bcR.p, bcRinv.p, bcSlip.p, Nn, L->mask());
    build_diagonal(*L);
    L->nnzb = -1; // counted on request (hot_get_level_nnzb)
    sync();
    stats.ms_hessian += wall_ms() - t0;
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
