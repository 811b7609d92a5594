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
#include "ab_src/hessian_ab1.hip"
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
