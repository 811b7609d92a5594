// libhotmi355x — Hessian assembly of rounds 1 - 4 ("row-tile owns its rows", particle chunks staged in LDS): A/B build only since round 5
// (production: hessian_rows.hip); the cell table of the sorted particles and the matrix-free block diagonal, which are production code.
//
// Same mathematics as hessian.hip's k_hessian (reference Projects/multigrid/ImplicitSolver.h:498-552): every ordered
// node pair (i, j) of every particle contributes  V_p dt^2 sum_{v,q} dP_{(a,v),(b,q)} g_i[v] g_j[q]  to row dof_i,
// slot linearOffset(node_i - node_j).  The first version scattered those blocks with global fp64 atomics
// (1.7e9 of them at 2 M particles: atomic-throughput bound, 76 ms).  Here:
//   pass 1  k_dpdf45        per particle: SVD, PSD-projected A/B blocks, rotate the 21 couplings into the symmetric
//                           9x9 V_p dt^2 dP/dF, stored as 45 scalars (SoA).
//   pass 2  k_hessian_tiles one workgroup per aligned 2x2x2 tile of grid nodes.  The 8 rows x 125 slots x 9 values
//                           live in LDS (72 KB fp64).  Only particles whose base cell lies in the 4x4x4 cells around
//                           the tile touch these rows; they are fetched cell by cell through the per-cell ranges of
//                           the sorted particle array, 64 at a time (dP, X, Fn staged in LDS, g = Fn^T grad w for the
//                           27 nodes computed once per particle).  The particles of one base cell share their 27
//                           support nodes, so a (cell segment, tile row, column node) item sums its 3x3 block over
//                           the segment in registers and adds it to the LDS tile once (9 ds_add per item).
//                           At the end the tile is written to HBM with plain coalesced stores (mass term folded in).
//   k_mf_diag_col / k_mf_diag_finish: the block diagonal of the matrix-free operator (buildDiagonal) from the same 45
//                           scalars, for the --matfree preconditioner.
//   Each (particle, i, j) block is computed exactly once in the whole launch (by the tile owning row i); a particle's
//   45 scalars are re-read by the <= 8 tiles its 3x3x3 support intersects.
#include "hot_impl.h"
#include "hot_constitutive.h"

namespace hot {

__host__ __device__ constexpr int sym45(int a, int b) { return a <= b ? (a * 9 - (a * (a - 1)) / 2 + (b - a)) : (b * 9 - (b * (b - 1)) / 2 + (a - b)); }

template <class T>
__global__ __launch_bounds__(256) void k_dpdf45(const T* __restrict__ Ft, const T* __restrict__ Vol, const T* __restrict__ Mu, const T* __restrict__ Lam, T* __restrict__ dp, int64_t Np,
    T dt, int project)
{
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= Np) return;
    Mat3<T> Fc;
#pragma unroll
    for (int c = 0; c < 9; ++c) Fc.a[c] = Ft[(int64_t)c * Np + p];
    HessBlocks<T> h;
    corotated_hessian(Fc, Mu[p], Lam[p], project != 0, h);
    const T sc = Vol[p] * dt * dt;
    // W[(a',b'),(r,s)] = sum_{c,d} K_{a'b',cd} U(r,c) V(s,d): first rotate the right index pair, then the left one
    // (2 x 9 x 21 / 9 x 9 x 9 multiply-adds instead of 45 x 21 four-factor products)
    const Mat3<T>& U = h.U;
    const Mat3<T>& V = h.V;
    // K is non-zero only for (aa,cc) [A], (ab,ab) and (ab,ba) [B blocks]
    auto Kval = [&](int a, int b, int c, int d) -> T {
        if (a == b && c == d) return h.A(a, c);
        if (a != b && ((a == c && b == d) || (a == d && b == c))) {
            int lo = a < b ? a : b, hi = a < b ? b : a;
            const T* B = (lo == 0 && hi == 1) ? h.B01 : ((lo == 1 && hi == 2) ? h.B12 : h.B20);
            // B01: rows/cols ordered (01, 10); B12: (12, 21); B20: (20, 02)
            int ia, ic;
            if (lo == 0 && hi == 2) {
                ia = (a == 2) ? 0 : 1, ic = (c == 2) ? 0 : 1;
            }
            else {
                ia = (a == lo) ? 0 : 1, ic = (c == lo) ? 0 : 1;
            }
            return B[ia + ic];
        }
        return (T)0;
    };
#pragma unroll
    for (int ij = 0; ij < 9; ++ij) {
        const int jj = ij / 3, ii = ij - jj * 3;
#pragma unroll
        for (int rs = ij; rs < 9; ++rs) {
            const int ss = rs / 3, rr = rs - ss * 3;
            T v = (T)0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    T ub = U(ii, a) * V(jj, b);
                    if (a == b) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) v += Kval(a, a, c, c) * ub * U(rr, c) * V(ss, c);
                    }
                    else {
                        v += Kval(a, b, a, b) * ub * U(rr, a) * V(ss, b) + Kval(a, b, b, a) * ub * U(rr, b) * V(ss, a);
                    }
                }
            dp[(int64_t)sym45(ij, rs) * Np + p] = v * sc;
        }
    }
}

// ---- cell table: particles sorted by cell key; (key -> [first, next))
__global__ void k_cell_heads(const uint64_t* __restrict__ keys, int32_t* flags, int64_t n, int index_bits)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    flags[p] = (p == 0 || (keys[p] >> index_bits) != (keys[p - 1] >> index_bits)) ? 1 : 0;
}
__global__ void k_cell_fill(const uint64_t* __restrict__ keys, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int32_t* cell_first, HashMap h, int64_t n,
    int ncell, int index_bits)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (p == 0) cell_first[ncell] = (int32_t)n;
    if (!flags[p]) return;
    int c = scan[p];
    cell_first[c] = (int32_t)p;
    uint32_t s = hash_insert_min(h, keys[p] >> index_bits, (unsigned long long)c);
    h.id[s] = c;
}
__global__ void k_hash_clear3(HashMap h)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > h.mask) return;
    h.keys[i] = ~0ULL;
    h.minrank[i] = ~0ULL;
    h.id[i] = -1;
}

// groups start on cell boundaries: the first cell of group g is the cell whose first particle is group_first[g]
__global__ void k_group_cell0(const int32_t* __restrict__ group_first, const int32_t* __restrict__ cell_first, int32_t* group_cell0, int ng, int ncell)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > ng) return;
    int target = group_first[g], lo = 0, hi = ncell; // cell_first[lo] <= target ; cell_first[ncell] = Np
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cell_first[mid] <= target)
            lo = mid;
        else
            hi = mid;
    }
    group_cell0[g] = g == ng ? ncell : lo;
}

template <class T>
void Ctx<T>::build_cell_table()
{
    constexpr int index_bits = 32 - G::block_bits;
    int64_t n = Np;
    // keys2 still holds the sorted keys of this step's hot_sort
    flags.reserve(n), scan.reserve(n);
    HOT_LAUNCH(this, "cell_heads", k_cell_heads, div_up(n, 256), 256, 0, keys2.p, flags.p, n, index_bits);
    Ncell = exclusive_scan_i32(flags.p, scan.p, n);
    uint32_t cap = 1024;
    while (cap < 2u * (uint32_t)Ncell + 16u) cap <<= 1;
    cell_first.reserve(Ncell + 1), ch_keys.reserve(cap), ch_rank.reserve(cap), ch_id.reserve(cap);
    cell_map.keys = ch_keys.p, cell_map.minrank = ch_rank.p, cell_map.id = ch_id.p, cell_map.mask = cap - 1;
    HOT_LAUNCH(this, "hash_clear", k_hash_clear3, div_up(cap, 256), 256, 0, cell_map);
    HOT_LAUNCH(this, "cell_fill", k_cell_fill, div_up(n, 256), 256, 0, keys2.p, flags.p, scan.p, cell_first.p, cell_map, n, Ncell, index_bits);
    group_cell0.reserve(Ng + 1);
    HOT_LAUNCH(this, "group_cell0", k_group_cell0, div_up(Ng + 1, 256), 256, 0, group_first.p, cell_first.p, group_cell0.p, Ng, Ncell);
}

#ifdef HOT_AB_KERNELS
#include "ab_src/hessian_tiles_ab1.hip"
#endif

#ifdef HOT_AB_KERNELS
#include "ab_src/hessian_tiles_ab2.hip"
#endif // HOT_AB_KERNELS

// ------------------------------------------------------------------------------------------------ matrix-free diagonal
// buildDiagonal (reference Projects/multigrid/ImplicitSolver.h:605-665): the 3x3 diagonal blocks of the matrix-free
// operator,  D_i = m_i I + dt^2 sum_p V_p sum_{v,q} ddF[(.,v),(.,q)] g_i[v] g_i[q],  g_i = Fn^T grad w_i.  Column `cc`
// of every block per launch (3 quantities, the same partial-tile + ordered reduce path as the force scatter).
template <class T>
__global__ __launch_bounds__(256) void k_mf_diag_col(const T* __restrict__ X, const T* __restrict__ Fn, const T* __restrict__ dp, int64_t Np, const int32_t* __restrict__ group_first,
    const int32_t* __restrict__ group_origin, T* __restrict__ part, T one_over_dx, int cc)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    using AT = AccT<T>; // double tile also in fp32 (LDS float atomics are slow, see k_force_cells)
    __shared__ AT acc[3][TILE];
    const int g = blockIdx.x;
    for (int t = threadIdx.x; t < 3 * TILE; t += 256) (&acc[0][0])[t] = (AT)0;
    __syncthreads();
    const int first = group_first[g], last = group_first[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    for (int p = first + threadIdx.x; p < last; p += 256) {
        T xp[3] = { X[p], X[Np + p], X[2 * Np + p] };
        int base[3];
        T w[3][3], dw[3][3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xp[d], base[d], w[d], dw[d]);
        T F9[9], D[45];
#pragma unroll
        for (int c = 0; c < 9; ++c) F9[c] = Fn[(int64_t)c * Np + p];
#pragma unroll
        for (int q = 0; q < 45; ++q) D[q] = dp[(int64_t)q * Np + p];
        for (int n = 0; n < 27; ++n) {
            const int i = n / 9, j = (n / 3) % 3, k = n % 3;
            const T wi = i == 0 ? w[0][0] : (i == 1 ? w[0][1] : w[0][2]), dwi = i == 0 ? dw[0][0] : (i == 1 ? dw[0][1] : dw[0][2]);
            const T wj = j == 0 ? w[1][0] : (j == 1 ? w[1][1] : w[1][2]), dwj = j == 0 ? dw[1][0] : (j == 1 ? dw[1][1] : dw[1][2]);
            const T wk = k == 0 ? w[2][0] : (k == 1 ? w[2][1] : w[2][2]), dwk = k == 0 ? dw[2][0] : (k == 1 ? dw[2][1] : dw[2][2]);
            const T g0 = one_over_dx * dwi * wj * wk, g1 = wi * one_over_dx * dwj * wk, g2 = wi * wj * one_over_dx * dwk;
            T gi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gi[c] = F9[c * 3] * g0 + F9[c * 3 + 1] * g1 + F9[c * 3 + 2] * g2; // (Fn^T grad w)[c] = sum_r Fn(r, c) gw[r]
            const int t = ((base[0] - ox + i) * TY + (base[1] - oy + j)) * TZ + (base[2] - oz + k);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                T v = (T)0;
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int vv = 0; vv < 3; ++vv) v += D[sym45(a + 3 * vv, cc + 3 * q)] * gi[vv] * gi[q];
                lds_atomic_add(&acc[a][t], (AT)v);
            }
        }
    }
    __syncthreads();
    T* out = part + (int64_t)g * 3 * TILE;
    for (int t = threadIdx.x; t < 3 * TILE; t += 256) out[t] = (T)(&acc[0][0])[t];
}
template <class T>
__global__ void k_mf_diag_finish(const T* __restrict__ tile /*[9][slots]: column-major blocks*/, const int32_t* __restrict__ dofSlot, const T* __restrict__ mass, T* __restrict__ dinv, int nn,
    int64_t slots, int Ainv)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    const int64_t s = dofSlot[n];
    Mat3<T> D;
#pragma unroll
    for (int c = 0; c < 9; ++c) D.a[c] = tile[(int64_t)c * slots + s];
    const T m = mass[n];
    D.a[0] += m, D.a[4] += m, D.a[8] += m;
    Mat3<T> R;
    if (Ainv == 0) {
#pragma unroll
        for (int c = 0; c < 9; ++c) R.a[c] = (c % 4 == 0) ? (T)1 / D.a[c] : (T)0;
    }
    else
        R = m3_inverse(D);
#pragma unroll
    for (int c = 0; c < 9; ++c) dinv[9 * (int64_t)n + c] = R.a[c];
}

template <class T>
void Ctx<T>::matfree_diagonal(T* dinv)
{
    int64_t slots = (int64_t)Nb * EPB;
    pDP.reserve(45 * (size_t)Np);
    HOT_LAUNCH(this, "hessian_dpdf", k_dpdf45<T>, div_up(Np, 256), 256, 0, pFt.p, pVol.p, pMu.p, pLam.p, pDP.p, Np, dt, cfg.project);
    DBuf<T>& tile = ap; // scratch (9 * slots)
    tile.reserve(9 * slots);
    for (int cc = 0; cc < 3; ++cc) {
        HOT_LAUNCH(this, "matfree_diag_scatter", k_mf_diag_col<T>, Ng, 256, 0, pX.p, pFn.p, pDP.p, Np, group_first.p, group_origin.p, gPart.p, (T)1 / dx, cc);
        reduce_tiles(3, tile.p + (3 * cc) * slots, tile.p + (3 * cc + 1) * slots, tile.p + (3 * cc + 2) * slots, nullptr, nullptr, "matfree_diag_reduce");
    }
    if (halo_mode()) {
        T* arr[9];
        for (int k = 0; k < 9; ++k) arr[k] = tile.p + (int64_t)k * slots;
        tile_exchange(arr, 9);
    }
    else if (sharded())
        allreduce_tiles(tile.p, 9);
    HOT_LAUNCH(this, "matfree_diag_finish", k_mf_diag_finish<T>, div_up(Nn, 256), 256, 0, tile.p, dofSlot.p, mass.p, dinv, Nn, slots, cfg.Ainv);
}

template void Ctx<float>::build_cell_table();
template void Ctx<double>::build_cell_table();
template void Ctx<float>::matfree_diagonal(float*);
template void Ctx<double>::matfree_diagonal(double*);
#ifdef HOT_AB_KERNELS
template void Ctx<float>::assemble_tiles(Level<float>&);
template void Ctx<double>::assemble_tiles(Level<double>&);
#endif

} // namespace hot
