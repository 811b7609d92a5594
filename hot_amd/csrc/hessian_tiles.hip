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
constexpr int HT_THREADS = 1024; // one workgroup per CU (the LDS tile), so the workgroup itself must supply the waves

template <class T>
struct TileLds {
    static constexpr int CH = 64; // particles per chunk
    static constexpr size_t bytes = (size_t)8 * 1125 * sizeof(T) /*tile*/ + (size_t)CH * 45 * sizeof(T) /*dP*/ + (size_t)CH * 81 * sizeof(T) /*g*/ + (64 + 65 + 8 + CH * 3 + CH * 8 + 4 + CH + 67) * sizeof(int32_t) + (size_t)CH * 12 * sizeof(T);
};

template <class T>
__global__ __launch_bounds__(HT_THREADS) void k_hessian_tiles(const T* __restrict__ X, const T* __restrict__ Fn, const T* __restrict__ dp, int64_t Np, const uint64_t* __restrict__ blocks,
    const int32_t* __restrict__ gIdx, const int32_t* __restrict__ cell_first, HashMap cmap, const T* __restrict__ mass, T* __restrict__ val, T one_over_dx, int ntiles)
{
    using G = Geo<T>;
    constexpr int CH = TileLds<T>::CH;
    constexpr int TPBX = G::BX / 2, TPBY = G::BY / 2, TPBZ = G::BZ / 2, TPB = TPBX * TPBY * TPBZ; // 2x2x2 tiles per block
    extern __shared__ __attribute__((aligned(16))) char ht_smem[];
    T* tile = (T*)ht_smem; // [8][1125]
    T* sdp = tile + 8 * 1125; // [CH][45]
    T* sg = sdp + CH * 45; // [CH][27][3]
    int32_t* cstart = (int32_t*)(sg + CH * 81); // [64] first particle of each contributing cell
    int32_t* cpref = cstart + 64; // [65] prefix of particle counts
    int32_t* rdof = cpref + 65; // [8]
    int32_t* pbase = rdof + 8; // [CH][3] base node relative to the tile origin
    int32_t* items = pbase + CH * 3; // [CH*8] packed (particle-in-chunk << 3 | row)
    int32_t* nitems = items + CH * 8;
    int32_t* pidx = nitems + 4; // [CH] global particle index of each chunk member
    int32_t* segs = pidx + CH; // [64] cell segments of the chunk: cell | first << 8 | end << 16
    T* sxf = (T*)(segs + 67); // [CH][12] X and Fn of the chunk members (67: keeps the int area a multiple of 8 bytes)
    const int tid = threadIdx.x;
    // workgroup i runs on XCD i % 8 (MI355X_MICROARCH.md, dispatch).  Runs of 32 consecutive tiles (4-8 SPGrid blocks)
    // share most of their particle records: give each run to one XCD so that its L2 serves the re-reads.
    const int id = blockIdx.x, run = (id & 7) + 8 * (id >> 8), tile_id = run * 32 + ((id >> 3) & 31);
    if (tile_id >= ntiles) return;
    const int b = tile_id / TPB, tt = tile_id % TPB;
    int bx, by, bz;
    G::linear_to_coord(blocks[b], bx, by, bz);
    const int tx0 = bx + 2 * (tt / (TPBY * TPBZ)), ty0 = by + 2 * ((tt / TPBZ) % TPBY), tz0 = bz + 2 * (tt % TPBZ); // tile origin (node coords)
    if (tid < 8) {
        int ex = (tx0 - bx) + (tid >> 2), ey = (ty0 - by) + ((tid >> 1) & 1), ez = (tz0 - bz) + (tid & 1);
        int elem = (ex << (G::yb + G::zb)) | (ey << G::zb) | ez;
        rdof[tid] = gIdx[(int64_t)b * G::EPB + elem];
    }
    if (tid < 64) {
        // contributing base cells: tile origin + (-2..1)^3
        int cx = tx0 - 2 + (tid >> 4), cy = ty0 - 2 + ((tid >> 2) & 3), cz = tz0 - 2 + (tid & 3);
        int first = 0, cnt = 0;
        if ((cx | cy | cz) >= 0) {
            int32_t c = hash_find_id(cmap, G::linear_offset(cx, cy, cz) >> G::data_bits);
            if (c >= 0) first = cell_first[c], cnt = cell_first[c + 1] - first;
        }
        cstart[tid] = first;
        cpref[tid + 1] = cnt;
    }
    for (int e = tid; e < 8 * 1125; e += HT_THREADS) tile[e] = (T)0;
    __syncthreads();
    bool any = false;
    for (int r = 0; r < 8; ++r) any = any || rdof[r] >= 0;
    if (!any) return;
    if (tid == 0) {
        cpref[0] = 0;
        for (int c = 0; c < 64; ++c) cpref[c + 1] += cpref[c];
    }
    __syncthreads();
    const int total = cpref[64];
    for (int chunk = 0; chunk < total; chunk += CH) {
        const int cnt = min(CH, total - chunk);
        if (tid == 0) nitems[0] = 0, nitems[1] = 0;
        if (tid < cnt) {
            int flat = chunk + tid;
            int lo = 0, hi = 64; // cpref[lo] <= flat < cpref[hi]
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (cpref[mid] <= flat)
                    lo = mid;
                else
                    hi = mid;
            }
            pidx[tid] = cstart[lo] + (flat - cpref[lo]);
        }
        __syncthreads();
        // ---- stage the chunk: dP (45), g = Fn^T grad w (27 x 3), tile-relative base node, work items
        for (int e = tid; e < cnt * 45; e += HT_THREADS) {
            int l = e / 45, q = e - l * 45;
            int p = pidx[l];
            sdp[l * 45 + q] = dp[(int64_t)q * Np + p];
        }
        for (int e = tid; e < cnt * 12; e += HT_THREADS) { // X (3) and Fn (9) of every chunk member, once
            int l = e / 12, q = e - l * 12;
            int p = pidx[l];
            sxf[l * 12 + q] = q < 3 ? X[(int64_t)q * Np + p] : Fn[(int64_t)(q - 3) * Np + p];
        }
        __syncthreads();
        for (int e = tid; e < cnt * 27; e += HT_THREADS) {
            int l = e / 27, nd = e - l * 27;
            const T* xf = sxf + l * 12;
            int base[3];
            T w[3][3], dw[3][3];
#pragma unroll
            for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xf[d], base[d], w[d], dw[d]);
            int i = nd / 9, j = (nd / 3) % 3, k = nd % 3;
            T wi = i == 0 ? w[0][0] : (i == 1 ? w[0][1] : w[0][2]), dwi = i == 0 ? dw[0][0] : (i == 1 ? dw[0][1] : dw[0][2]);
            T wj = j == 0 ? w[1][0] : (j == 1 ? w[1][1] : w[1][2]), dwj = j == 0 ? dw[1][0] : (j == 1 ? dw[1][1] : dw[1][2]);
            T wk = k == 0 ? w[2][0] : (k == 1 ? w[2][1] : w[2][2]), dwk = k == 0 ? dw[2][0] : (k == 1 ? dw[2][1] : dw[2][2]);
            T g0 = one_over_dx * dwi * wj * wk, g1 = wi * one_over_dx * dwj * wk, g2 = wi * wj * one_over_dx * dwk;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) sg[(l * 27 + nd) * 3 + cc] = xf[3 + cc * 3] * g0 + xf[3 + cc * 3 + 1] * g1 + xf[3 + cc * 3 + 2] * g2;
        }
        __syncthreads();
        // ---- work items.  The particles of one base cell share their 27 support nodes, so for a (cell segment, tile
        // row r, column node jl) item the 3x3 block of every particle lands in the same (row, slot): it is summed in
        // registers over the segment and added to the LDS tile ONCE (9 ds_add per item instead of 9 per particle).
        if (tid < 64) {
            const int s0 = max(cpref[tid], chunk), s1 = min(cpref[tid + 1], chunk + cnt);
            if (s1 > s0) {
                const int k = atomicAdd(nitems + 1, 1);
                segs[k] = tid | ((s0 - chunk) << 8) | ((s1 - chunk) << 16); // cell, first, end (chunk-relative, <= CH)
            }
        }
        __syncthreads();
        const int nseg = nitems[1];
        for (int e = tid; e < nseg * 8; e += HT_THREADS) {
            const int sg_ = e >> 3, r = e & 7, cell = segs[sg_] & 255;
            const int ax = (r >> 2) - ((cell >> 4) - 2), ay = ((r >> 1) & 1) - (((cell >> 2) & 3) - 2), az = (r & 1) - ((cell & 3) - 2); // node index inside the kernel
            if ((unsigned)ax < 3u && (unsigned)ay < 3u && (unsigned)az < 3u && rdof[r] >= 0) items[atomicAdd(nitems, 1)] = e;
        }
        __syncthreads();
        const int ni = *nitems * 27;
        for (int it = tid; it < ni; it += HT_THREADS) {
            const int e = items[it / 27], j = it % 27;
            const int sd = segs[e >> 3], r = e & 7, cell = sd & 255, l0 = (sd >> 8) & 255, l1 = sd >> 16;
            const int ax = (r >> 2) - ((cell >> 4) - 2), ay = ((r >> 1) & 1) - (((cell >> 2) & 3) - 2), az = (r & 1) - ((cell & 3) - 2);
            const int i = ax * 9 + ay * 3 + az;
            const int jx = j / 9, jy = (j / 3) % 3, jz = j % 3;
            T acc[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) acc[q] = (T)0;
            for (int l = l0; l < l1; ++l) {
                const T* D = sdp + l * 45;
                const T* gi = sg + (l * 27 + i) * 3;
                const T* gj = sg + (l * 27 + j) * 3;
                T Gm[9]; // G[v + 3 q] = g_i[v] g_j[q]
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int v = 0; v < 3; ++v) Gm[v + 3 * q] = gi[v] * gj[q];
                // block(a, b) = sum_{v,q} dP[(a + 3 v), (b + 3 q)] G[v][q]
#pragma unroll
                for (int bb = 0; bb < 3; ++bb)
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        T t = acc[a + 3 * bb];
#pragma unroll
                        for (int q = 0; q < 3; ++q)
#pragma unroll
                            for (int v = 0; v < 3; ++v) t += D[sym45(a + 3 * v, bb + 3 * q)] * Gm[v + 3 * q];
                        acc[a + 3 * bb] = t;
                    }
            }
            T* o = tile + r * 1125 + ((ax - jx + 2) * 25 + (ay - jy + 2) * 5 + (az - jz + 2)) * 9;
#pragma unroll
            for (int q = 0; q < 9; ++q) lds_atomic_add(o + q, acc[q]);
        }
        __syncthreads();
    }
    // ---- write the tile: inertia term M on the diagonal slot (ImplicitSolver.h:486-496)
    for (int e = tid; e < 8 * 1125; e += HT_THREADS) {
        int r = e / 1125, q = e - r * 1125;
        int dof = rdof[r];
        if (dof < 0) continue;
        T v = tile[e];
        if (q >= 62 * 9 && q < 63 * 9 && ((q - 62 * 9) % 4 == 0)) v += mass[dof];
        val[(int64_t)dof * 1125 + q] = v;
    }
}

#endif

#ifdef HOT_AB_KERNELS
// ---- second version of pass 2.  The first one evaluates block(i,j) = sum_{v,q} dP[(a,v),(b,q)] g_i[v] g_j[q] from scratch
// for every (particle, row, column): 81 multiply-adds and 51 LDS reads each, and its work-item phase is bound by the
// fp64 FMA rate of the CU (measured with clock64: 68 % of a tile's 124 us).  Here the contraction is split:
//   K phase      K_i[a][b][q] = sum_v dP[(a,v),(b,q)] g_i[v]        once per (particle, tile row in its support), kept in LDS
//   pair phase   block(i,j)[a][b] = sum_q K_i[a][b][q] g_j[q]        a lane owns (cell segment, row, a, the 3 columns j = (jx,jy,0..2)):
//                                                                     9 K reads + 9 g reads feed 27 multiply-adds for 3 block rows
// i.e. 27 instead of 81 multiply-adds per block.  Chunks are packed by LDS budget (particles and K entries); particles of
// cells that touch no active row of the tile are skipped altogether.
__constant__ uint8_t kSymRow[45] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 7, 7, 8 };
__constant__ uint8_t kSymCol[45] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 3, 4, 5, 6, 7, 8, 2, 3, 4, 5, 6, 7, 8, 3, 4, 5, 6, 7, 8, 4, 5, 6, 7, 8, 5, 6, 7, 8, 6, 7, 8, 7, 8, 8 };

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts)
__device__ __forceinline__ int wave_scan_incl(int x)
{
    const int t = x;
    x += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, t, 0x113, 0xf, 0xf, true); // row_shr:3
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xe, true); // row_shr:4, banks 1-3
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xc, true); // row_shr:8, banks 2-3
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true); // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true); // row_bcast:31 into rows 2 and 3
    return x;
}

// ---- MFMA 16x16x4 (one A and one B scalar per lane, 4 results per lane): D[i][j] += sum_k A[i][k] B[k][j] with A in lane
// i + 16 k, B in lane j + 16 k.  Result rows of a lane: f64 (v_mfma_f64_16x16x4_f64) row = (lane >> 4) + 4 reg, f32
// (v_mfma_f32_16x16x4_f32) row = 4 (lane >> 4) + reg; column = lane & 15 for both.
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
template <class T>
struct Mfma16;
template <>
struct Mfma16<double> {
    using Acc = v4f64;
    __device__ static __forceinline__ Acc mac(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct Mfma16<float> {
    using Acc = v4f32;
    __device__ static __forceinline__ Acc mac(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

template <class T>
struct TileLds2 {
    static constexpr int CH = sizeof(T) == 8 ? 40 : 64; // particles per chunk
    static constexpr int KMAX = sizeof(T) == 8 ? 136 : 224; // (particle, row) entries per chunk
    static constexpr int NINT = 64 * 3 + 8 + 2 * (64 * 4 + 2 * CH) + KMAX + 512 + 16 + 48; // cstart, ccnt, cmask, rdof, 2 x (segs, soff, sbase, ibase, pidx, pseg), entinfo, items, 2 x ctl, symtab
    static constexpr size_t bytes = (size_t)8 * 1125 * sizeof(AccT<T>) + ((size_t)CH * (81 + 81 + 12) + (size_t)KMAX * 27) * sizeof(T) + (size_t)NINT * sizeof(int32_t);
};

// Development aid (-DHOT_HT_CLOCKS, tools/hess_phases.sh): shader clocks thread 0 of every workgroup spends between the barriers
// of k_hessian_tiles2, summed per phase: 0 prologue, 1 - 4 phases A - D of the chunk loop, 5 write-out.
#ifdef HOT_HT_CLOCKS
__device__ unsigned long long ht_clk[8];
#define HT_CLK(i) \
    do { \
        if (tid == 0) { \
            const unsigned long long t_ = clock64(); \
            clk_[i] += t_ - t0_, t0_ = t_; \
        } \
    } while (0)
#else
#define HT_CLK(i)
#endif

template <class T, bool USE_MFMA>
__global__ __launch_bounds__(HT_THREADS) void k_hessian_tiles2(const T* __restrict__ X, const T* __restrict__ Fn, const T* __restrict__ dp, int64_t Np, const uint64_t* __restrict__ blocks,
    const int32_t* __restrict__ gIdx, const int32_t* __restrict__ cell_first, HashMap cmap, const T* __restrict__ mass, T* __restrict__ val, T one_over_dx, int ntiles,
    const uint8_t* __restrict__ own /*sharded: rows this rank owns (they get the inertia term here), else null*/, uint8_t* __restrict__ written /*sharded: rows this launch wrote*/)
{
    using G = Geo<T>;
    constexpr int CH = TileLds2<T>::CH, KMAX = TileLds2<T>::KMAX;
    constexpr int TPBY = G::BY / 2, TPBZ = G::BZ / 2, TPB = (G::BX / 2) * TPBY * TPBZ;
    extern __shared__ __attribute__((aligned(16))) char ht_smem[];
    using AT = AccT<T>; // the tile is double also in the fp32 build: LDS float atomics are ~40x slower (see k_force_cells)
    AT* tile = (AT*)ht_smem; // [8][1125]
    T* sdp = (T*)(tile + 8 * 1125); // [CH][9][9] full symmetric dP
    T* sg = sdp + CH * 81; // [CH][27][3]
    T* sxf = sg + CH * 81; // [CH][12]
    T* sk = sxf + CH * 12; // [KMAX][27]: K[a + 3 b][q]; before the K phase its head holds the per-axis spline weights [CH][3][6]
    int32_t* cstart = (int32_t*)(sk + KMAX * 27); // [64] first particle of each contributing cell
    int32_t* ccnt = cstart + 64; // [64] its particle count
    int32_t* cmask = ccnt + 64; // [64] tile rows inside its 3x3x3 support (and active)
    int32_t* rdof = cmask + 64; // [8]
    // chunk tables, two sets (the next chunk is packed while the current one is in its K phase):
    int32_t* segs = rdof + 8; // [2][64] cell | first << 8 | end << 16 (chunk-relative)
    int32_t* soff = segs + 2 * 64; // [2][64] offset of the segment inside its cell
    int32_t* sbase = soff + 2 * 64; // [2][64] first K entry of the segment
    int32_t* ibase = sbase + 2 * 64; // [2][64] first pair-phase item of the segment
    int32_t* pidx = ibase + 2 * 64; // [2][CH] global particle index
    int32_t* pseg = pidx + 2 * CH; // [2][CH] segment of the chunk member
    int32_t* entinfo = pseg + 2 * CH; // [KMAX] chunk member | node index of the row inside the member's kernel << 8
    int32_t* items = entinfo + KMAX; // [512] segment << 3 | row
    int32_t* ctl = items + 512; // [2][8] nitems, nseg, cnt, nent, floor(2^20 / cnt) + 1
    int32_t* symtab = ctl + 16; // [45] row | column << 4 of the packed upper triangle of a 9 x 9 matrix
    const int tid = threadIdx.x;
#ifdef HOT_HT_CLOCKS
    unsigned long long clk_[6] = { 0, 0, 0, 0, 0, 0 }, t0_ = clock64();
#endif
    const int id = blockIdx.x, run = (id & 7) + 8 * (id >> 8), tile_id = run * 32 + ((id >> 3) & 31); // runs of 32 tiles per XCD, as above
    if (tile_id >= ntiles) return;
    const int b = tile_id / TPB, tt = tile_id % TPB;
    int bx, by, bz;
    G::linear_to_coord(blocks[b], bx, by, bz);
    const int tx0 = bx + 2 * (tt / (TPBY * TPBZ)), ty0 = by + 2 * ((tt / TPBZ) % TPBY), tz0 = bz + 2 * (tt % TPBZ);
    if (tid < 8) {
        int ex = (tx0 - bx) + (tid >> 2), ey = (ty0 - by) + ((tid >> 1) & 1), ez = (tz0 - bz) + (tid & 1);
        rdof[tid] = gIdx[(int64_t)b * G::EPB + ((ex << (G::yb + G::zb)) | (ey << G::zb) | ez)];
    }
    if (tid < 45) symtab[tid] = kSymRow[tid] | (kSymCol[tid] << 4);
    for (int e = tid; e < 8 * 1125; e += HT_THREADS) tile[e] = (AT)0;
    __syncthreads();
    bool any = false;
    for (int r = 0; r < 8; ++r) any = any || rdof[r] >= 0;
    if (!any) return;
    // The chunk packer is the LAST wavefront (it has the fewest K-phase tasks): lane pl = cell
    const int pl = tid - (HT_THREADS - 64);
    int my_first = 0, my_n = 0, my_w = 0; // its lanes: first particle, particle count (0 if no row is touched), rows touched of cell `pl`
    if (tid < 64) {
        const int ox = (tid >> 4) - 2, oy = ((tid >> 2) & 3) - 2, oz = (tid & 3) - 2; // base cell = tile origin + (-2..1)^3
        const int cx = tx0 + ox, cy = ty0 + oy, cz = tz0 + oz;
        int first = 0, cnt = 0, mask = 0;
        if ((cx | cy | cz) >= 0) {
            int32_t c = hash_find_id(cmap, G::linear_offset(cx, cy, cz) >> G::data_bits);
            if (c >= 0) first = cell_first[c], cnt = cell_first[c + 1] - first;
        }
        for (int r = 0; r < 8; ++r) {
            const int ax = (r >> 2) - ox, ay = ((r >> 1) & 1) - oy, az = (r & 1) - oz; // row node inside the cell's kernel
            if ((unsigned)ax < 3u && (unsigned)ay < 3u && (unsigned)az < 3u && rdof[r] >= 0) mask |= 1 << r;
        }
        cstart[tid] = first, ccnt[tid] = cnt, cmask[tid] = mask;
    }
    int pk_c = 0, pk_off = 0; // packing cursor: cell, offset inside it
    __syncthreads();
    if (pl >= 0) my_first = cstart[pl], my_w = __popc(cmask[pl]), my_n = my_w ? ccnt[pl] : 0;
    if (own) { // sharded: a tile none of whose rows this rank owns and none of whose cells hold particles of its shard is not its business
        bool mine = false;
        for (int r = 0; r < 8; ++r) mine = mine || (rdof[r] >= 0 && own[rdof[r]]);
        bool any_particles = false;
        for (int c = 0; c < 64; ++c) any_particles = any_particles || (ccnt[c] > 0 && cmask[c] != 0);
        if (!mine && !any_particles) return; // workgroup-uniform (LDS tables)
    }
    // ---- pack a chunk into table set `nb`: whole or partial cells until CH particles or KMAX entries.  One wavefront, lane =
    // cell: the particles (and K entries) from the chunk start to the end of each cell by two prefix sums, the first cell that
    // does not fit whole is split.
    auto pack = [&](int nb) {
        const int rem = pl < pk_c ? 0 : (pl == pk_c ? my_n - pk_off : my_n);
        const int pin = wave_scan_incl(rem), ein = wave_scan_incl(rem * my_w);
        const unsigned long long notfull = __ballot(pin > CH || ein > KMAX);
        const int f = notfull ? __ffsll((long long)notfull) - 1 : 64;
        const int pbef = pin - rem, ebef = ein - rem * my_w;
        int take = pl < f ? rem : 0;
        if (pl == f) take = min(min(rem, CH - pbef), (KMAX - ebef) / my_w); // my_w > 0 here: rem > 0
        const unsigned long long inc = __ballot(take > 0);
        const int k = __popcll(inc & ((1ull << pl) - 1ull));
        const int iin = wave_scan_incl(take > 0 ? my_w : 0); // pair-phase items (segment, row) up to and including this cell
        if (take > 0) {
            const int off = pl == pk_c ? pk_off : 0;
            segs[nb * 64 + k] = pl | (pbef << 8) | ((pbef + take) << 16), soff[nb * 64 + k] = off, sbase[nb * 64 + k] = ebef, ibase[nb * 64 + k] = iin - my_w;
            for (int t = 0; t < take; ++t) pseg[nb * CH + pbef + t] = k, pidx[nb * CH + pbef + t] = my_first + off + t;
        }
        const int last = f < 64 ? f : 63;
        const int cnt_all = __shfl(pbef + take, last), ent_all = __shfl(ebef + take * my_w, last);
        const int take_f = __shfl(take, last);
        const int items_all = __shfl(iin, 63);
        if (pl == 0) ctl[nb * 8 + 0] = items_all, ctl[nb * 8 + 1] = __popcll(inc), ctl[nb * 8 + 2] = cnt_all, ctl[nb * 8 + 3] = ent_all, ctl[nb * 8 + 4] = cnt_all ? (1 << 20) / cnt_all + 1 : 0;
        if (f < 64) {
            pk_off = (f == pk_c ? pk_off : 0) + take_f;
            pk_c = f;
        }
        else
            pk_c = 64, pk_off = 0;
    };
    // ---- the chunk's dP (45 scalars), X and Fn go from global memory into registers one chunk ahead (issued before the
    // previous chunk's pair phase) and land in LDS at the top of the chunk.  Lanes run over the chunk members first
    // (neighbours in every SoA component).
    constexpr int NS = (CH * 45 + HT_THREADS - 1) / HT_THREADS;
    static_assert(CH * 12 <= HT_THREADS, "one X / Fn slot per thread");
    T ld[NS], ldx = (T)0;
    auto fetch = [&](int nb) {
        const int cnt = ctl[nb * 8 + 2];
        const unsigned magic = (unsigned)ctl[nb * 8 + 4]; // e / cnt = e * magic >> 20, exact for e * cnt < 2^20 (e < 45 * 64, cnt <= 64)
        const int32_t* pi = pidx + nb * CH;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int e = tid + k * HT_THREADS, q = (int)(((unsigned)e * magic) >> 20);
            ld[k] = e < cnt * 45 ? dp[(int64_t)q * Np + pi[e - q * cnt]] : (T)0;
        }
        if (tid < cnt * 12) {
            const int q = (int)(((unsigned)tid * magic) >> 20), p = pi[tid - q * cnt];
            ldx = q < 3 ? X[(int64_t)q * Np + p] : Fn[(int64_t)(q - 3) * Np + p];
        }
    };
    if (pl >= 0) pack(0);
    __syncthreads();
    fetch(0);
    HT_CLK(0);
    for (int cur = 0;; cur ^= 1) {
        const int nseg = ctl[cur * 8 + 1], cnt = ctl[cur * 8 + 2], nent = ctl[cur * 8 + 3];
        if (cnt == 0) break;
        const unsigned magic = (unsigned)ctl[cur * 8 + 4];
        const int32_t* ibase_c = ibase + cur * 64;
        const int32_t* segs_c = segs + cur * 64;
        const int32_t* sbase_c = sbase + cur * 64;
        const int32_t* pseg_c = pseg + cur * CH;
        // ---- phase A: entry / item tables, the staged values to LDS (dP 45 -> full 9x9), spline weights of the chunk's
        // (particle, axis) pairs straight from the X register
        for (int e = tid; e < cnt * 8; e += HT_THREADS) {
            const int l = e >> 3, r = e & 7, sp = pseg_c[l], sd = segs_c[sp], cell = sd & 255, mask = cmask[cell];
            if ((mask >> r) & 1) {
                const int ax = (r >> 2) - ((cell >> 4) - 2), ay = ((r >> 1) & 1) - (((cell >> 2) & 3) - 2), az = (r & 1) - ((cell & 3) - 2);
                entinfo[sbase_c[sp] + (l - ((sd >> 8) & 255)) * __popc(mask) + __popc(mask & ((1 << r) - 1))] = l | ((ax * 9 + ay * 3 + az) << 8);
            }
        }
        for (int e = tid; e < nseg * 8; e += HT_THREADS) {
            const int mask = cmask[segs_c[e >> 3] & 255];
            if ((mask >> (e & 7)) & 1) items[ibase_c[e >> 3] + __popc(mask & ((1 << (e & 7)) - 1))] = e;
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int e = tid + k * HT_THREADS;
            if (e < cnt * 45) {
                const int q = (int)(((unsigned)e * magic) >> 20), l = e - q * cnt, rc = symtab[q], ra = rc & 15, cb = rc >> 4;
                sdp[l * 81 + ra * 9 + cb] = ld[k], sdp[l * 81 + cb * 9 + ra] = ld[k];
            }
        }
        T* sw = sk; // [CH][3][6]: w[3], dw[3] / dx
        if (tid < cnt * 12) {
            const int q = (int)(((unsigned)tid * magic) >> 20), l = tid - q * cnt;
            sxf[l * 12 + q] = ldx;
            if (q < 3) {
                int base;
                T w[3], dw[3];
                bspline<T>(one_over_dx, ldx, base, w, dw);
#pragma unroll
                for (int k = 0; k < 3; ++k) sw[(l * 3 + q) * 6 + k] = w[k], sw[(l * 3 + q) * 6 + 3 + k] = one_over_dx * dw[k];
            }
        }
        __syncthreads();
        HT_CLK(1);
        // ---- phase B: g = Fn^T grad w for the 27 nodes
        for (int e = tid; e < cnt * 27; e += HT_THREADS) {
            const int l = e / 27, nd = e - l * 27;
            const int i = nd / 9, j = (nd / 3) % 3, k = nd % 3;
            const T* wl = sw + l * 18;
            const T wi = wl[i], dwi = wl[3 + i], wj = wl[6 + j], dwj = wl[9 + j], wk = wl[12 + k], dwk = wl[15 + k];
            const T g0 = dwi * wj * wk, g1 = wi * dwj * wk, g2 = wi * wj * dwk;
            const T* xf = sxf + l * 12;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) sg[e * 3 + cc] = xf[3 + cc * 3] * g0 + xf[3 + cc * 3 + 1] * g1 + xf[3 + cc * 3 + 2] * g2;
        }
        __syncthreads();
        HT_CLK(2);
        // ---- phase C: the last wavefront packs the next chunk into the other table set; K phase: lane = (entry, (a, b)) -> the 3
        // values over q
        if (pl >= 0) pack(cur ^ 1);
        for (int e = tid; e < nent * 9; e += HT_THREADS) {
            const int entry = e / 9, ab = e - entry * 9, a = ab % 3, bb = ab / 3;
            const int info = entinfo[entry], l = info & 255;
            const T* D = sdp + l * 81 + bb;
            const T* gi = sg + (l * 27 + (info >> 8)) * 3;
            const T g0 = gi[0], g1 = gi[1], g2 = gi[2];
#pragma unroll
            for (int q = 0; q < 3; ++q) sk[e * 3 + q] = D[a * 9 + 3 * q] * g0 + D[(a + 3) * 9 + 3 * q] * g1 + D[(a + 6) * 9 + 3 * q] * g2;
        }
        __syncthreads();
        HT_CLK(3);
        // ---- phase D: the next chunk's loads go out, then the pair phase of this one
        fetch(cur ^ 1);
        // ---- pair phase on the matrix cores (A/B build only — measured SLOWER than the scalar version below on MI355X: C2 fp64
        // 19.9 vs 14.4 ms, C3 fp32 64 vs 37 ms.  The chip's FP64 matrix rate equals its FP64 vector rate (78.6 TFLOP/s), and with
        // M = 3 x rows (10 on average) padded to 16 and N = 27 padded to 32 half of every MFMA is padding; what is left is index
        // arithmetic, predicated LDS reads and the same LDS atomics.  Kept as the documented experiment.)  For one cell segment, the blocks of its active tile rows against the 27 column nodes are
        //   Out[(row, a)][j] (for each b) = sum over the segment's particles p and q of K_p[row][a][b][q] g_p[j][q]
        // i.e. for every b a small GEMM with M = 3 x rows (<= 24), N = 27, contraction length 3 x particles: units of 16 x 16 x 4
        // MFMAs (f64: v_mfma_f64_16x16x4_f64), (segment, M tile, b, N tile) dealt round-robin over the 16 waves.  A = K from `sk`,
        // B = g from `sg`, both straight from LDS, one scalar per lane and MFMA; the 4 results per lane go to the LDS tile with
        // the same atomics as before.  (The scalar version — 27 multiply-adds per block and particle on 27 lanes per item — kept
        // only ~370 of the 1024 lanes busy and was bound by their dependent LDS-read / FMA chains: 86 k of a tile's 205 k clocks.)
        if (USE_MFMA) {
            const int w = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
            int g0 = 0;
            for (int s = 0; s < nseg; ++s) {
                const int sd = segs_c[s], cell = sd & 255, l0 = (sd >> 8) & 255, l1 = sd >> 16, mask = cmask[cell];
                const int nrows = __popc(mask), np = l1 - l0, m3 = nrows * 3;
                const int cntu = ((m3 + 15) >> 4) * 6;
                for (int sub = ((w - g0) % 16 + 16) % 16; sub < cntu; sub += 16) {
                    const int mt = sub / 6, rem = sub - mt * 6, bb = rem >> 1, nt = rem & 1;
                    const int mA = li + 16 * mt, rpA = mA / 3, aA = mA - rpA * 3;
                    const bool va = mA < m3;
                    const T* Ap = sk + (sbase_c[s] + (va ? rpA : 0)) * 27 + (aA + 3 * bb) * 3;
                    const int jB = li + 16 * nt;
                    const bool vb = jB < 27;
                    const T* Bp = sg + (l0 * 27 + (vb ? jB : 0)) * 3;
                    typename Mfma16<T>::Acc acc = { 0, 0, 0, 0 };
                    int pp = lk == 3 ? 1 : 0, qq = lk == 3 ? 0 : lk; // k index 4 t + lk = 3 pp + qq
                    const int ksteps = (3 * np + 3) >> 2;
                    for (int t = 0; t < ksteps; ++t) {
                        const bool vk = pp < np;
                        const T av = (va && vk) ? Ap[pp * nrows * 27 + qq] : (T)0;
                        const T bv = (vb && vk) ? Bp[pp * 81 + qq] : (T)0;
                        acc = Mfma16<T>::mac(av, bv, acc);
                        pp += 1, qq += 1; // k += 4
                        if (qq >= 3) qq -= 3, pp += 1;
                    }
                    if (vb) {
                        const int jx = jB / 9, jy = (jB / 3) % 3, jz = jB % 3;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int m = Mfma16<T>::row(lane, reg) + 16 * mt;
                            if (m < m3) {
                                const int rp = m / 3, a = m - rp * 3;
                                int r = 0, seen = 0; // the rp-th active row of the cell
                                for (int bit = 0; bit < 8; ++bit)
                                    if ((mask >> bit) & 1) {
                                        if (seen == rp) r = bit;
                                        ++seen;
                                    }
                                const int ax = (r >> 2) - ((cell >> 4) - 2), ay = ((r >> 1) & 1) - (((cell >> 2) & 3) - 2), az = (r & 1) - ((cell & 3) - 2);
                                lds_atomic_add(tile + r * 1125 + ((ax - jx + 2) * 25 + (ay - jy + 2) * 5 + (az - jz + 2)) * 9 + a + 3 * bb, (AT)acc[reg]);
                            }
                        }
                    }
                }
                g0 += cntu;
            }
        }
        else {
            // ---- pair phase (production: scalar multiply-adds)
            // An item = (segment, row, column triple jg, a) walks its segment's particles with the 3 (column z) x 3 (b) sums in
            // registers; a chunk holds KMAX / (particles per cell) x 27 ~ 460 of them (fp64), so the walk is split over 2 or 4
            // lanes per item while that still fits one pass.  (Tried: all three `a` in one item, 27 K + 9 g values per 81
            // multiply-adds instead of 18 per 27 — 128 registers, spills, C2 15.4 vs 13.6 ms.)
            const int ni = ctl[cur * 8] * 27;
            const int split = 4 * ni <= HT_THREADS ? 4 : (2 * ni <= HT_THREADS ? 2 : 1);
            for (int it0 = tid; it0 < ni * split; it0 += HT_THREADS) {
                const int part = (it0 >= ni) + (it0 >= 2 * ni) + (it0 >= 3 * ni), it = it0 - part * ni;
                const int e = items[it / 27], rem = it % 27, jg = rem / 3, a = rem % 3, s = e >> 3, r = e & 7;
                const int sd = segs_c[s], cell = sd & 255, mask = cmask[cell];
                int l0 = (sd >> 8) & 255, l1 = sd >> 16;
                if (split > 1) {
                    const int per = (l1 - l0 + split - 1) / split;
                    l0 += part * per, l1 = min(l1, l0 + per);
                    if (l0 >= l1) continue;
                }
                const int nrows = __popc(mask), rowpos = __popc(mask & ((1 << r) - 1));
                const int ax = (r >> 2) - ((cell >> 4) - 2), ay = ((r >> 1) & 1) - (((cell >> 2) & 3) - 2), az = (r & 1) - ((cell & 3) - 2);
                const int jx = jg / 3, jy = jg % 3;
                T acc[3][3]; // [column z][b]
    #pragma unroll
                for (int z = 0; z < 3; ++z)
    #pragma unroll
                    for (int q = 0; q < 3; ++q) acc[z][q] = (T)0;
                const T* Kp = sk + (sbase_c[s] + (l0 - ((sd >> 8) & 255)) * nrows + rowpos) * 27 + a * 3;
                const T* gp = sg + (l0 * 27 + jg * 3) * 3;
    #pragma unroll 2
                for (int l = l0; l < l1; ++l, Kp += nrows * 27, gp += 81) {
                    T g[9]; // g_j[q] of the columns j = (jx, jy, z): g[3 z + q]
    #pragma unroll
                    for (int q = 0; q < 9; ++q) g[q] = gp[q];
    #pragma unroll
                    for (int bb = 0; bb < 3; ++bb)
    #pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const T k = Kp[bb * 9 + q];
    #pragma unroll
                            for (int z = 0; z < 3; ++z) acc[z][bb] += k * g[3 * z + q];
                        }
                }
                AT* o = tile + r * 1125 + ((ax - jx + 2) * 25 + (ay - jy + 2) * 5 + (az + 2)) * 9 + a;
    #pragma unroll
                for (int z = 0; z < 3; ++z)
    #pragma unroll
                    for (int bb = 0; bb < 3; ++bb) lds_atomic_add(o - z * 9 + 3 * bb, (AT)acc[z][bb]);
            }
        }
        __syncthreads();
        HT_CLK(4);
    }
    for (int e = tid; e < 8 * 1125; e += HT_THREADS) {
        int r = e / 1125, q = e - r * 1125;
        int dof = rdof[r];
        if (dof < 0) continue;
        T v = (T)tile[e];
        if (q >= 62 * 9 && q < 63 * 9 && ((q - 62 * 9) % 4 == 0) && (!own || own[dof])) v += mass[dof];
        val[(int64_t)dof * 1125 + q] = v;
        if (written && q == 0) written[dof] = 1;
    }
#ifdef HOT_HT_CLOCKS
    HT_CLK(5);
    if (tid == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&ht_clk[i], clk_[i]);
#endif
}

template <class T>
void Ctx<T>::assemble_tiles(Level<T>& L)
{
    pDP.reserve(45 * (size_t)Np);
    HOT_LAUNCH(this, "hessian_dpdf", k_dpdf45<T>, div_up(Np, 256), 256, 0, pFt.p, pVol.p, pMu.p, pLam.p, pDP.p, Np, dt, cfg.project);
    if (!attr_tiles_set) {
#ifdef HOT_AB_KERNELS
        HOT_HIP(hipFuncSetAttribute((const void*)k_hessian_tiles<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TileLds<T>::bytes));
#endif
        HOT_HIP(hipFuncSetAttribute((const void*)k_hessian_tiles2<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TileLds2<T>::bytes));
#ifdef HOT_AB_KERNELS
        HOT_HIP(hipFuncSetAttribute((const void*)k_hessian_tiles2<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TileLds2<T>::bytes));
#endif
        attr_tiles_set = true;
    }
    constexpr int TPB = (G::BX / 2) * (G::BY / 2) * (G::BZ / 2);
#ifdef HOT_AB_KERNELS
    if (ab_flag("HOT_HESSIAN_TILES_V1")) { // 81 multiply-adds per (particle, row, column)
        HOT_LAUNCH(this, "hessian_assemble", k_hessian_tiles<T>, 256 * div_up(Nb * TPB, 256), HT_THREADS, TileLds<T>::bytes, pX.p, pFn.p, pDP.p, Np, blocks.p, gIdx.p, cell_first.p, cell_map, mass.p, L.val.p, (T)1 / dx, Nb * TPB);
        return;
    }
#endif
    if (L.part) {
        written.reserve(Nn);
        HOT_HIP(hipMemsetAsync(written.p, 0, Nn, stream));
    }
#ifdef HOT_AB_KERNELS
    if (ab_flag("HOT_HESSIAN_MFMA")) {
        HOT_LAUNCH(this, "hessian_assemble", (k_hessian_tiles2<T, true>), 256 * div_up(Nb * TPB, 256), HT_THREADS, TileLds2<T>::bytes, pX.p, pFn.p, pDP.p, Np, blocks.p, gIdx.p, cell_first.p, cell_map, mass.p, L.val.p, (T)1 / dx, Nb * TPB,
            L.mask(), L.part ? written.p : (uint8_t*)nullptr);
        return;
    }
#endif
    HOT_LAUNCH(this, "hessian_assemble", (k_hessian_tiles2<T, false>), 256 * div_up(Nb * TPB, 256), HT_THREADS, TileLds2<T>::bytes, pX.p, pFn.p, pDP.p, Np, blocks.p, gIdx.p, cell_first.p, cell_map, mass.p, L.val.p, (T)1 / dx, Nb * TPB,
        L.mask(), L.part ? written.p : (uint8_t*)nullptr);
#ifdef HOT_HT_CLOCKS
    unsigned long long h[8] = {};
    HOT_HIP(hipStreamSynchronize(stream));
    HOT_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(ht_clk), sizeof(h)));
    const double tiles = (double)Nb * TPB;
    fprintf(stderr, "hessian tile clocks per workgroup: prologue %.0f A %.0f B %.0f C %.0f D %.0f write-out %.0f\n", h[0] / tiles, h[1] / tiles, h[2] / tiles, h[3] / tiles, h[4] / tiles, h[5] / tiles);
    memset(h, 0, sizeof(h));
    HOT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ht_clk), h, sizeof(h)));
#endif
}

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
