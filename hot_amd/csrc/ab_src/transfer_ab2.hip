// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../transfer.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
// Round-1 production P2G (A/B build only now).  The particles of one base cell share their 27 support nodes, so a (cell, node column) work item sums
// the contributions of the whole cell to its 3 nodes in registers and touches the LDS accumulator once per node and
// quantity: ~8x fewer ds_add_f64 than one-add-per-particle (k_p2g above, kept for A/B).  Particle data and the per-
// particle 1-D weights are staged in LDS (coalesced loads, weights computed once per particle instead of once per
// node) and read back with wave-broadcast reads (all lanes of a cell read the same particle).
// 256 threads measured best (320 = one item round per fp64 page, but lower occupancy: P2G 0.27 vs 0.24 ms at 2 M particles)
constexpr int P2G_THREADS = 512, P2G_CHUNK = 256; // particles are staged 256 at a time by the first 256 threads; all 512 work on the items

template <class T, bool WITH_CN>
__global__ __launch_bounds__(P2G_THREADS) void k_p2g_cells(const T* __restrict__ X, const T* __restrict__ V, const T* __restrict__ M, const T* __restrict__ C,
    const T* __restrict__ Mu, const T* __restrict__ Lam, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin,
    const int32_t* __restrict__ group_cell0, const int32_t* __restrict__ cell_first, T* __restrict__ part, T dx, T one_over_dx)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    constexpr int NQ = WITH_CN ? 5 : 4, NS = 25 + (WITH_CN ? 1 : 0), CH = sizeof(T) == 4 ? 2 * P2G_CHUNK : P2G_CHUNK; // fp32 groups hold twice the particles: same LDS bytes, one staging round
    using AT = AccT<T>; // double also in the fp32 build: LDS float atomics are ~40x slower (see k_force_cells)
    __shared__ AT acc[NQ][TILE];
    __shared__ T sp[NS][CH]; // x(3) m(1) m*v(3) m*C(9) w(3x3) [cn]
    __shared__ int32_t sbase[3][CH];
    __shared__ int32_t segs[G::EPB + 2];
    __shared__ int32_t nseg;
    const int g = blockIdx.x, tid = threadIdx.x;
    for (int t = tid; t < NQ * TILE; t += P2G_THREADS) (&acc[0][0])[t] = (AT)0;
    const int first = group_first[g], last = group_first[g + 1];
    const int c0 = group_cell0[g], c1 = group_cell0[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    for (int ch = first; ch < last; ch += CH) {
        if (tid == 0) nseg = 0;
        __syncthreads(); // also orders the previous chunk's reads of sp / segs before they are overwritten
        const int p = ch + tid;
        if (tid < CH && p < last) {
            const T m = M[p];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const T x = X[(int64_t)d * Np + p];
                int base;
                T w[3], dw[3];
                bspline<T>(one_over_dx, x, base, w, dw);
                sp[d][tid] = x, sp[4 + d][tid] = m * V[(int64_t)d * Np + p], sbase[d][tid] = base;
                sp[16 + 3 * d][tid] = w[0], sp[17 + 3 * d][tid] = w[1], sp[18 + 3 * d][tid] = w[2];
            }
            sp[3][tid] = m;
#pragma unroll
            for (int c = 0; c < 9; ++c) sp[7 + c][tid] = m * C[(int64_t)c * Np + p];
            if (WITH_CN) {
                // |dP/dF(F = I)|_F of the fixed-corotated model: A = 2 mu I + lambda 11^T, B blocks = mu [[1,1],[1,1]]
                // (already PSD, so --project does not change it): sqrt(3(2mu+l)^2 + 6 l^2 + 12 mu^2)
                const T mu = Mu[p], la = Lam[p];
                sp[NS - 1][tid] = m * hsqrt((T)3 * ((T)2 * mu + la) * ((T)2 * mu + la) + (T)6 * la * la + (T)12 * mu * mu);
            }
        }
        for (int c = c0 + tid; c < c1; c += P2G_THREADS) {
            const int s0 = max(cell_first[c], ch), s1 = min(cell_first[c + 1], min(ch + CH, last));
            if (s1 > s0) segs[atomicAdd(&nseg, 1)] = (s0 - ch) | ((s1 - ch) << 16);
        }
        __syncthreads();
        // items: (cell segment, node column, half of the segment).  A full fp64 page has 32 cells x 9 columns = 288
        // columns, which would be one full round of the 256 threads plus a nearly empty one; splitting every segment in
        // two gives 576 half-length items = 2.25 short rounds.
        const int ni = nseg * 18;
        for (int it = tid; it < ni; it += P2G_THREADS) {
            const int sd = segs[it / 18], jk = (it % 18) >> 1, hf = it & 1, s0 = sd & 0xffff, s1 = sd >> 16;
            const int mid = (s0 + s1 + 1) >> 1, l0 = hf ? mid : s0, l1 = hf ? s1 : mid;
            if (l0 >= l1) continue;
            const int j = jk / 3, k = jk - 3 * j;
            T a[3][NQ];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int q = 0; q < NQ; ++q) a[i][q] = (T)0;
            const int b0 = sbase[0][l0], b1 = sbase[1][l0], b2 = sbase[2][l0]; // the same for every particle of the cell
            for (int l = l0; l < l1; ++l) {
                const T d1 = (T)(b1 + j) * dx - sp[1][l], d2 = (T)(b2 + k) * dx - sp[2][l];
                const T m = sp[3][l];
                const T c0_ = sp[7][l], c1_ = sp[8][l], c2_ = sp[9][l], m0 = sp[4][l], m1 = sp[5][l], m2 = sp[6][l], xp0 = sp[0][l];
                T cn = (T)0;
                if (WITH_CN) cn = sp[NS - 1][l];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const T wijk = sp[16 + i][l] * sp[19 + j][l] * sp[22 + k][l];
                    const T d0 = (T)(b0 + i) * dx - xp0;
                    a[i][0] += m * wijk;
                    a[i][1] += (c0_ * d0 + sp[10][l] * d1 + sp[13][l] * d2 + m0) * wijk;
                    a[i][2] += (c1_ * d0 + sp[11][l] * d1 + sp[14][l] * d2 + m1) * wijk;
                    a[i][3] += (c2_ * d0 + sp[12][l] * d1 + sp[15][l] * d2 + m2) * wijk;
                    if (WITH_CN) a[i][NQ - 1] += cn * wijk;
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int t = ((b0 - ox + i) * TY + (b1 - oy + j)) * TZ + (b2 - oz + k);
#pragma unroll
                for (int q = 0; q < NQ; ++q) lds_atomic_add(&acc[q][t], (AT)a[i][q]);
            }
        }
    }
    __syncthreads();
    // partial tile of this group, coalesced; summed per node by k_tile_reduce (no global atomics)
    T* out = part + (int64_t)g * NQ * TILE;
    for (int t = tid; t < NQ * TILE; t += P2G_THREADS) out[t] = (T)(&acc[0][0])[t];
}
