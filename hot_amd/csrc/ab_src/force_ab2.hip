// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../force.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
// Round-2 production force scatter (A/B build only now): same (cell, node column) work items as k_p2g_cells (transfer.hip) — the particles of a
// base cell share their 27 nodes, so the three nodes of a column are summed in registers over the cell and added to the
// LDS tile once.  The 1-D weights and their derivatives are computed once per particle while staging (the first
// version, k_force_scatter above, recomputed them per node through rotated tables that ended up in scratch memory).
constexpr int FORCE_THREADS = 512, FORCE_CHUNK = 256; // see P2G_THREADS (transfer.hip)

template <class T>
__global__ __launch_bounds__(FORCE_THREADS) void k_force_cells(const T* __restrict__ X, const T* __restrict__ stress, int64_t Np, const int32_t* __restrict__ group_first,
    const int32_t* __restrict__ group_origin, const int32_t* __restrict__ group_cell0, const int32_t* __restrict__ cell_first, T* __restrict__ part, T one_over_dx, T scale)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    constexpr int CH = sizeof(T) == 4 ? 2 * FORCE_CHUNK : FORCE_CHUNK; // fp32 groups (4x4x4 cells) hold twice the particles: same LDS bytes, one staging round
    // LDS float atomics (ds_add_f32) retire ~40x slower than the double ones on this chip (measured: 34 k of 43 k clocks of a
    // workgroup's item phase; with a double tile 0.4 k): the fp32 build accumulates its tile in double as well
    using AT = AccT<T>;
    __shared__ AT acc[3][TILE];
    __shared__ T sp[27][CH]; // S(9) w(3x3) dw(3x3)
    __shared__ int32_t sbase[3][CH];
    __shared__ int32_t segs[G::EPB + 2];
    __shared__ int32_t nseg;
    const int g = blockIdx.x, tid = threadIdx.x;
    for (int t = tid; t < 3 * TILE; t += FORCE_THREADS) (&acc[0][0])[t] = (AT)0;
    const int first = group_first[g], last = group_first[g + 1];
    const int c0 = group_cell0[g], c1 = group_cell0[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    for (int ch = first; ch < last; ch += CH) {
        if (tid == 0) nseg = 0;
        __syncthreads(); // also orders the previous chunk's reads of sp / segs before they are overwritten
        const int p = ch + tid;
        if (tid < CH && p < last) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int base;
                T w[3], dw[3];
                bspline<T>(one_over_dx, X[(int64_t)d * Np + p], base, w, dw);
                sbase[d][tid] = base;
#pragma unroll
                for (int q = 0; q < 3; ++q) sp[9 + 3 * d + q][tid] = w[q], sp[18 + 3 * d + q][tid] = dw[q];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) sp[c][tid] = scale * stress[(int64_t)c * Np + p];
        }
        for (int c = c0 + tid; c < c1; c += FORCE_THREADS) {
            const int s0 = max(cell_first[c], ch), s1 = min(cell_first[c + 1], min(ch + CH, last));
            if (s1 > s0) segs[atomicAdd(&nseg, 1)] = (s0 - ch) | ((s1 - ch) << 16);
        }
        __syncthreads();
        const int ni = nseg * 18; // (segment, node column, half of the segment), see k_p2g_cells
        for (int it = tid; it < ni; it += FORCE_THREADS) {
            const int sd = segs[it / 18], jk = (it % 18) >> 1, hf = it & 1, s0 = sd & 0xffff, s1 = sd >> 16;
            const int mid = (s0 + s1 + 1) >> 1, l0 = hf ? mid : s0, l1 = hf ? s1 : mid;
            if (l0 >= l1) continue;
            const int j = jk / 3, k = jk - 3 * j;
            T a[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) a[i][0] = a[i][1] = a[i][2] = (T)0;
            for (int l = l0; l < l1; ++l) {
                const T wy = sp[12 + j][l], wz = sp[15 + k][l], dwy = sp[21 + j][l], dwz = sp[24 + k][l];
                T S[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) S[c] = sp[c][l];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const T wi = sp[9 + i][l], dwi = one_over_dx * sp[18 + i][l];
                    const T wij = wi * wy, dwij_i = dwi * wy, dwij_j = wi * one_over_dx * dwy;
                    const T g0 = dwij_i * wz, g1 = dwij_j * wz, g2 = wij * one_over_dx * dwz;
                    a[i][0] += -(S[0] * g0 + S[3] * g1 + S[6] * g2);
                    a[i][1] += -(S[1] * g0 + S[4] * g1 + S[7] * g2);
                    a[i][2] += -(S[2] * g0 + S[5] * g1 + S[8] * g2);
                }
            }
            const int b0 = sbase[0][l0], b1 = sbase[1][l0], b2 = sbase[2][l0]; // the same for every particle of the cell
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int t = ((b0 - ox + i) * TY + (b1 - oy + j)) * TZ + (b2 - oz + k);
                lds_atomic_add(&acc[0][t], (AT)a[i][0]), lds_atomic_add(&acc[1][t], (AT)a[i][1]), lds_atomic_add(&acc[2][t], (AT)a[i][2]);
            }
        }
    }
    __syncthreads();
    T* out = part + (int64_t)g * 3 * TILE; // partial tile, summed per node by k_tile_reduce
    for (int t = tid; t < 3 * TILE; t += FORCE_THREADS) out[t] = (T)(&acc[0][0])[t];
}
