// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../force.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
// pass B: rasterizeForceToTVStack — f_i -= dt * stress grad w_i, LDS accumulators, one global atomic per touched node
template <class T>
__global__ __launch_bounds__(256) void k_force_scatter(const T* __restrict__ X, const T* __restrict__ stress, int64_t Np, const int32_t* __restrict__ group_first,
    const int32_t* __restrict__ group_origin, const int32_t* __restrict__ group_nb, T* __restrict__ part, T one_over_dx, T scale)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    __shared__ T acc[3][TILE];
    __shared__ int32_t nb8[8];
    const int g = blockIdx.x;
    if (threadIdx.x < 8) nb8[threadIdx.x] = group_nb[g * 8 + threadIdx.x];
    for (int t = threadIdx.x; t < TILE; t += 256) acc[0][t] = acc[1][t] = acc[2][t] = (T)0;
    __syncthreads();
    const int first = group_first[g], last = group_first[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    const int rk = threadIdx.x % 3, rj = (threadIdx.x / 3) % 3;
    for (int p = first + threadIdx.x; p < last; p += 256) {
        T xp[3] = { X[p], X[Np + p], X[2 * Np + p] };
        int base[3];
        T w[3][3], dw[3][3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xp[d], base[d], w[d], dw[d]);
        T S[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) S[c] = scale * stress[(int64_t)c * Np + p];
        // rotated y / z weight tables and tile offsets
        T wy[3], dwy[3], wz[3], dwz[3];
        rot3(w[1], rj, wy), rot3(dw[1], rj, dwy), rot3(w[2], rk, wz), rot3(dw[2], rk, dwz);
        const int cx = base[0] - ox, cy = base[1] - oy, cz = base[2] - oz;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T wi = w[0][i], dwi = one_over_dx * dw[0][i];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int jj = j + rj;
                jj = jj >= 3 ? jj - 3 : jj;
                T wij = wi * wy[j], dwij_i = dwi * wy[j], dwij_j = wi * one_over_dx * dwy[j];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    int kk = k + rk;
                    kk = kk >= 3 ? kk - 3 : kk;
                    T g0 = dwij_i * wz[k], g1 = dwij_j * wz[k], g2 = wij * one_over_dx * dwz[k];
                    int t = ((cx + i) * TY + (cy + jj)) * TZ + (cz + kk);
                    lds_atomic_add(&acc[0][t], -(S[0] * g0 + S[3] * g1 + S[6] * g2));
                    lds_atomic_add(&acc[1][t], -(S[1] * g0 + S[4] * g1 + S[7] * g2));
                    lds_atomic_add(&acc[2][t], -(S[2] * g0 + S[5] * g1 + S[8] * g2));
                }
            }
        }
    }
    __syncthreads();
    T* out = part + (int64_t)g * 3 * TILE; // partial tile, summed per node by k_tile_reduce
    for (int t = threadIdx.x; t < 3 * TILE; t += 256) out[t] = (&acc[0][0])[t];
}

