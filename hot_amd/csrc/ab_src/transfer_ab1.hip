// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../transfer.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
template <class T, bool WITH_CN>
__global__ __launch_bounds__(256) void k_p2g(const T* __restrict__ X, const T* __restrict__ V, const T* __restrict__ M, const T* __restrict__ C,
    const T* __restrict__ Mu, const T* __restrict__ Lam, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin,
    const int32_t* __restrict__ group_nb, T* __restrict__ part, T dx, T one_over_dx)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    constexpr int NQ = WITH_CN ? 5 : 4;
    __shared__ T acc[NQ][TILE];
    __shared__ int32_t nb8[8];
    const int g = blockIdx.x;
    for (int t = threadIdx.x; t < NQ * TILE; t += 256) (&acc[0][0])[t] = (T)0;
    if (threadIdx.x < 8) nb8[threadIdx.x] = group_nb[g * 8 + threadIdx.x];
    __syncthreads();
    const int first = group_first[g], last = group_first[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    for (int p = first + threadIdx.x; p < last; p += 256) {
        T xp[3] = { X[p], X[Np + p], X[2 * Np + p] };
        T m = M[p];
        T mom[3] = { m * V[p], m * V[Np + p], m * V[2 * Np + p] };
        T Cm[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) Cm[c] = m * C[(int64_t)c * Np + p];
        int base[3];
        T w[3][3], dw[3][3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xp[d], base[d], w[d], dw[d]);
        T cn = (T)0;
        if (WITH_CN) {
            // |dP/dF(F = I)|_F of the fixed-corotated model: A = 2 mu I + lambda 11^T, B blocks = mu [[1,1],[1,1]]
            // (already PSD, so --project does not change it): sqrt(3(2mu+l)^2 + 6 l^2 + 12 mu^2)
            T mu = Mu[p], la = Lam[p];
            cn = m * hsqrt((T)3 * ((T)2 * mu + la) * ((T)2 * mu + la) + (T)6 * la * la + (T)12 * mu * mu);
        }
        const int cx = base[0] - ox, cy = base[1] - oy, cz = base[2] - oz;
        // rotate the visiting order per lane so that the particles of one cell (adjacent lanes) hit different
        // LDS addresses in the same instruction
        int rot = threadIdx.x % 27;
        for (int n = 0; n < 27; ++n) {
            int q = n + rot;
            q = q >= 27 ? q - 27 : q;
            int i = q / 9, j = (q / 3) % 3, k = q % 3;
            T wijk = w[0][i] * w[1][j] * w[2][k];
            T d0 = (T)(base[0] + i) * dx - xp[0], d1 = (T)(base[1] + j) * dx - xp[1], d2 = (T)(base[2] + k) * dx - xp[2];
            int t = ((cx + i) * TY + (cy + j)) * TZ + (cz + k);
            lds_atomic_add(&acc[0][t], m * wijk);
            lds_atomic_add(&acc[1][t], (Cm[0] * d0 + Cm[3] * d1 + Cm[6] * d2 + mom[0]) * wijk);
            lds_atomic_add(&acc[2][t], (Cm[1] * d0 + Cm[4] * d1 + Cm[7] * d2 + mom[1]) * wijk);
            lds_atomic_add(&acc[3][t], (Cm[2] * d0 + Cm[5] * d1 + Cm[8] * d2 + mom[2]) * wijk);
            if (WITH_CN) lds_atomic_add(&acc[NQ - 1][t], cn * wijk);
        }
    }
    __syncthreads();
    // partial tile of this group, coalesced; summed per node by k_tile_reduce (no global atomics)
    T* out = part + (int64_t)g * NQ * TILE;
    for (int t = threadIdx.x; t < NQ * TILE; t += 256) out[t] = (&acc[0][0])[t];
}

