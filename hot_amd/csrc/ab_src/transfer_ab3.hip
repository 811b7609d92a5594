// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../transfer.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
// Rounds 2 - 5 (A/B build since round 6, HOT_P2G_CELLS2): the (cell segment, node row, half) work items of k_p2g_cells (above, A/B build), but only the 16 (17 with the
// CN quantity) per-particle scalars x, m, m v, m C are staged — the nine 1-D weights are recomputed per item from x (a few
// multiply-adds against nine LDS reads) and the base cell comes from the segment's first particle.  35 KB instead of 56 KB per
// 256-particle chunk: four 256-thread workgroups per CU instead of two 512-thread ones, i.e. twice as many independent
// header -> staging -> items chains in flight per CU.  (Tried in round 3: persistent workgroups walking the groups with the next
// unit's 16 scalars requested into registers before the item phase of the current one — C2 0.156 vs 0.132 ms, C3 0.405 vs 0.375:
// with four workgroups per CU the staging latency is already covered by the other three, what is left is the item phase itself,
// VALU 44 % + LDS 56 % of the kernel's cycles (profiles/r03_sq_counters_C2.json).)
// Development aid (-DHOT_HT_CLOCKS, tools/hess_phases.sh): shader clocks of thread 0 between the barriers of k_p2g_cells2, summed
// over the workgroups: 0 header + zeroing, 1 staging, 2 items, 3 write-out.
#ifndef HOT_P2G_NO_ITEMS
#define HOT_P2G_NO_ITEMS 0
#endif
#ifdef HOT_HT_CLOCKS
extern __device__ unsigned long long p2g_clk[12];
#define P2G_CLK(i) \
    do { \
        if (tid == 0) { \
            const unsigned long long t_ = clock64(); \
            clk_[i] += t_ - t0_, t0_ = t_; \
        } \
    } while (0)
#else
#define P2G_CLK(i)
#endif

template <class T, bool WITH_CN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_p2g_cells2(const T* __restrict__ X, const T* __restrict__ V, const T* __restrict__ M, const T* __restrict__ C,
    const T* __restrict__ Mu, const T* __restrict__ Lam, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin,
    const int32_t* __restrict__ group_cell0, const int32_t* __restrict__ cell_first, T* __restrict__ part, T dx, T one_over_dx)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    constexpr int NQ = WITH_CN ? 5 : 4, NS = 16 + (WITH_CN ? 1 : 0), THREADS = 256, CH = sizeof(T) == 4 ? 512 : 256;
    using AT = AccT<T>;
    __shared__ AT acc[NQ][TILE];
    __shared__ T sp[NS][CH]; // x(3) m(1) m*v(3) m*C(9) [cn]
    __shared__ int32_t segs[G::EPB + 2];
    __shared__ int32_t nseg;
    const int g = blockIdx.x, tid = threadIdx.x;
#ifdef HOT_HT_CLOCKS
    unsigned long long clk_[4] = { 0, 0, 0, 0 }, t0_ = clock64();
#endif
    for (int t = tid; t < NQ * TILE; t += THREADS) (&acc[0][0])[t] = (AT)0;
    const int first = group_first[g], last = group_first[g + 1];
    const int c0 = group_cell0[g], c1 = group_cell0[g + 1];
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    for (int ch = first; ch < last; ch += CH) {
        if (tid == 0) nseg = 0;
        __syncthreads();
        P2G_CLK(0);
        for (int l = tid; l < (HOT_P2G_NO_ITEMS == 3 ? 0 : CH) && ch + l < last; l += THREADS) { // (3: experiment, no particle loads)
            const int p = ch + l;
            const T m = M[p];
#pragma unroll
            for (int d = 0; d < 3; ++d) sp[d][l] = X[(int64_t)d * Np + p], sp[4 + d][l] = m * V[(int64_t)d * Np + p];
            sp[3][l] = m;
#pragma unroll
            for (int c = 0; c < 9; ++c) sp[7 + c][l] = m * C[(int64_t)c * Np + p];
            if (WITH_CN) {
                const T mu = Mu[p], la = Lam[p];
                sp[NS - 1][l] = m * hsqrt((T)3 * ((T)2 * mu + la) * ((T)2 * mu + la) + (T)6 * la * la + (T)12 * mu * mu);
            }
        }
        for (int c = c0 + tid; c < c1; c += THREADS) {
            const int s0 = max(cell_first[c], ch), s1 = min(cell_first[c + 1], min(ch + CH, last));
            if (s1 > s0) segs[atomicAdd(&nseg, 1)] = (s0 - ch) | ((s1 - ch) << 16);
        }
        __syncthreads();
        P2G_CLK(1);
        // items: (cell segment, node row j, half of the segment) -> the 9 nodes (i, k) of that row with their sums in registers: the
        // scalars of a particle are read from LDS once per 9 nodes instead of once per 3 (the item phase is LDS 56 % + VALU 44 % of the
        // round-2 kernel's cycles).  fp32: all NQ quantities in one item (C3 0.302 from 0.375 ms).  fp64: 45 sums need 216 registers
        // (two workgroups per CU instead of four: 0.141 ms against the 3-node items' 0.132), so the quantities are dealt to two items,
        // {m, m v0, cn} and {m v1, m v2}.
        auto run9 = [&](auto mask_c, int it) {
            constexpr int MASK = decltype(mask_c)::value;
            const int sd = segs[it / 6], j = (it % 6) >> 1, hf = it & 1, s0 = sd & 0xffff, s1 = sd >> 16;
            const int mid = (s0 + s1 + 1) >> 1, l0 = hf ? mid : s0, l1 = hf ? s1 : mid;
            if (l0 >= l1) return;
            T a[3][3][NQ]; // [i][k][quantity]
#pragma unroll
            for (int e = 0; e < 9 * NQ; ++e) (&a[0][0][0])[e] = (T)0;
            // the base cell is the same for every particle of the segment
            const int b0 = base_node_of<T>(one_over_dx, sp[0][l0]), b1 = base_node_of<T>(one_over_dx, sp[1][l0]), b2 = base_node_of<T>(one_over_dx, sp[2][l0]);
            const T fb0 = (T)b0, fb1 = (T)b1, fb2 = (T)b2;
            for (int l = l0; l < l1; ++l) {
                const T x0 = sp[0][l], x1 = sp[1][l], x2 = sp[2][l];
                // 1-D quadratic B-spline weights, the arithmetic of bspline() (BSplines.h:55-81)
                auto w3 = [&](T x, T fb, T(&w)[3]) {
                    const T d0 = fma(one_over_dx, x, -fb); // exact product, like the fused multiply-add a -O3 -march=native host build makes of it (hot_common.h bspline)
                    const T z = (T)1.5 - d0, d1 = d0 - (T)1, zz = (T)1.5 - ((T)1 - d1);
                    w[0] = (T)0.5 * z * z, w[1] = (T)0.75 - d1 * d1, w[2] = (T)0.5 * zz * zz;
                };
                T wi[3], wj3[3], wk[3];
                w3(x0, fb0, wi), w3(x1, fb1, wj3), w3(x2, fb2, wk);
                const T wj = j == 0 ? wj3[0] : (j == 1 ? wj3[1] : wj3[2]);
                const T d1 = (T)(b1 + j) * dx - x1;
                T m = (T)0, cn = (T)0, u[3], cc[3], ee[3];
                if constexpr ((MASK & 1) != 0) m = sp[3][l];
                if constexpr (WITH_CN && (MASK & (1 << (NQ - 1))) != 0) cn = sp[NS - 1][l];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (MASK & (2 << q)) u[q] = sp[10 + q][l] * d1 + sp[4 + q][l], cc[q] = sp[7 + q][l], ee[q] = sp[13 + q][l];
                T d0[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) d0[i] = (T)(b0 + i) * dx - x0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T d2 = (T)(b2 + k) * dx - x2, wjk = wj * wk[k];
                    T t[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (MASK & (2 << q)) t[q] = ee[q] * d2 + u[q];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const T wijk = wi[i] * wjk;
                        if constexpr ((MASK & 1) != 0) a[i][k][0] += m * wijk;
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            if (MASK & (2 << q)) a[i][k][1 + q] += (cc[q] * d0[i] + t[q]) * wijk;
                        if constexpr (WITH_CN && (MASK & (1 << (NQ - 1))) != 0) a[i][k][NQ - 1] += cn * wijk;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int t = ((b0 - ox + i) * TY + (b1 - oy + j)) * TZ + (b2 - oz + k);
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (MASK & (1 << q)) lds_atomic_add(&acc[q][t], (AT)a[i][k][q]);
                }
        };
        const int n6 = HOT_P2G_NO_ITEMS ? 0 : nseg * 6; // (HOT_P2G_NO_ITEMS: experiment, staging alone)
        if constexpr (sizeof(T) == 4) {
            for (int it = tid; it < n6; it += THREADS) run9(std::integral_constant<int, (1 << NQ) - 1>{}, it);
        }
        else {
            for (int it = tid; it < 2 * n6; it += THREADS) {
                if (it < n6)
                    run9(std::integral_constant<int, 1 | 2 | (WITH_CN ? 16 : 0)>{}, it);
                else
                    run9(std::integral_constant<int, 4 | 8>{}, it - n6);
            }
        }
    }
    __syncthreads();
    P2G_CLK(2);
    T* out = part + (int64_t)g * NQ * TILE;
#if HOT_P2G_NO_ITEMS != 2 // (2: experiment, no write-out either)
    for (int t = tid; t < NQ * TILE; t += THREADS) out[t] = (T)(&acc[0][0])[t];
#endif
#ifdef HOT_HT_CLOCKS
    P2G_CLK(3);
    if (tid == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&p2g_clk[i], clk_[i]);
#endif
}
