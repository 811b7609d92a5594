// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../hessian.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
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

