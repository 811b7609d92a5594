// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../mg_solve.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
// One colour of one half-sweep of symmetric block GS.  FWD: h_i = Dinv (rhs_i - sum_{j<i} A_ij h_j), also writes
// hD_i = D_i h_i ; BWD: du_i = Dinv (rhs_i - sum_{j>i} A_ij du_j).  "<" is the packed (colour, block, index) key.
template <class T, bool FWD>
__global__ __launch_bounds__(64) void k_gs_color(const int32_t* __restrict__ col, const T* __restrict__ val, const uint32_t* __restrict__ ckey, const int32_t* __restrict__ gs_order,
    const int32_t* __restrict__ block_start, const T* __restrict__ diagVal, const T* __restrict__ diagBlockInv, const T* __restrict__ rhs, T* x, T* hD, int block0, int nblk)
{
    __shared__ T xl[64][3];
    const int lane = threadIdx.x;
    const int b = block0 + blockIdx.x;
    if (blockIdx.x >= nblk) return;
    const int start = block_start[b], cnt = block_start[b + 1] - start;
    for (int s = 0; s < cnt; ++s) {
        const int ii = FWD ? s : cnt - 1 - s;
        const int i = gs_order[start + ii];
        const uint32_t keyi = ckey[i];
        const int32_t* c = col + (int64_t)i * 125;
        const T* v = val + (int64_t)i * 1125;
        T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int k = lane + 64 * r;
            if (k < 125) {
                int j = c[k];
                uint32_t keyj = ckey[j];
                bool take = FWD ? (keyj < keyi) : (keyj > keyi);
                if (take) {
                    T x0, x1, x2;
                    if ((keyj >> 7) == (keyi >> 7)) {
                        int lj = (int)(keyj & 127u) - 1;
                        x0 = xl[lj][0], x1 = xl[lj][1], x2 = xl[lj][2];
                    }
                    else {
                        x0 = x[3 * (int64_t)j], x1 = x[3 * (int64_t)j + 1], x2 = x[3 * (int64_t)j + 2];
                    }
                    const T* bb = v + k * 9;
                    s0 += bb[0] * x0 + bb[3] * x1 + bb[6] * x2;
                    s1 += bb[1] * x0 + bb[4] * x1 + bb[7] * x2;
                    s2 += bb[2] * x0 + bb[5] * x1 + bb[8] * x2;
                }
            }
        }
        s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
        T r0 = rhs[3 * (int64_t)i] - s0, r1 = rhs[3 * (int64_t)i + 1] - s1, r2 = rhs[3 * (int64_t)i + 2] - s2;
        const T* di = diagBlockInv + 9 * (int64_t)i;
        T h0 = di[0] * r0 + di[3] * r1 + di[6] * r2, h1 = di[1] * r0 + di[4] * r1 + di[7] * r2, h2 = di[2] * r0 + di[5] * r1 + di[8] * r2;
        if (lane == 0) {
            xl[ii][0] = h0, xl[ii][1] = h1, xl[ii][2] = h2;
            x[3 * (int64_t)i] = h0, x[3 * (int64_t)i + 1] = h1, x[3 * (int64_t)i + 2] = h2;
            if (FWD) {
                const T* d = diagVal + 9 * (int64_t)i;
                hD[3 * (int64_t)i] = d[0] * h0 + d[3] * h1 + d[6] * h2;
                hD[3 * (int64_t)i + 1] = d[1] * h0 + d[4] * h1 + d[7] * h2;
                hD[3 * (int64_t)i + 2] = d[2] * h0 + d[5] * h1 + d[8] * h2;
            }
        }
        __syncthreads(); // single-wave workgroup: orders the LDS write before the next node's reads
    }
}

