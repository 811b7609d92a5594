// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../hessian_tiles.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
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

