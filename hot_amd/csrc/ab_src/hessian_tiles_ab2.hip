// A/B build only (-DHOT_AB_KERNELS, libhotmi355x_ab.so): included by ../hessian_tiles.hip inside `#ifdef HOT_AB_KERNELS`; not part of the product library.
// First-generation kernels and launch-structure alternatives that tests/test_gpu_variants.py and the tools compare the production kernels with.
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

