// libhotmi355x — APIC particle <-> grid transfers over the SPGrid block grid.
//
//   k_p2g_cells    particlesToGridHelper<true,false> (reference Lib/MPM/MpmSimulationBase.cpp:611-656): one particle
//                  group (= one SPGrid page worth of base cells) per workgroup; the (BX+2)(BY+2)(BZ+2) nodes the
//                  group can touch are accumulated in LDS ((cell, node) items summed in registers, one ds_add per
//                  item and quantity) and written as a partial tile; k_tile_reduce sums the <= 8 partial tiles of
//                  every node in a fixed order.  The reference's 8 sequential colour passes (:621-655) exist only to
//                  avoid write races between pages.  k_p2g is the first version (one ds_add per particle), kept for A/B.
//   k_block_count / k_number_nodes
//                  MpmGrid::getNumNodes (Lib/MPM/MpmGrid.h:148-161) — serial in the reference; here a per-block
//                  ballot + an exclusive scan over the insertion-ordered block list reproduce the ids bit-exactly;
//                  also v /= m (MpmSimulationBase.cpp:521-532), buildMassMatrix (:817-826) and id2coord
//                  (ImplicitSolver.h:474-477).
//   k_g2p          constructNewVelocityFromNewtonResult (:891-901) + gridToParticlesHelper<true,false,false>
//                  (:930-1007) + evolveStrain (Force/FBasedMpmForceHelper.cpp:99-114) + applyPlasticity (:1044-1064)
//                  fused: node tile staged in LDS, particle streams fully coalesced.
#include "hot_impl.h"
#include "hot_constitutive.h"

namespace hot {

// decode tile node t -> node slot (which of the 8 neighbour pages, which element)
template <class T>
__device__ __forceinline__ int tile_slot(int t, const int32_t* __restrict__ nb8)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2;
    int tz = t % TZ, ty = (t / TZ) % TY, tx = t / (TZ * TY);
    int ox = tx >> G::xb, oy = ty >> G::yb, oz = tz >> G::zb;
    int elem = ((tx & (G::BX - 1)) << (G::yb + G::zb)) | ((ty & (G::BY - 1)) << G::zb) | (tz & (G::BZ - 1));
    return nb8[ox * 4 + oy * 2 + oz] * G::EPB + elem;
}

#ifdef HOT_AB_KERNELS
#include "ab_src/transfer_ab1.hip"
#endif

#ifdef HOT_AB_KERNELS
#include "ab_src/transfer_ab2.hip"
#endif

#ifdef HOT_AB_KERNELS
#include "ab_src/transfer_ab3.hip"
#endif

// Round 6: P2G as a stream.  k_p2g_cells2 (above) is one workgroup per particle group with ONE chain header -> staging loads -> barrier -> items ->
// barrier -> write-out in its life; four of them share a compute unit and more than half of their wave cycles are spent parked in that chain
// (profiles/r05_sq_counters_C2.json).  Here a workgroup is persistent (two per compute unit, groups dealt round robin) and runs the chain as a
// pipeline over its units (a unit = up to CH particles of one group):
//   * the 16 per-particle scalars x, m, v, C of unit n+1 travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wavefront instruction,
//     no registers, no ds_write pass) into the second of two staging buffers while the items of unit n run; the unit's slice of cell_first likewise;
//   * mass * C and mass * v are no longer formed while staging (the DMA moves raw arrays): an item multiplies its weight by the mass once per node,
//     which is the product the mass sum needs anyway; the CN quantity's per-particle factor is formed by one thread per particle from mu / lambda
//     requested a unit ahead;
//   * with two workgroups per compute unit (LDS-bound) an item may use 256 registers: in fp64 too all five quantities of the item's nine nodes are
//     summed in one item (45 sums), so a particle's 1-D weights are recomputed 3 times, not 6;
//   * the partial tile of group n-1 is written out at the head of unit n and drains under its items.
// The waits: ONE `s_waitcnt vmcnt(0)` per unit, at the head, where everything in flight (the unit's DMA, issued a unit ago; the tile stores) is old;
// the barriers inside a unit are raw s_barrier + lgkmcnt(0) (a __syncthreads() would drain the DMA of the next unit).  The staging buffers are
// two distinct __shared__ objects and the unit loop is unrolled by two, so that the compiler's wait-count insertion can tell the buffer the items
// read from the buffer the DMA fills.
#ifdef HOT_HT_CLOCKS
__device__ unsigned long long p2g_clk[12]; // k_p2g_stream, thread 0 (an item wavefront) | thread 192 (the serving wavefront), summed over the workgroups: 0 wait at barrier (1), 1 items | serving, 2 wait at barrier (2), 3 tile write-out, 5 units
#define P2GS_CLK(i) \
    do { \
        if (tid == 0 || tid == 192) { \
            const unsigned long long t_ = clock64(); \
            sclk_[i] += t_ - st0_, st0_ = t_; \
        } \
    } while (0)
#else
#define P2GS_CLK(i)
#endif
template <class T>
__device__ __forceinline__ void lds_only_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// ds_add_f64 the compiler does not see as an LDS access: its wait-count insertion orders every LDS atomic behind a pending LDS-DMA
// (s_waitcnt vmcnt(0) in front of the first ds_add of an item, measured in the ISA: the next unit's DMA would be drained under the current items),
// whichever objects the two touch.  Nothing is returned; the barriers of k_p2g_stream wait for lgkmcnt(0) themselves.
#ifndef HOT_P2GS_EXP
#define HOT_P2GS_EXP 0
#endif
template <int OFFSET>
__device__ __forceinline__ void lds_add_f64_asm(uint32_t lds_byte_address, double v)
{
    static_assert(OFFSET >= 0 && OFFSET < 65536, "ds offset field is 16 bits");
    asm volatile("ds_add_f64 %0, %1 offset:%2" ::"v"(lds_byte_address), "v"(v), "n"(OFFSET) : "memory");
}
template <int N, class F>
__device__ __forceinline__ void static_for(F f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
template <class T, bool WITH_CN>
__global__ __launch_bounds__(256) void k_p2g_stream(const T* __restrict__ X, const T* __restrict__ V, const T* __restrict__ M, const T* __restrict__ C, const T* __restrict__ Mu,
    const T* __restrict__ Lam, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin, const int32_t* __restrict__ group_cell0,
    const int32_t* __restrict__ cell_first, T* __restrict__ part, T dx, T one_over_dx, int Ng)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    constexpr int NQ = WITH_CN ? 5 : 4, NS = 16 + (WITH_CN ? 1 : 0), CH = sizeof(T) == 4 ? 512 : 256;
    constexpr int IT = 192; // threads that run items (wavefronts 0 - 2); wavefront 3 serves them
    using AT = AccT<T>;
    __shared__ __attribute__((aligned(16))) T sp[2][NS][CH];
    __shared__ AT acc[NQ][TILE];
    __shared__ int32_t segs[2][G::EPB + 2];
    __shared__ int32_t nsegs[2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t acc_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&acc[0][0];
#ifdef HOT_HT_CLOCKS
    unsigned long long sclk_[6] = { 0, 0, 0, 0, 0, 0 }, st0_ = clock64();
#endif
    struct Unit {
        int g, ch, first, last, c0, c1, ox, oy, oz;
    };
    auto header = [&](int g, Unit& u) __attribute__((always_inline)) { // wave-uniform: scalar loads
        u.g = g, u.first = group_first[g], u.last = group_first[g + 1], u.c0 = group_cell0[g], u.c1 = group_cell0[g + 1];
        u.ox = group_origin[3 * g], u.oy = group_origin[3 * g + 1], u.oz = group_origin[3 * g + 2];
        u.ch = u.first;
    };
    auto advance = [&](const Unit& u, Unit& n) __attribute__((always_inline)) -> bool { // the unit after u of this workgroup
        if (u.ch + CH < u.last) {
            n = u, n.ch = u.ch + CH;
            return true;
        }
        const int g = u.g + (int)gridDim.x;
        if (g >= Ng) return false;
        header(g, n);
        return true;
    };
    // ---- the serving wavefront: everything of unit u that does not need the tile.  32 DMA pieces of 1 KiB (16 scalars x 2 halves; lane i moves
    // 16 bytes; a lane past the unit's last particle re-reads the unit's first bytes, its LDS slots are never looked at; the arrays carry 16 bytes
    // of slack behind their last element, reserve_particles), the CN factor m sqrt(..) from mu / lambda (plain loads, one wait with the DMA), the
    // cell segments of the unit from cell_first (lane = cell).
    auto serve = [&](const Unit& u, int b) __attribute__((always_inline)) {
        const int valid = (min(u.last, u.ch + CH) - u.ch) * (int)sizeof(T); // bytes per scalar
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int q = e >> 1, half = e & 1;
            const T* src = q < 3 ? X + (int64_t)q * Np : (q == 3 ? M : (q < 7 ? V + (int64_t)(q - 4) * Np : C + (int64_t)(q - 7) * Np));
            const int off = half * 1024 + lane * 16;
            const char* ga = (const char*)(src + u.ch) + (off < valid ? off : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga, (__attribute__((address_space(3))) void*)((char*)&sp[b][q][0] + half * 1024), 16, 0, 0);
        }
        constexpr int PPL = CH / 64; // particles per lane
        T rmu[PPL], rla[PPL];
        if constexpr (WITH_CN) {
#pragma unroll
            for (int r = 0; r < PPL; ++r) {
                const int64_t p = min((int64_t)u.ch + lane + 64 * r, Np - 1);
                rmu[r] = Mu[p], rla[r] = Lam[p];
            }
        }
        int cfa[(G::EPB + 63) / 64], cfb[(G::EPB + 63) / 64];
#pragma unroll
        for (int r = 0; r < (G::EPB + 63) / 64; ++r) {
            const int c = min(lane + 64 * r, u.c1 - u.c0 - 1);
            cfa[r] = cell_first[u.c0 + c], cfb[r] = cell_first[u.c0 + c + 1];
        }
        if (lane == 0) nsegs[b] = 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // the unit's scalars are in LDS
        if constexpr (WITH_CN) {
#pragma unroll
            for (int r = 0; r < PPL; ++r) {
                const int l = lane + 64 * r;
                const T mu = rmu[r], la = rla[r];
                sp[b][NS - 1][l] = sp[b][3][l] * hsqrt((T)3 * ((T)2 * mu + la) * ((T)2 * mu + la) + (T)6 * la * la + (T)12 * mu * mu);
            }
        }
#pragma unroll
        for (int r = 0; r < (G::EPB + 63) / 64; ++r) {
            const int s0 = max(cfa[r], u.ch), s1 = min(cfb[r], min(u.ch + CH, u.last));
            if (lane + 64 * r < u.c1 - u.c0 && s1 > s0) segs[b][atomicAdd(&nsegs[b], 1)] = (s0 - u.ch) | ((s1 - u.ch) << 16);
        }
    };
    // ---- the item wavefronts: (cell segment, node row j, half of the segment) -> the nine nodes (i, k) of the row, all NQ quantities summed in
    // registers; the two halves of a (segment, row) sit in neighbouring lanes and are added by a DPP move before ONE lane of the pair adds the 9 NQ
    // sums to the tile (half the LDS atomics, and no two lanes of a wavefront instruction add to the same address)
    auto items = [&](const Unit& u, int b) __attribute__((always_inline)) {
        const int n6 = nsegs[b] * 6;
        // (items numbered segment-major: the six lanes of a cell read the same particles — LDS broadcasts.  Numbered row-major — (j, segment, half), so that the lanes of
        // one ds_add_f64 never meet on a node — the item phase took 9.6 k instead of 8.65 k clocks per unit: the reads of 32 different cells per wavefront cost more
        // than the same-address atomics they avoid; profiles/r06_p2g_experiments.txt)
        for (int it = tid; it < n6; it += IT) {
            const int sd = segs[b][it / 6], j = (it % 6) >> 1, hf = it & 1, s0 = sd & 0xffff, s1 = sd >> 16;
            const int mid = (s0 + s1 + 1) >> 1, l0 = hf ? mid : s0, l1 = hf ? s1 : mid;
            T a[3][3][NQ]; // [i][k][quantity]
#pragma unroll
            for (int e = 0; e < 9 * NQ; ++e) (&a[0][0][0])[e] = (T)0;
            // the base cell is the same for every particle of the segment
            const int b0 = base_node_of<T>(one_over_dx, sp[b][0][s0]), b1 = base_node_of<T>(one_over_dx, sp[b][1][s0]), b2 = base_node_of<T>(one_over_dx, sp[b][2][s0]);
            const T fb0 = (T)b0, fb1 = (T)b1, fb2 = (T)b2;
            for (int l = l0; l < (HOT_P2GS_EXP == 2 ? l0 : l1); ++l) { // (HOT_P2GS_EXP, clock builds only: 1 = one atomic per item, 2 = no particle loop)
                const T x0 = sp[b][0][l], x1 = sp[b][1][l], x2 = sp[b][2][l];
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
                const T m = sp[b][3][l];
                T cn = (T)0, u3[3], cc[3], ee[3];
                if constexpr (WITH_CN) cn = sp[b][NS - 1][l];
#pragma unroll
                for (int q = 0; q < 3; ++q) u3[q] = sp[b][10 + q][l] * d1 + sp[b][4 + q][l], cc[q] = sp[b][7 + q][l], ee[q] = sp[b][13 + q][l];
                T d0[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) d0[i] = (T)(b0 + i) * dx - x0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T d2 = (T)(b2 + k) * dx - x2, wjk = wj * wk[k], mwjk = m * wjk;
                    T t[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) t[q] = ee[q] * d2 + u3[q];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const T mw = wi[i] * mwjk; // mass x weight
                        a[i][k][0] += mw;
#pragma unroll
                        for (int q = 0; q < 3; ++q) a[i][k][1 + q] += (cc[q] * d0[i] + t[q]) * mw;
                        if constexpr (WITH_CN) a[i][k][NQ - 1] += cn * (wi[i] * wjk);
                    }
                }
            }
#if HOT_P2GS_EXP == 1
            {
                T sum = 0;
#pragma unroll
                for (int e = 0; e < 9 * NQ; ++e) sum += (&a[0][0][0])[e];
                lds_add_f64_asm<0>(acc_base + 8 * (((b0 - u.ox) * TY + (b1 - u.oy + j)) * TZ + (b2 - u.oz)), (double)sum);
                continue;
            }
#endif
            // first half + second half (lanes 2 n, 2 n + 1: quad_perm [1, 0, 3, 2]); then the even lane adds sums 0 .. H - 1 to the tile, the odd lane sums
            // H .. 9 NQ - 1: a ds_add_f64 costs the same ~100 clocks whether 32 or 64 of its lanes carry a sum (measured: 45 per item with every second
            // lane switched off took as long as 45 on all lanes), so the pair's 9 NQ sums go out in H = ceil(9 NQ / 2) wavefront instructions, not 9 NQ
#pragma unroll
            for (int e = 0; e < 9 * NQ; ++e) (&a[0][0][0])[e] += dpp_move<0xb1, 0xf>((&a[0][0][0])[e]);
            constexpr int H = (9 * NQ + 1) / 2;
            const uint32_t abase = acc_base + 8 * (((b0 - u.ox) * TY + (b1 - u.oy + j)) * TZ + (b2 - u.oz));
            static_for<H>([&](auto sc) __attribute__((always_inline)) {
                constexpr int s0_ = decltype(sc)::value, s1_ = s0_ + H < 9 * NQ ? s0_ + H : 0; // (9 NQ odd: the odd lane's last slot adds 0 to sum 0's node)
                constexpr int o0 = (((s0_ / NQ) / 3) * TY * TZ + (s0_ / NQ) % 3) * 8 + (s0_ % NQ) * TILE * 8, o1 = (((s1_ / NQ) / 3) * TY * TZ + (s1_ / NQ) % 3) * 8 + (s1_ % NQ) * TILE * 8;
                const T v = hf ? (s0_ + H < 9 * NQ ? (&a[0][0][0])[s1_] : (T)0) : (&a[0][0][0])[s0_];
                lds_add_f64_asm<0>(abase + (hf ? o1 : o0), (double)v);
            });
        }
    };
    if ((int)blockIdx.x >= Ng) return;
    for (int t = tid; t < NQ * TILE; t += 256) (&acc[0][0])[t] = (AT)0;
    Unit cur, nxt;
    header((int)blockIdx.x, cur);
    if (wave == 3) serve(cur, 0);
    int b = 0;
    for (;;) {
        const bool have_next = advance(cur, nxt);
        lds_only_barrier<T>(); // (1) unit cur is staged in buffer b; the tile is zero or holds the group's earlier units
        P2GS_CLK(0);
        if (wave == 3) {
            if (have_next) serve(nxt, b ^ 1);
        }
        else
            items(cur, b);
        P2GS_CLK(1);
        lds_only_barrier<T>(); // (2) the unit's sums are in the tile; the next unit is staged
        P2GS_CLK(2);
        if (cur.ch + CH >= cur.last) { // the group's partial tile, coalesced; summed per node by k_tile_reduce
            T* out = part + (int64_t)cur.g * NQ * TILE;
            for (int t = tid; t < NQ * TILE; t += 256) out[t] = (T)(&acc[0][0])[t], (&acc[0][0])[t] = (AT)0;
        }
        P2GS_CLK(3);
#ifdef HOT_HT_CLOCKS
        sclk_[5] += 1;
#endif
        if (!have_next) break;
        cur = nxt, b ^= 1;
    }
#ifdef HOT_HT_CLOCKS
    if (tid == 0 || tid == 192)
        for (int i = 0; i < 6; ++i) atomicAdd(&p2g_clk[i + (tid ? 6 : 0)], sclk_[i]);
#endif
}

template <class T>
__global__ __launch_bounds__(256) void k_block_count(const T* __restrict__ gM, int32_t* block_count, int nb)
{
    using G = Geo<T>;
    int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    bool has = false;
    if (b < nb && lane < G::EPB) has = gM[(int64_t)b * G::EPB + lane] != (T)0;
    unsigned long long mask = __ballot(has);
    if (b < nb && lane == 0) block_count[b] = __popcll(mask);
}

template <class T>
__global__ __launch_bounds__(256) void k_number_nodes(const T* __restrict__ gM, T* gMV, int32_t* gIdx, const int32_t* __restrict__ block_base,
    const uint64_t* __restrict__ blocks, int32_t* dofSlot, int32_t* id2coord, T* mass, T* nodeV, int nb, int64_t slots)
{
    using G = Geo<T>;
    int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    bool valid = b < nb && lane < G::EPB;
    int64_t s = (int64_t)b * G::EPB + lane;
    T m = valid ? gM[s] : (T)0;
    bool has = valid && m != (T)0;
    unsigned long long mask = __ballot(has);
    if (!valid) return;
    int idx = -1;
    T v0 = 0, v1 = 0, v2 = 0;
    if (has) {
        idx = block_base[b] + __popcll(mask & ((1ULL << lane) - 1ULL));
        // g.v /= g.m (MpmSimulationBase.cpp:526): a true division, which also stays finite for the denormal masses
        // that corner nodes can get in fp32 (1 / m would overflow)
        v0 = gMV[s] / m, v1 = gMV[slots + s] / m, v2 = gMV[2 * slots + s] / m;
        int bi, bj, bk;
        G::linear_to_coord(blocks[b], bi, bj, bk);
        int ez = lane & (G::BZ - 1), ey = (lane >> G::zb) & (G::BY - 1), ex = lane >> (G::zb + G::yb);
        dofSlot[idx] = (int32_t)s;
        id2coord[3 * idx] = bi + ex, id2coord[3 * idx + 1] = bj + ey, id2coord[3 * idx + 2] = bk + ez;
        mass[idx] = m;
        nodeV[3 * idx] = v0, nodeV[3 * idx + 1] = v1, nodeV[3 * idx + 2] = v2;
    }
    gIdx[s] = idx;
    gMV[s] = v0, gMV[slots + s] = v1, gMV[2 * slots + s] = v2;
}

// DOF id (or -1) of every node of every particle group's tile (gathers of nodal fields then need one index load)
template <class T>
__global__ void k_tile_dof(const int32_t* __restrict__ group_nb, const int32_t* __restrict__ gIdx, int32_t* __restrict__ tileDof, int ng)
{
    using G = Geo<T>;
    constexpr int TILE = (G::BX + 2) * (G::BY + 2) * (G::BZ + 2);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)ng * TILE) return;
    const int g = (int)(e / TILE), t = (int)(e - (int64_t)g * TILE);
    tileDof[e] = gIdx[tile_slot<T>(t, group_nb + 8 * g)];
}

template <class T>
void Ctx<T>::p2g()
{
    need(Ng > 0, "hot_p2g before hot_sort");
    double t0 = wall_ms();
    int64_t slots = (int64_t)Nb * EPB;
    T one_over_dx = (T)1 / dx;
    const int nq = cfg.useCN ? 5 : 4;
#ifdef HOT_AB_KERNELS
    if (ab_flag("HOT_P2G_V1")) { // one LDS atomic per particle, node and quantity
        if (cfg.useCN)
            HOT_LAUNCH(this, "p2g", (k_p2g<T, true>), Ng, 256, 0, pX.p, pV.p, pM.p, pC.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_nb.p, gPart.p, dx, one_over_dx);
        else
            HOT_LAUNCH(this, "p2g", (k_p2g<T, false>), Ng, 256, 0, pX.p, pV.p, pM.p, pC.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_nb.p, gPart.p, dx, one_over_dx);
    }
    else if (ab_flag("HOT_P2G_CELLS1")) { // the 25-scalar staging version
        if (cfg.useCN)
            HOT_LAUNCH(this, "p2g", (k_p2g_cells<T, true>), Ng, P2G_THREADS, 0, pX.p, pV.p, pM.p, pC.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_cell0.p, cell_first.p, gPart.p, dx, one_over_dx);
        else
            HOT_LAUNCH(this, "p2g", (k_p2g_cells<T, false>), Ng, P2G_THREADS, 0, pX.p, pV.p, pM.p, pC.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_cell0.p, cell_first.p, gPart.p, dx, one_over_dx);
    }
    else if (ab_flag("HOT_P2G_CELLS2")) { // rounds 2 - 5: one workgroup per particle group, register staging
        if (cfg.useCN)
            HOT_LAUNCH(this, "p2g", (k_p2g_cells2<T, true>), Ng, 256, 0, pX.p, pV.p, pM.p, pC.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_cell0.p, cell_first.p, gPart.p, dx, one_over_dx);
        else
            HOT_LAUNCH(this, "p2g", (k_p2g_cells2<T, false>), Ng, 256, 0, pX.p, pV.p, pM.p, pC.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_cell0.p, cell_first.p, gPart.p, dx, one_over_dx);
    }
    else
#endif
    {
        // persistent workgroups, two per compute unit (what their LDS allows), groups dealt round robin
        const int grid = std::min(Ng, std::max(1, ab_int("HOT_P2G_WGS_PER_CU", 2)) * device_cus());
        if (cfg.useCN)
            HOT_LAUNCH(this, "p2g", (k_p2g_stream<T, true>), grid, 256, 0, pX.p, pV.p, pM.p, pC.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_cell0.p, cell_first.p, gPart.p, dx, one_over_dx, Ng);
        else
            HOT_LAUNCH(this, "p2g", (k_p2g_stream<T, false>), grid, 256, 0, pX.p, pV.p, pM.p, pC.p, pMu.p, pLam.p, Np, group_first.p, group_origin.p, group_cell0.p, cell_first.p, gPart.p, dx, one_over_dx, Ng);
    }
#ifdef HOT_HT_CLOCKS
    {
        unsigned long long h[12] = {};
        HOT_HIP(hipStreamSynchronize(stream));
        HOT_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(p2g_clk), sizeof(h)));
        if (h[5])
            fprintf(stderr, "p2g_stream clocks per unit (%llu units, %d groups), item wavefront | serving wavefront: wait (1) %.0f | %.0f, items | serving %.0f | %.0f, wait (2) %.0f | %.0f, tile write-out %.0f | %.0f\n", h[5], Ng,
                h[0] / (double)h[5], h[6] / (double)h[5], h[1] / (double)h[5], h[7] / (double)h[5], h[2] / (double)h[5], h[8] / (double)h[5], h[3] / (double)h[5], h[9] / (double)h[5]);
        else
            fprintf(stderr, "p2g clocks per workgroup (%d groups): header %.0f staging %.0f items %.0f write-out %.0f\n", Ng, h[0] / (double)Ng, h[1] / (double)Ng, h[2] / (double)Ng, h[3] / (double)Ng);
        memset(h, 0, sizeof(h));
        HOT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(p2g_clk), h, sizeof(h)));
    }
#endif
    reduce_tiles(nq, gM.p, gMV.p, gMV.p + slots, gMV.p + 2 * slots, gCN.p, "p2g_reduce");
    if (halo_mode()) { // the ranks that share a block add their partial sums: complete on every block this rank covers, zero on the others
        T* arr[5] = { gM.p, gMV.p, gMV.p + slots, gMV.p + 2 * slots, gCN.p };
        tile_exchange(arr, nq);
    }
    else if (sharded()) { // the shards' partial node sums -> the body's (one all-reduce of nq values per node slot)
        DBuf<T>& st = ap; // scratch, otherwise used only while the hierarchy is built
        st.reserve((size_t)nq * slots);
        copy(slots, gM.p, st.p), copy(3 * (size_t)slots, gMV.p, st.p + slots);
        if (nq == 5) copy(slots, gCN.p, st.p + 4 * slots);
        allreduce_tiles(st.p, nq);
        copy(slots, st.p, gM.p), copy(3 * (size_t)slots, st.p + slots, gMV.p);
        if (nq == 5) copy(slots, st.p + 4 * slots, gCN.p);
    }
    HOT_LAUNCH(this, "block_count", k_block_count<T>, div_up(Nb, 4), 256, 0, gM.p, block_count.p, Nb);
    if (halo_mode()) { // node counts of the blocks this rank does not cover (0 here) from the ranks that do
        IndexPhase ip(this);
        c_allreduce(block_count.p, Nb, HOT_COMM_I32, HOT_COMM_MAX, true);
    }
    scan.reserve(Nb + 1);
    Nn = exclusive_scan_i32(block_count.p, scan.p, Nb);
    HOT_CHECK((int64_t)Nn * 125 < (1LL << 31), HOT_ERR_CAPACITY, "num_nodes*125 overflows int32 (ImplicitSolver.h:479-480)");
    size_t n = std::max(Nn, 1);
    dofSlot.reserve(n, 1.25), id2coord.reserve(3 * n, 1.25), mass.reserve(n, 1.25), nodeV.reserve(3 * n, 1.25), bcIdx.reserve(n, 1.25);
    vn.reserve(3 * n, 1.25), dv.reserve(3 * n, 1.25), dv0.reserve(3 * n, 1.25), cnTol.reserve(n, 1.25), rhs.reserve(3 * n, 1.25);
    work0.reserve(3 * n, 1.25), work1.reserve(3 * n, 1.25), work2.reserve(3 * n, 1.25), work3.reserve(3 * n, 1.25);
    if (sharded()) { // id prefixes: the nodes of the blocks first touched by lower ranks (scan = exclusive block prefix of the node counts)
        nstart0.assign(comm.size + 1, Nn);
        for (int r = 0; r < comm.size; ++r)
            if (block_first[r] < Nb) HOT_HIP(hipMemcpyAsync(&nstart0[r], scan.p + block_first[r], sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        sync();
    }
    HOT_LAUNCH(this, "number_nodes", k_number_nodes<T>, div_up(Nb, 4), 256, 0, gM.p, gMV.p, gIdx.p, scan.p, blocks.p, dofSlot.p, id2coord.p, mass.p, nodeV.p, Nb, slots);
    if (halo_mode()) replicate_numbering(); // coordinates of ALL nodes (index structure), slot <-> id tables rebuilt from them
    {
        constexpr int TILE = (G::BX + 2) * (G::BY + 2) * (G::BZ + 2);
        tileDof.reserve((size_t)Ng * TILE, 1.25);
        HOT_LAUNCH(this, "tile_dof", k_tile_dof<T>, div_up((size_t)Ng * TILE, 256), 256, 0, group_nb.p, gIdx.p, tileDof.p, Ng);
    }
    if (halo_mode()) level0_ownership(); // level 0 exists from here on: row ownership, exchange lists, the mask of the solver's vector algebra
    stats.ms_p2g = wall_ms() - t0;
}

// Halo mode: a rank numbers the nodes of the blocks it covers (their masses are complete there); the coordinates of the other nodes —
// needed by the replicated index structure: colouring, coarse numbering, stencil columns — come from the ranks that first touch
// them: rank r's first-touch blocks are the id range [nstart0[r], nstart0[r + 1]), one padded all-gather of 3 ints per node.
__global__ void k_ids_pack(const int32_t* __restrict__ id2coord, int first, int cnt, int32_t* __restrict__ out, int maxc)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 3 * maxc) out[e] = e < 3 * cnt ? id2coord[3 * (int64_t)first + e] : 0;
}
struct IdRanges {
    int first[65];
};
__global__ void k_ids_unpack(int32_t* __restrict__ id2coord, IdRanges rg, int R, int me, const int32_t* __restrict__ in, int maxc)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)R * 3 * maxc) return;
    const int r = (int)(e / (3 * maxc)), k = (int)(e - (int64_t)r * 3 * maxc);
    if (r == me || k >= 3 * (rg.first[r + 1] - rg.first[r])) return;
    id2coord[3 * (int64_t)rg.first[r] + k] = in[e];
}
template <class T>
__global__ void k_slots_from_coords(HashMap bm, const int32_t* __restrict__ id2coord, int32_t* __restrict__ dofSlot, int32_t* __restrict__ gIdx, int nn)
{
    using G = Geo<T>;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nn) return;
    const uint64_t off = G::linear_offset(id2coord[3 * n], id2coord[3 * n + 1], id2coord[3 * n + 2]);
    const int32_t b = hash_find_id(bm, off >> 12);
    const int64_t s = (int64_t)b * G::EPB + (int)((off & 0xfff) >> G::data_bits);
    dofSlot[n] = (int32_t)s;
    gIdx[s] = n;
}
template <class T>
void Ctx<T>::replicate_numbering()
{
    IndexPhase ip(this);
    const int R = comm.size, me = comm.rank;
    int maxc = 0;
    IdRanges rg{};
    for (int r = 0; r <= R; ++r) rg.first[r] = nstart0[r];
    for (int r = 0; r < R; ++r) maxc = std::max(maxc, nstart0[r + 1] - nstart0[r]);
    if (maxc == 0) return;
    xsend.reserve((size_t)3 * maxc * 4), xrecv.reserve((size_t)3 * maxc * 4 * R);
    HOT_LAUNCH(this, "ids_pack", k_ids_pack, div_up(3 * (size_t)maxc, 256), 256, 0, id2coord.p, nstart0[me], nstart0[me + 1] - nstart0[me], (int32_t*)xsend.p, maxc);
    c_allgather(xsend.p, xrecv.p, (int64_t)3 * maxc * 4, true);
    HOT_LAUNCH(this, "ids_unpack", k_ids_unpack, div_up((size_t)R * 3 * maxc, 256), 256, 0, id2coord.p, rg, R, me, (const int32_t*)xrecv.p, maxc);
    HOT_LAUNCH(this, "slots_from_coords", k_slots_from_coords<T>, div_up(Nn, 256), 256, 0, block_map, id2coord.p, dofSlot.p, gIdx.p, Nn);
}

template <class T>
void Ctx<T>::get_grid(int32_t* ic, void* m, void* v)
{
    need(Nn > 0, "hot_get_grid before hot_p2g");
    if (halo_mode()) gather_all(*levels[0], mass.p, 1), gather_all(*levels[0], nodeV.p, 3); // the C ABI hands out complete arrays
    download(ic, id2coord.p, 3 * (size_t)Nn);
    download(m, mass.p, Nn);
    download(v, nodeV.p, 3 * (size_t)Nn);
    sync();
}

// ------------------------------------------------------------------------------------------------ G2P
// FACT: the 27-node sums by sum factorisation — the three nodes of a (i, j) column are first summed against the z weights (w_k, w_k d2_k,
// dw_k / dx: 27 multiply-adds per column), the column sums then enter the 21 accumulators once (26 per column): 480 instead of 760
// floating-point instructions per particle (the kernel is bound by the FP64 issue rate, not by HBM: 930 DP instructions per particle
// = 47 us at C2 at 16 lanes per clock and SIMD, against 44 us for its 350 MB at 8 TB/s).  The association of the sums differs from the
// reference's node-by-node order at round-off.  FACT = false (node by node) is kept for the A/B build (HOT_G2P_V1).
template <class T, int PLASTIC, bool FACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_g2p(T* __restrict__ X, T* __restrict__ V, T* __restrict__ C, T* __restrict__ F, const T* __restrict__ Fn, T* __restrict__ gradV_out,
    T* __restrict__ Mu, T* __restrict__ Lam, T* __restrict__ Jp, int64_t Np, const int32_t* __restrict__ group_first, const int32_t* __restrict__ group_origin,
    const int32_t* __restrict__ group_nb, const int32_t* __restrict__ gIdx, const T* __restrict__ nodeV, const T* __restrict__ dv, T dx, T one_over_dx, T dt, T apic_r,
    T cfl, T yield_stress, T sn0, T sn1, T sn2, T sn3, T sn4, int32_t* flags_out)
{
    using G = Geo<T>;
    constexpr int TY = G::BY + 2, TZ = G::BZ + 2, TILE = (G::BX + 2) * TY * TZ;
    __shared__ T nv[3][TILE];
    const int g = blockIdx.x;
    const int first = group_first[g], last = group_first[g + 1];
    // the position of this thread's first particle is requested before the tile gather, and the tile's DOF ids come from the per-group
    // table (tileDof) instead of the nb8 -> gIdx chain: the workgroup's dependent round trips (indices -> nodal values, particle data)
    // run side by side.  Fn is NOT held across the 27-node loop (round 3: with it the fp64 kernel needed 214 registers, two wavefronts
    // per SIMD; it is read after the loop, when only the 21 sums are live, and the other wavefronts cover that round trip).
    const int p0 = first + threadIdx.x;
    T xpre[3] = { 0, 0, 0 };
    if (p0 < last) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xpre[d] = X[(int64_t)d * Np + p0];
    }
    for (int t = threadIdx.x; t < TILE; t += 256) {
        int idx = gIdx[(int64_t)g * TILE + t]; // gIdx here = tileDof
        T a = 0, b = 0, c = 0;
        if (idx >= 0) {
            a = nodeV[3 * idx] + dv[3 * idx], b = nodeV[3 * idx + 1] + dv[3 * idx + 1], c = nodeV[3 * idx + 2] + dv[3 * idx + 2];
        }
        nv[0][t] = a, nv[1][t] = b, nv[2][t] = c;
    }
    __syncthreads();
    const int ox = group_origin[3 * g], oy = group_origin[3 * g + 1], oz = group_origin[3 * g + 2];
    const T D_inverse = (T)4 / (dx * dx);
    int myflags = 0;
    for (int p = first + threadIdx.x; p < last; p += 256) {
        T xp[3];
        if (p == p0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) xp[d] = xpre[d];
        }
        else { // groups of more than 256 particles
#pragma unroll
            for (int d = 0; d < 3; ++d) xp[d] = X[(int64_t)d * Np + p];
        }
        int base[3];
        T w[3][3], dw[3][3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bspline<T>(one_over_dx, xp[d], base[d], w[d], dw[d]);
        const int cx = base[0] - ox, cy = base[1] - oy, cz = base[2] - oz;
        T pic[3] = { 0, 0, 0 };
        T B[9], gv[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) B[c] = (T)0, gv[c] = (T)0;
        if constexpr (FACT) {
            T wz[3], wd2[3], dwz[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) wz[k] = w[2][k], wd2[k] = w[2][k] * ((T)(base[2] + k) * dx - xp[2]), dwz[k] = one_over_dx * dw[2][k];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const T wi = w[0][i], dwi = one_over_dx * dw[0][i];
                const T d0 = (T)(base[0] + i) * dx - xp[0];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int t = ((cx + i) * TY + (cy + j)) * TZ + cz;
                    T s0[3], s1[3], s2[3]; // column sums: sum_k w_k v, sum_k w_k d2_k v, sum_k dw_k / dx v
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const T va = nv[c][t], vb = nv[c][t + 1], vc = nv[c][t + 2];
                        s0[c] = fma(wz[2], vc, fma(wz[1], vb, wz[0] * va));
                        s1[c] = fma(wd2[2], vc, fma(wd2[1], vb, wd2[0] * va));
                        s2[c] = fma(dwz[2], vc, fma(dwz[1], vb, dwz[0] * va));
                    }
                    const T wij = wi * w[1][j], gi = dwi * w[1][j], gj = wi * (one_over_dx * dw[1][j]);
                    const T d1 = (T)(base[1] + j) * dx - xp[1];
                    const T a0 = wij * d0, a1 = wij * d1;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        pic[c] = fma(wij, s0[c], pic[c]);
                        B[c] = fma(a0, s0[c], B[c]), B[3 + c] = fma(a1, s0[c], B[3 + c]), B[6 + c] = fma(wij, s1[c], B[6 + c]);
                        gv[c] = fma(gi, s0[c], gv[c]), gv[3 + c] = fma(gj, s0[c], gv[3 + c]), gv[6 + c] = fma(wij, s2[c], gv[6 + c]);
                    }
                }
            }
        }
        else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T wi = w[0][i], dwi = one_over_dx * dw[0][i];
            T d0 = (T)(base[0] + i) * dx - xp[0];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                T wij = wi * w[1][j];
                T dwij_i = dwi * w[1][j], dwij_j = wi * one_over_dx * dw[1][j];
                T d1 = (T)(base[1] + j) * dx - xp[1];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    T wijk = wij * w[2][k];
                    T g0 = dwij_i * w[2][k], g1 = dwij_j * w[2][k], g2 = wij * one_over_dx * dw[2][k];
                    T d2 = (T)(base[2] + k) * dx - xp[2];
                    int t = ((cx + i) * TY + (cy + j)) * TZ + (cz + k);
                    T v0 = nv[0][t], v1 = nv[1][t], v2 = nv[2][t];
                    pic[0] += wijk * v0, pic[1] += wijk * v1, pic[2] += wijk * v2;
                    T wv0 = wijk * v0, wv1 = wijk * v1, wv2 = wijk * v2;
                    B[0] += wv0 * d0, B[1] += wv1 * d0, B[2] += wv2 * d0;
                    B[3] += wv0 * d1, B[4] += wv1 * d1, B[5] += wv2 * d1;
                    B[6] += wv0 * d2, B[7] += wv1 * d2, B[8] += wv2 * d2;
                    gv[0] += v0 * g0, gv[1] += v1 * g0, gv[2] += v2 * g0;
                    gv[3] += v0 * g1, gv[4] += v1 * g1, gv[5] += v2 * g1;
                    gv[6] += v0 * g2, gv[7] += v1 * g2, gv[8] += v2 * g2;
                }
            }
        }
        }
        Mat3<T> Fo;
#pragma unroll
        for (int c = 0; c < 9; ++c) Fo.a[c] = Fn[(int64_t)c * Np + p];
        V[p] = pic[0], V[Np + p] = pic[1], V[2 * Np + p] = pic[2];
        T ra = (apic_r + (T)1) * (T)0.5, rb = (apic_r - (T)1) * (T)0.5;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) C[(int64_t)(c * 3 + r) * Np + p] = ra * (B[c * 3 + r] * D_inverse) + rb * (B[r * 3 + c] * D_inverse);
        T inc0 = dt * pic[0], inc1 = dt * pic[1], inc2 = dt * pic[2];
        X[p] = xp[0] + inc0, X[Np + p] = xp[1] + inc1, X[2 * Np + p] = xp[2] + inc2;
        T inc = inc0 * inc0 + inc1 * inc1 + inc2 * inc2, dx2 = dx * dx;
        if (inc > dx2) myflags |= 1;
        if (inc > dx2 * (T)0.25 * (cfl * cfl)) myflags |= 2;
        if (gradV_out)
#pragma unroll
            for (int c = 0; c < 9; ++c) gradV_out[(int64_t)c * Np + p] = gv[c];
        // F = (I + dt gradV) Fn   (restoreStrain + evolveStrain)
        Mat3<T> A, Fnew;
#pragma unroll
        for (int c = 0; c < 9; ++c) A.a[c] = dt * gv[c] + ((c % 4 == 0) ? (T)1 : (T)0);
        Fnew = m3_mul(A, Fo);
        if (PLASTIC == 1) {
            von_mises_project(Fnew, Mu[p], Lam[p], yield_stress);
        }
        else if (PLASTIC == 2) {
            T mu = Mu[p], la = Lam[p], jp = Jp[p];
            snow_project(Fnew, mu, la, jp, sn0, sn1, sn2, sn3, sn4);
            Mu[p] = mu, Lam[p] = la, Jp[p] = jp;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) F[(int64_t)c * Np + p] = Fnew.a[c];
    }
    // one global atomic per wavefront at most (2 M same-address atomics cost more than the whole transfer)
    unsigned long long m1 = __ballot(myflags & 1), m2 = __ballot(myflags & 2);
    if ((threadIdx.x & 63) == 0 && (m1 | m2)) {
        int bits = (m1 ? 1 : 0) | (m2 ? 2 : 0);
        int cur = __hip_atomic_load(flags_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((cur & bits) != bits) atomicOr(flags_out, bits); // already-set bits need no further traffic
    }
}

template <class T>
void Ctx<T>::g2p(double dt_, int32_t* flags)
{
    need(Nn > 0, "hot_g2p before hot_p2g/hot_begin_step");
    double t0 = wall_ms();
    int32_t* dflags = (int32_t*)(dscal.p + 200);
    HOT_HIP(hipMemsetAsync(dflags, 0, 4, stream));
    T one_over_dx = (T)1 / dx;
    if (halo_mode()) halo_gather(*levels[0], dv.p); // dv at the nodes of this rank's particle tiles that other ranks own
#define G2P_ARGS pX.p, pV.p, pC.p, pF.p, pFn.p, (keep_debug ? pGradV.p : (T*)nullptr), pMu.p, pLam.p, pJp.p, Np, group_first.p, group_origin.p, group_nb.p, tileDof.p, nodeV.p, dv.p, dx, \
                 one_over_dx, (T)dt_, (T)cfg.apic_rpic_ratio, (T)cfg.cfl, (T)cfg.yield_stress, (T)cfg.snow[0], (T)cfg.snow[1], (T)cfg.snow[2], (T)cfg.snow[3], (T)cfg.snow[4], dflags
#ifdef HOT_AB_KERNELS
    if (ab_flag("HOT_G2P_V1")) { // node-by-node sums
        if (cfg.plasticity == 1)
            HOT_LAUNCH(this, "g2p", (k_g2p<T, 1, false>), Ng, 256, 0, G2P_ARGS);
        else if (cfg.plasticity == 2)
            HOT_LAUNCH(this, "g2p", (k_g2p<T, 2, false>), Ng, 256, 0, G2P_ARGS);
        else
            HOT_LAUNCH(this, "g2p", (k_g2p<T, 0, false>), Ng, 256, 0, G2P_ARGS);
    }
    else
#endif
    if (cfg.plasticity == 1)
        HOT_LAUNCH(this, "g2p", (k_g2p<T, 1, true>), Ng, 256, 0, G2P_ARGS);
    else if (cfg.plasticity == 2)
        HOT_LAUNCH(this, "g2p", (k_g2p<T, 2, true>), Ng, 256, 0, G2P_ARGS);
    else
        HOT_LAUNCH(this, "g2p", (k_g2p<T, 0, true>), Ng, 256, 0, G2P_ARGS);
#undef G2P_ARGS
    int32_t f = 0;
    HOT_HIP(hipMemcpyAsync(&f, dflags, 4, hipMemcpyDeviceToHost, stream));
    sync();
    if (sharded()) {
        int32_t bits[2] = { f & 1, (f >> 1) & 1 };
        c_allreduce(bits, 2, HOT_COMM_I32, HOT_COMM_MAX, false);
        f = bits[0] | (bits[1] << 1);
    }
    if (flags) *flags = f;
    stats.ms_g2p = wall_ms() - t0;
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
