// libhotmi355x — Hessian assembly, production kernel of round 5: output-stationary row tiles, barrier-free wavefront tasks,
// the contraction on broadcast FMAs (v_fmac_*_dpp row_newbcast), particle data staged per wavefront, never per workgroup.
//
// Mathematics (reference Projects/multigrid/ImplicitSolver.h:498-552): every ordered node pair (i, j) of every particle adds
//   H(i, j)[a][b] = V_p dt^2 sum_{v,q} dP_{(a,v),(b,q)} g_i[v] g_j[q],   g_i = Fn^T grad w_i
// to row dof_i, slot linearOffset(node_i - node_j).  With  E[(a,r),(b,s)] = V_p dt^2 sum_{v,q} Fn(r,v) Fn(s,q) dP_{(a,v),(b,q)}
// (symmetric 9 x 9, 45 scalars, formed once per particle by k_dpdf_rec) the deformation gradient leaves the inner loops:
//   H(i, j)[a][b] = sum_{r,s} E[(a,r),(b,s)] gw_i[r] gw_j[s],   gw_i = grad w_i(x_p)
// and is evaluated as  K_i[a][b][s] = sum_r E[(a,r),(b,s)] gw_i[r]  (27 values per (particle, row node)), then
// H(i, j)[a][b] = sum_s K_i[a][b][s] gw_j[s]  (27 multiply-adds per block).
//
// k_hessian_tiles2 (round 2 - 4, hessian_tiles.hip, now A/B build only) staged dP, g and K of 40-particle chunks in LDS behind four
// barriers per chunk and paid 18 LDS reads per 27 multiply-adds in its pair phase: 13.7 ms at C2, 62 % of the wave cycles parked on
// barriers, 10 % of the FP64 rate.  Here (5.2 ms at C2):
//   * pass 1, k_dpdf_rec: one 128-scalar record per particle: E (45) and grad w of the 27 kernel nodes (81);
//   * pass 1b, k_tile_cells: per 2x2x2 row tile the particle ranges of the 4x4x4 base cells around it and its 8 row DOFs;
//   * pass 2, k_hessian_rows: one workgroup (7 wavefronts in fp64, 8 in fp32) per tile, its 8 rows x 125 slots x 9 values in LDS (72 KB, two
//     workgroups per CU); nothing else of the workgroup is shared, there is no barrier between the prologue and the write-out;
//   * a TASK = (base cell, x-plane of the tile): the <= 4 tile rows of that plane inside the cell's 3x3x3 support (fp32: the whole cell,
//     <= 8 rows — kHrCellTasks below).  Wavefronts draw tasks from an LDS counter, those with most rows first.  A wavefront walks the cell's particles with lane = (half h, column node j): the 27
//     column nodes of the cell twice, half 0 owning the (a, b) entries 0..4 of every 3x3 block, half 1 the entries 4..8;
//   * per particle the record comes with two coalesced loads (requested one particle ahead) and is parked in the wavefront's LDS stage;
//     a lane picks its three E values, grad w of its column node and — at wave-uniform addresses — grad w of the task's row nodes from
//     there: one LDS round trip per particle;
//   * per (particle, row): lanes 0..14 of every 16-lane DPP row form the 15 K values of their half (3 FMAs), then every lane runs
//     15 broadcast FMAs  acc[ab] += K[lane 3 ab + s of my DPP row] * gw_j[s]  — the K operand never leaves the register file —
//     into 5 accumulators per row (40 VGPRs for the task's 4 rows);
//   * after the cell's last particle the accumulators go to the LDS tile with 5 (4) ds_add_f64 per lane and row.
// 18 VALU instructions per (particle, row) for 27 x 27 multiply-adds on 64 lanes: 70 % of them useful, no LDS atomics or index
// decoding inside the particle loop.  How it got from 8.1 to 5.2 ms: DESIGN.md section 6.
#include "hot_impl.h"
#include "hot_constitutive.h"

namespace hot {

__host__ __device__ constexpr int sym45i(int a, int b) { return a <= b ? (a * 9 - (a * (a - 1)) / 2 + (b - a)) : (b * 9 - (b * (b - 1)) / 2 + (a - b)); }

constexpr int REC = 128; // scalars per particle record: E (45), grad w of the 27 kernel nodes (81), pad

// ---- pass 1: the particle record
template <class T>
__global__ __launch_bounds__(256) void k_dpdf_rec(const T* __restrict__ X, const T* __restrict__ Fn, const T* __restrict__ Ft, const T* __restrict__ Vol, const T* __restrict__ Mu,
    const T* __restrict__ Lam, T* __restrict__ rec, int64_t Np, T dt, T one_over_dx, int project)
{
    // (a lane past the last particle computes the last particle's record again and stores nothing: the stores below are a wavefront's joint work)
    const int64_t p = min((int64_t)blockIdx.x * 256 + threadIdx.x, Np - 1);
    Mat3<T> Fc;
#pragma unroll
    for (int c = 0; c < 9; ++c) Fc.a[c] = Ft[(int64_t)c * Np + p];
    HessBlocks<T> h;
    corotated_hessian(Fc, Mu[p], Lam[p], project != 0, h);
    const T sc = Vol[p] * dt * dt;
    const Mat3<T>& U = h.U;
    const Mat3<T>& V = h.V;
    auto Kval = [&](int a, int b, int c, int d) -> T { // non-zero only for (aa,cc) [A], (ab,ab) and (ab,ba) [B blocks]
        if (a == b && c == d) return h.A(a, c);
        if (a != b && ((a == c && b == d) || (a == d && b == c))) {
            int lo = a < b ? a : b, hi = a < b ? b : a;
            const T* B = (lo == 0 && hi == 1) ? h.B01 : ((lo == 1 && hi == 2) ? h.B12 : h.B20);
            int ia, ic;
            if (lo == 0 && hi == 2) {
                ia = (a == 2) ? 0 : 1, ic = (c == 2) ? 0 : 1;
            }
            else {
                ia = (a == lo) ? 0 : 1, ic = (c == lo) ? 0 : 1;
            }
            return B[ia + ic];
        }
        return (T)0;
    };
    // D[(a,v),(b,q)] = V_p dt^2 dP/dF in the rotated-back frame, index (a + 3 v, b + 3 q), full symmetric 9 x 9 in registers
    T D[81];
#pragma unroll
    for (int ij = 0; ij < 9; ++ij) {
        const int jj = ij / 3, ii = ij - jj * 3;
#pragma unroll
        for (int rs = ij; rs < 9; ++rs) {
            const int ss = rs / 3, rr = rs - ss * 3;
            T v = (T)0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    T ub = U(ii, a) * V(jj, b);
                    if (a == b) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) v += Kval(a, a, c, c) * ub * U(rr, c) * V(ss, c);
                    }
                    else {
                        v += Kval(a, b, a, b) * ub * U(rr, a) * V(ss, b) + Kval(a, b, b, a) * ub * U(rr, b) * V(ss, a);
                    }
                }
            D[ij * 9 + rs] = D[rs * 9 + ij] = v * sc;
        }
    }
    // E[(a,r),(b,s)] = sum_{v,q} Fn(r,v) Fn(s,q) D[(a,v),(b,q)],  Fn(r,c) at component r + 3 c
    T F9[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) F9[c] = Fn[(int64_t)c * Np + p];
    T Th[81]; // Th[(a,r),(b,q)] = sum_v Fn(r,v) D[(a,v),(b,q)]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int n = 0; n < 9; ++n) Th[(a + 3 * r) * 9 + n] = F9[r] * D[a * 9 + n] + F9[r + 3] * D[(a + 3) * 9 + n] + F9[r + 6] * D[(a + 6) * 9 + n];
    T w[3][3], dw[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int base;
        bspline<T>(one_over_dx, X[(int64_t)d * Np + p], base, w[d], dw[d]);
#pragma unroll
        for (int k = 0; k < 3; ++k) dw[d][k] *= one_over_dx;
    }
    // The record leaves through LDS, a quarter (32 scalars) at a time: a lane holds ONE particle's values, and stored from there its 128 scalars would be
    // 128 instructions of 64 eight-byte pieces 1 KB apart (rounds 4 - 5: 0.96 ms per C2 assembly, most of it these stores); transposed in the wavefront's
    // own LDS stage (32 x 64 scalars, rows padded to 65: conflict-free both ways, no barrier — nothing but this wavefront touches it) a store instruction
    // writes 32 consecutive scalars of two particles: whole 128-byte lines.
    __shared__ T stage[4][32 * 65];
    T* st = stage[threadIdx.x >> 6];
    const int ln = threadIdx.x & 63;
    const int64_t pw0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); // the wavefront's first particle
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        auto emit = [&](int idx, T v) __attribute__((always_inline)) {
            if ((idx >> 5) == q) st[(idx & 31) * 65 + ln] = v; // (idx is a constant after unrolling: a quarter computes its own 32 values only)
        };
#pragma unroll
        for (int m = 0; m < 9; ++m)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int n = b + 3 * s;
                    if (n >= m) emit(sym45i(m, n), F9[s] * Th[m * 9 + b] + F9[s + 3] * Th[m * 9 + b + 3] + F9[s + 6] * Th[m * 9 + b + 6]);
                }
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            const int j0 = j / 9, j1 = (j / 3) % 3, j2 = j % 3;
            emit(45 + 3 * j, dw[0][j0] * (w[1][j1] * w[2][j2])), emit(46 + 3 * j, (w[0][j0] * w[2][j2]) * dw[1][j1]), emit(47 + 3 * j, (w[0][j0] * w[1][j1]) * dw[2][j2]);
        }
        emit(126, (T)0), emit(127, (T)0);
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const int pp = 2 * k + (ln >> 5), il = ln & 31;
            if (pw0 + pp < Np) rec[(pw0 + pp) * REC + 32 * q + il] = st[il * 65 + pp];
        }
    }
}

// ---- the 15 broadcast FMAs of a (particle, row): acc[ab] += K(lane 3 ab + s of this lane's 16-lane row) * g[s].  The s_nop covers
// the VALU-write -> DPP-read distance of K that the compiler's hazard recogniser does not see inside an asm statement.
__device__ __forceinline__ void dpp_row_fma(double (&acc)[5], double K, double g0, double g1, double g2)
{
#define HOT_DPPF(i, g, n) "v_fmac_f64_dpp %" #i ", %5, %" #g " row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n"
    asm("s_nop 1\n" HOT_DPPF(0, 6, 0) HOT_DPPF(1, 6, 3) HOT_DPPF(2, 6, 6) HOT_DPPF(3, 6, 9) HOT_DPPF(4, 6, 12) HOT_DPPF(0, 7, 1) HOT_DPPF(1, 7, 4) HOT_DPPF(2, 7, 7) HOT_DPPF(3, 7, 10)
            HOT_DPPF(4, 7, 13) HOT_DPPF(0, 8, 2) HOT_DPPF(1, 8, 5) HOT_DPPF(2, 8, 8) HOT_DPPF(3, 8, 11) HOT_DPPF(4, 8, 14)
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4])
        : "v"(K), "v"(g0), "v"(g1), "v"(g2));
#undef HOT_DPPF
}
__device__ __forceinline__ void dpp_row_fma(float (&acc)[5], float K, float g0, float g1, float g2)
{
#define HOT_DPPF(i, g, n) "v_fmac_f32_dpp %" #i ", %5, %" #g " row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n"
    asm("s_nop 1\n" HOT_DPPF(0, 6, 0) HOT_DPPF(1, 6, 3) HOT_DPPF(2, 6, 6) HOT_DPPF(3, 6, 9) HOT_DPPF(4, 6, 12) HOT_DPPF(0, 7, 1) HOT_DPPF(1, 7, 4) HOT_DPPF(2, 7, 7) HOT_DPPF(3, 7, 10)
            HOT_DPPF(4, 7, 13) HOT_DPPF(0, 8, 2) HOT_DPPF(1, 8, 5) HOT_DPPF(2, 8, 8) HOT_DPPF(3, 8, 11) HOT_DPPF(4, 8, 14)
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4])
        : "v"(K), "v"(g0), "v"(g1), "v"(g2));
#undef HOT_DPPF
}

// ---- pass 1b: per tile the first particle and the particle count of the 4x4x4 base cells around it and the DOF of its 8 rows, so that a
// tile workgroup starts from ONE coalesced load instead of a chain of five dependent ones (block offset -> hash probe -> cell id -> cell
// range; ~10 us per workgroup beside a busy neighbour, a sixth of the kernel).  TileTab: [tile][64] {first, count} then [tile][8] DOF.
template <class T>
__global__ __launch_bounds__(256) void k_tile_cells(const uint64_t* __restrict__ blocks, const int32_t* __restrict__ gIdx, const int32_t* __restrict__ cell_first, HashMap cmap, int2* __restrict__ tcell,
    int32_t* __restrict__ trow, int ntiles)
{
    using G = Geo<T>;
    constexpr int TPBY = G::BY / 2, TPBZ = G::BZ / 2, TPB = (G::BX / 2) * TPBY * TPBZ;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int tile_id = (int)(e >> 6), c64 = (int)(e & 63);
    if (tile_id >= ntiles) return;
    const int b = tile_id / TPB, tt = tile_id % TPB;
    int bx, by, bz;
    G::linear_to_coord(blocks[b], bx, by, bz);
    const int tx0 = bx + 2 * (tt / (TPBY * TPBZ)), ty0 = by + 2 * ((tt / TPBZ) % TPBY), tz0 = bz + 2 * (tt % TPBZ);
    const int cx = tx0 + (c64 >> 4) - 2, cy = ty0 + ((c64 >> 2) & 3) - 2, cz = tz0 + (c64 & 3) - 2; // base cell = tile origin + (-2..1)^3
    int first = 0, cnt = 0;
    if ((cx | cy | cz) >= 0) {
        int32_t c = hash_find_id(cmap, G::linear_offset(cx, cy, cz) >> G::data_bits);
        if (c >= 0) first = cell_first[c], cnt = cell_first[c + 1] - first;
    }
    tcell[e] = make_int2(first, cnt);
    if (c64 < 8) {
        int ex = (tx0 - bx) + (c64 >> 2), ey = (ty0 - by) + ((c64 >> 1) & 1), ez = (tz0 - bz) + (c64 & 1);
        trow[(int64_t)tile_id * 8 + c64] = gIdx[(int64_t)b * G::EPB + ((ex << (G::yb + G::zb)) | (ey << G::zb) | ez)];
    }
}

// ---- lane roles of a task wavefront: half h = lane >> 5 owns the block entries 4 h .. 4 h + 4; its lane jl = lane & 31 (< 27) the column node
// (j0, j1, j2) of the cell's kernel; lane n = lane & 15 (< 15) of every 16-lane DPP row forms K[(a, b) = 4 h + n / 3][s = n % 3]
struct RowsLane {
    int h, jl, j0, j1, j2;
    unsigned eo0, eo1, eo2; // record offsets of E[(a, r), (b, s)], r = 0, 1, 2
    unsigned go; // record offset of grad w of the lane's column node
};
__device__ __forceinline__ RowsLane rows_lane(int lane)
{
    RowsLane L;
    L.h = lane >> 5, L.jl = min(lane & 31, 26);
    L.j0 = L.jl / 9, L.j1 = (L.jl / 3) % 3, L.j2 = L.jl % 3;
    const int n = min(lane & 15, 14), ab = 4 * L.h + n / 3, ks = n % 3, ka = ab % 3, kb = ab / 3;
    L.eo0 = sym45i(ka, kb + 3 * ks), L.eo1 = sym45i(ka + 3, kb + 3 * ks), L.eo2 = sym45i(ka + 6, kb + 3 * ks);
    L.go = 45 + 3 * L.jl;
    return L;
}
struct RowsTask { // wave-uniform
    int lx0, ly0, lz0; // kernel index of the tile's row 0 inside the cell: row (rx, ry, rz) sits at (lx0 + rx, ly0 + ry, lz0 + rz)
    int m8; // the task's rows (rx << 2 | ry << 1 | rz): inside the cell's support and active; one x-plane of the tile or (kHrCellTasks) both
};

// One task: the particles [rp, rp + cnt records) of a cell against NR rows of a tile plane.  Records are requested TWO particles ahead into three
// rotating register sets, and the request stream runs on into the wavefront's NEXT task (its first two records: [nrp, nrp + ncnt)), so that a
// task starts with its records 0 and 1 in flight since the previous task's last two particles — with 8 particles per cell a cold start per
// task left a memory round trip exposed per ~8 visits (one record ahead, no look-ahead: 5.3 ms at C2).  The NR x 3 wave-uniform stage
// loads of a particle are issued together.
template <class T>
struct RowsPre { // the wavefront's records in flight: on entry of a walk [0] = its record 0, [1] = its record 1 (valid if cnt > 1)
    T r[3][2];
};
template <class T, int NR>
__device__ __forceinline__ void hr_walk(const T* __restrict__ rp /*wave-uniform*/, int cnt, const T* __restrict__ nrp, int ncnt, RowsPre<T>& P, const RowsTask& tk, T* __restrict__ stage,
    AccT<T>* __restrict__ tile, int lane, const RowsLane& ld)
{
    using AT = AccT<T>;
    int qs[NR], so[NR]; // rows of the task in ascending order (ordinals beyond the row count repeat row 0: computed, not stored), record offsets of their grad w
    {
        int m = tk.m8;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            qs[r] = m ? __builtin_ctz(m) : __builtin_ctz(tk.m8);
            m &= m - 1;
            so[r] = 45 + 3 * ((tk.lx0 + (qs[r] >> 2)) * 9 + (tk.ly0 + ((qs[r] >> 1) & 1)) * 3 + (tk.lz0 + (qs[r] & 1)));
        }
    }
    T acc[NR][5];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 5; ++e) acc[r][e] = (T)0;
    // The record of a particle (128 scalars) comes with TWO coalesced loads per wavefront, one particle ahead, is parked in the wavefront's
    // LDS stage and picked apart from there — a lane's three E values, grad w of its column node, grad w of the task's row nodes (wave-uniform
    // addresses) — in ONE round trip.  (As nine gathering global loads per particle the kernel waited for the texture addresser, 16 clocks per
    // vector memory instruction whatever its width: 8.2 ms at C2 with 48 % of the VALU cycles used; with the 1-D weights in the record and
    // grad w formed and exchanged through a second LDS strip, three dependent LDS round trips per particle: 6.25 ms.)
    auto work = [&](T rec0, T rec1) {
        stage[lane] = rec0, stage[64 + lane] = rec1;
        __builtin_amdgcn_wave_barrier();
        const T e0 = stage[ld.eo0], e1 = stage[ld.eo1], e2 = stage[ld.eo2];
        const T g0 = stage[ld.go], g1 = stage[ld.go + 1], g2 = stage[ld.go + 2];
        constexpr int RG = NR < 4 ? NR : 4; // rows per group of stage loads (eight rows: two groups, 24 registers of operands instead of 48)
#pragma unroll
        for (int r0 = 0; r0 < NR; r0 += RG) {
            T gw[RG][3];
#pragma unroll
            for (int r = 0; r < RG; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) gw[r][k] = stage[so[r0 + r] + k];
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const T K = e0 * gw[r][0] + e1 * gw[r][1] + e2 * gw[r][2];
                dpp_row_fma(acc[r0 + r], K, g0, g1, g2);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // request i of the stream: own record i, past the end the next task's records 0 and 1 (a next task of one particle: its record 0 twice)
    auto request = [&](int i, T(&R)[2]) {
        const T* q = i < cnt ? rp + i * REC : nrp + (i - cnt < ncnt ? i - cnt : ncnt - 1) * REC;
        R[0] = q[lane], R[1] = q[64 + lane];
        asm volatile("" ::: "memory"); // issued HERE, two visits ahead of its use
    };
    if (cnt == 1) request(1, P.r[1]); // (what arrived as "record 1" was this task's record 0 again)
    for (int l = 0; l < cnt; l += 3) {
        request(l + 2, P.r[2]);
        work(P.r[0][0], P.r[0][1]);
        if (l + 1 >= cnt) break;
        request(l + 3, P.r[0]);
        work(P.r[1][0], P.r[1][1]);
        if (l + 2 >= cnt) break;
        request(l + 4, P.r[1]);
        work(P.r[2][0], P.r[2][1]);
    }
    { // the next task's records 0 and 1 sit in the sets cnt % 3 and (cnt + 1) % 3: make them sets 0 and 1
        const int ph = cnt % 3; // wave-uniform
        if (ph == 1) {
#pragma unroll
            for (int e = 0; e < 2; ++e) P.r[0][e] = P.r[1][e], P.r[1][e] = P.r[2][e];
        }
        else if (ph == 2) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const T t = P.r[0][e];
                P.r[0][e] = P.r[2][e], P.r[1][e] = t;
            }
        }
    }
    // ---- accumulators -> LDS tile: half 0 holds the block entries 0..4, half 1 the entries 4..8 (its entry 4 is the duplicate)
    if ((lane & 31) < 27) {
        const int nrow = __popc(tk.m8);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r < nrow) { // wave-uniform
                const int row = qs[r];
                AT* o = tile + row * 1125 + ((tk.lx0 + (row >> 2) - ld.j0 + 2) * 25 + (tk.ly0 + ((row >> 1) & 1) - ld.j1 + 2) * 5 + (tk.lz0 + (row & 1) - ld.j2 + 2)) * 9 + 4 * ld.h;
                if (ld.h == 0) lds_atomic_add(o, (AT)acc[r][0]);
#pragma unroll
                for (int e = 1; e < 5; ++e) lds_atomic_add(o + e, (AT)acc[r][e]);
            }
        }
    }
}


// task candidates (cell << 1 | x-plane) ordered by the number of tile rows the plane has inside the cell's support: 4, then 2, then 1, then none
struct HrOrder {
    uint8_t v[128];
};
constexpr HrOrder hr_order()
{
    HrOrder o{};
    int n = 0;
    for (int want = 4; want >= 0; want = want == 4 ? 2 : (want == 2 ? 1 : (want == 1 ? 0 : -1))) {
        for (int t = 0; t < 128; ++t) {
            const int cell = t >> 1, px = t & 1, ox = (cell >> 4) - 2, oy = ((cell >> 2) & 3) - 2, oz = (cell & 3) - 2;
            const int rows = (px - ox >= 0 && px - ox < 3) ? ((oy == -1 || oy == 0) ? 2 : 1) * ((oz == -1 || oz == 0) ? 2 : 1) : 0;
            if (rows == want) o.v[n++] = (uint8_t)t;
        }
        if (want == 0) break;
    }
    return o;
}
// A task of the fp32 build = a whole cell (up to 8 rows, 40 accumulator registers): 8 instead of 12 visits per particle, 12.4 instead of 14.0 ms at C3.
// In fp64 the 80 accumulator registers cost a wavefront per SIMD (162 VGPRs) and the kernel is slower (C2: 6.3 against 5.35 ms), so fp64 keeps the
// (cell, x-plane) tasks.  HOT_HR_CELL_TASKS = 0 / 1 forces either for both types (A/B builds).
#ifndef HOT_HR_CELL_TASKS
#define HOT_HR_CELL_TASKS 2
#endif
template <class T>
constexpr bool kHrCellTasks = HOT_HR_CELL_TASKS == 2 ? sizeof(T) == 4 : HOT_HR_CELL_TASKS != 0;
// Candidates = cells (entries 64 .. 127 of the table are unused), ordered by rows 8, 4, 2, 1.
constexpr HrOrder hr_order_cells()
{
    HrOrder o{};
    int n = 0;
    for (int want = 8; want >= 1; want >>= 1)
        for (int cell = 0; cell < 64; ++cell) {
            const int ox = (cell >> 4) - 2, oy = ((cell >> 2) & 3) - 2, oz = (cell & 3) - 2;
            const int rows = ((ox == -1 || ox == 0) ? 2 : 1) * ((oy == -1 || oy == 0) ? 2 : 1) * ((oz == -1 || oz == 0) ? 2 : 1);
            if (rows == want) o.v[n++] = (uint8_t)cell;
        }
    return o;
}
__constant__ HrOrder kHrOrderTab = hr_order(), kHrOrderCellsTab = hr_order_cells();

// LDS of a workgroup: the tile (72 000 bytes in either build), per wavefront a record stage (128 scalars), the integer tables.  Two
// workgroups per CU: 7 wavefronts each in fp64 (81 024 bytes), 8 in fp32.
#ifndef HOT_HR_WAVES64
#define HOT_HR_WAVES64 7
#endif
template <class T>
struct RowsLds {
    static constexpr int WAVES = sizeof(T) == 8 ? HOT_HR_WAVES64 : 8, THREADS = WAVES * 64;
    static constexpr int NINT = 64 * 2 + 8 + 128 + 8; // cstart, ccnt | rdof | tasks | ctl
    static constexpr size_t bytes = (size_t)8 * 1125 * sizeof(AccT<T>) + (size_t)WAVES * REC * sizeof(T) + (size_t)NINT * sizeof(int32_t);
};

// Development aid (-DHOT_HT_CLOCKS, tools/hess_phases.sh): shader clocks of every wavefront (lane 0), summed per phase: 0 prologue, 1 task fetch
// and set-up, 2 particle loop and accumulators -> tile, 4 wait at the final barrier, 5 write-out; 6 particle visits, 7 (particle, row) steps.
#ifdef HOT_HT_CLOCKS
__device__ unsigned long long hr_clk[65536 * 8]; // per tile (mod 65536): same-address atomics of 3e5 wavefronts would cost more than the kernel
#define HR_CLK(i) \
    do { \
        const unsigned long long t_ = clock64(); /* wave-uniform: the sums stay in SGPRs */ \
        clk_[i] += t_ - t0_, t0_ = t_; \
    } while (0)
#define HR_CNT(i, n) clk_[i] += (n)
#else
#define HR_CLK(i)
#define HR_CNT(i, n)
#endif

template <class T>
__global__ __launch_bounds__(RowsLds<T>::THREADS) void k_hessian_rows(const T* __restrict__ rec, const int2* __restrict__ tcell, const int32_t* __restrict__ trow, const T* __restrict__ mass, T* __restrict__ val, int ntiles, const uint8_t* __restrict__ own /*sharded: rows this rank owns, else null*/,
    uint8_t* __restrict__ written /*sharded: rows this launch wrote*/)
{
    using G = Geo<T>;
    using AT = AccT<T>;
    constexpr int HR_WAVES = RowsLds<T>::WAVES, HR_THREADS = RowsLds<T>::THREADS;
    extern __shared__ __attribute__((aligned(16))) char hr_smem[];
    AT* tile = (AT*)hr_smem; // [8][1125]
    T* stages = (T*)(tile + 8 * 1125); // [HR_WAVES][REC]: the record of the wavefront's current particle
    int32_t* cstart = (int32_t*)(stages + HR_WAVES * REC); // [64] first particle of each contributing cell
    int32_t* ccnt = cstart + 64; // [64] its particle count
    int32_t* rdof = ccnt + 64; // [8]
    int32_t* tasks = rdof + 8; // [128] candidates with rows and particles, planes of four rows first: cell | plane << 6 | row mask << 8
    int32_t* ctl = tasks + 128; // [0] task cursor, [1] number of tasks
    const int tid = threadIdx.x;
#ifdef HOT_HT_CLOCKS
    unsigned long long clk_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t0_ = clock64();
#endif
    // workgroup i runs on XCD i % 8: runs of 32 consecutive tiles (4 - 8 SPGrid blocks) share most of their particle records; a run goes to
    // one XCD so that its L2 serves the re-reads
    const int id = blockIdx.x, run = (id & 7) + 8 * (id >> 8), tile_id = run * 32 + ((id >> 3) & 31);
    if (tile_id >= ntiles) return;
    if (tid < 8) rdof[tid] = trow[(int64_t)tile_id * 8 + tid];
    if (tid == 0) ctl[0] = 0, ctl[1] = 0;
    if (tid >= 64 && tid < 128) {
        const int2 fc = tcell[(int64_t)tile_id * 64 + (tid - 64)];
        cstart[tid - 64] = fc.x, ccnt[tid - 64] = fc.y;
    }
    for (int e = tid; e < 8 * 1125; e += HR_THREADS) tile[e] = (AT)0;
    __syncthreads();
    bool any = false;
    for (int r = 0; r < 8; ++r) any = any || rdof[r] >= 0;
    if (!any) return;
    // ---- task list: candidates (cell, x-plane) in the static order kHrOrderTab (cells: kHrOrderCellsTab) — four-row planes first, then two, then one — with the rows that
    // are inside the cell's 3x3x3 support AND active; those with rows and particles are compacted by the first wavefront (two candidates per
    // lane).  (A rank sort by rows x particles took a seventh of the kernel: 770 VALU instructions on two wavefronts beside a busy neighbour.)
    if (tid < 64) {
        int tk[2], nz[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int cell, m8 = 0;
            if (kHrCellTasks<T>) {
                cell = kHrOrderCellsTab.v[tid];
                const int ox = (cell >> 4) - 2, oy = ((cell >> 2) & 3) - 2, oz = (cell & 3) - 2;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (k == 0 && (unsigned)((r >> 2) - ox) < 3u && (unsigned)(((r >> 1) & 1) - oy) < 3u && (unsigned)((r & 1) - oz) < 3u && rdof[r] >= 0) m8 |= 1 << r;
            }
            else {
                const int t = kHrOrderTab.v[64 * k + tid], px = t & 1;
                cell = t >> 1;
                const int ox = (cell >> 4) - 2, oy = ((cell >> 2) & 3) - 2, oz = (cell & 3) - 2;
                if ((unsigned)(px - ox) < 3u) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((unsigned)((q >> 1) - oy) < 3u && (unsigned)((q & 1) - oz) < 3u && rdof[4 * px + q] >= 0) m8 |= 1 << (4 * px + q);
                }
            }
            tk[k] = cell | (m8 << 8), nz[k] = m8 != 0 && ccnt[cell] > 0;
        }
        const unsigned long long b0 = __ballot(nz[0]), b1 = __ballot(nz[1]), below = (1ull << tid) - 1ull;
        const int n0 = __popcll(b0);
        if (nz[0]) tasks[__popcll(b0 & below)] = tk[0];
        if (nz[1]) tasks[n0 + __popcll(b1 & below)] = tk[1];
        if (tid == 0) ctl[1] = n0 + __popcll(b1);
    }
    __syncthreads();
    if (own) { // sharded: a tile none of whose rows this rank owns and none of whose cells hold particles of its shard is not its business
        bool mine = false;
        for (int r = 0; r < 8; ++r) mine = mine || (rdof[r] >= 0 && own[rdof[r]]);
        if (!mine && ctl[1] == 0) return; // workgroup-uniform (LDS tables)
    }
    const int ntask = ctl[1];
    HR_CLK(0);
    // ---- lane roles
    const int lane = tid & 63, wv = tid >> 6;
    const RowsLane ld = rows_lane(lane);
    T* stage = stages + wv * REC;
    // a wavefront knows its next task while it walks the current one (the request stream above runs on into it)
    auto draw = [&]() {
        int k = 0;
        if (lane == 0) k = atomicAdd(ctl, 1);
        return __builtin_amdgcn_readfirstlane(k);
    };
    auto task_range = [&](int k, int& task, const T*& rp, int& cnt) {
        task = __builtin_amdgcn_readfirstlane(tasks[k < ntask ? k : 0]);
        const int cell = task & 63;
        rp = rec + (int64_t) __builtin_amdgcn_readfirstlane(cstart[cell]) * REC, cnt = __builtin_amdgcn_readfirstlane(ccnt[cell]);
    };
    int k = draw();
    if (k < ntask) {
        int task, cnt;
        const T* rp;
        task_range(k, task, rp, cnt);
        RowsPre<T> P;
        P.r[0][0] = rp[lane], P.r[0][1] = rp[64 + lane];
        {
            const T* q = rp + (cnt > 1 ? REC : 0);
            P.r[1][0] = q[lane], P.r[1][1] = q[64 + lane];
        }
        while (true) {
            const int kn = draw();
            int ntk = task, ncnt = 1;
            const T* nrp = rp; // no next task: the stream ends on a record that is there
            if (kn < ntask) task_range(kn, ntk, nrp, ncnt);
            const int m8 = task >> 8, nr = __popc(m8), cell = task & 63;
            HR_CNT(6, cnt);
            HR_CNT(7, cnt * nr);
            const RowsTask tk = { -((cell >> 4) - 2), -(((cell >> 2) & 3) - 2), -((cell & 3) - 2), m8 };
            // the particle walk is compiled for 1, 2, 4 (and 8) rows; a count between — an inactive node — runs as the next size with rows
            // computed and dropped
            if (nr == 1)
                hr_walk<T, 1>(rp, cnt, nrp, ncnt, P, tk, stage, tile, lane, ld);
            else if (nr == 2)
                hr_walk<T, 2>(rp, cnt, nrp, ncnt, P, tk, stage, tile, lane, ld);
            else if (!kHrCellTasks<T> || nr <= 4)
                hr_walk<T, 4>(rp, cnt, nrp, ncnt, P, tk, stage, tile, lane, ld);
            else
                hr_walk<T, kHrCellTasks<T> ? 8 : 4>(rp, cnt, nrp, ncnt, P, tk, stage, tile, lane, ld);
            if (kn >= ntask) break;
            k = kn, task = ntk, rp = nrp, cnt = ncnt;
        }
    }
    HR_CLK(2);
    __syncthreads();
    HR_CLK(4);
    for (int e = tid; e < 8 * 1125; e += HR_THREADS) {
        int r = e / 1125, q = e - r * 1125;
        int dof = rdof[r];
        if (dof < 0) continue;
        T v = (T)tile[e];
        if (q >= 62 * 9 && q < 63 * 9 && ((q - 62 * 9) % 4 == 0) && (!own || own[dof])) v += mass[dof]; // inertia term on the diagonal slot (ImplicitSolver.h:486-496)
        val[(int64_t)dof * 1125 + q] = v;
        if (written && q == 0) written[dof] = 1;
    }
#ifdef HOT_HT_CLOCKS
    HR_CLK(5);
    if ((tid & 63) == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&hr_clk[(tile_id & 65535) * 8 + i], clk_[i]);
#endif
}

template <class T>
void Ctx<T>::assemble_rows(Level<T>& L)
{
    pDP.reserve((size_t)REC * (size_t)Np);
    HOT_LAUNCH(this, "hessian_dpdf", k_dpdf_rec<T>, div_up(Np, 256), 256, 0, pX.p, pFn.p, pFt.p, pVol.p, pMu.p, pLam.p, pDP.p, Np, dt, (T)1 / dx, cfg.project);
    if (!attr_rows_set) {
        HOT_HIP(hipFuncSetAttribute((const void*)k_hessian_rows<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RowsLds<T>::bytes));
        attr_rows_set = true;
    }
    constexpr int TPB = (G::BX / 2) * (G::BY / 2) * (G::BZ / 2);
    if (L.part) {
        written.reserve(Nn);
        HOT_HIP(hipMemsetAsync(written.p, 0, Nn, stream));
    }
    const int ntiles = Nb * TPB;
    tile_tab.reserve((size_t)ntiles * (2 * 64 + 8));
    int2* tcell = (int2*)tile_tab.p;
    int32_t* trow = tile_tab.p + (size_t)ntiles * 128;
    HOT_LAUNCH(this, "hessian_tile_cells", k_tile_cells<T>, div_up((int64_t)ntiles * 64, 256), 256, 0, blocks.p, gIdx.p, cell_first.p, cell_map, tcell, trow, ntiles);
    HOT_LAUNCH(this, "hessian_assemble", k_hessian_rows<T>, 256 * div_up(ntiles, 256), RowsLds<T>::THREADS, RowsLds<T>::bytes, pDP.p, tcell, trow, mass.p, L.val.p, ntiles, L.mask(),
        L.part ? written.p : (uint8_t*)nullptr);
#ifdef HOT_HT_CLOCKS
    std::vector<unsigned long long> hall(65536 * 8);
    unsigned long long hc[8] = {};
    HOT_HIP(hipStreamSynchronize(stream));
    HOT_HIP(hipMemcpyFromSymbol(hall.data(), HIP_SYMBOL(hr_clk), hall.size() * 8));
    for (size_t i = 0; i < hall.size(); ++i) hc[i & 7] += hall[i];
    const double tiles = (double)Nb * TPB;
    const double wv = tiles * RowsLds<T>::WAVES;
    fprintf(stderr, "hessian row tiles, clocks per wavefront: prologue %.0f fetch %.0f particle loop + flush %.0f (%.0f) barrier %.0f write-out %.0f; per workgroup %.0f visits, %.0f row steps; %.0f clocks per visit\n", hc[0] / wv, hc[1] / wv,
        hc[2] / wv, hc[3] / wv, hc[4] / wv, hc[5] / wv, hc[6] / tiles, hc[7] / tiles, (double)hc[2] / (double)(hc[6] ? hc[6] : 1));
    std::fill(hall.begin(), hall.end(), 0ull);
    HOT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(hr_clk), hall.data(), hall.size() * 8));
#endif
}

template void Ctx<float>::assemble_rows(Level<float>&);
template void Ctx<double>::assemble_rows(Level<double>&);

} // namespace hot
