// libhotmi355x — Galerkin multigrid hierarchy on the device.
//
// Replaces MultigridBuilder::build (reference Projects/multigrid/MultigridPreconditioner.h:554-703), which is serial
// and std::unordered_map based:
//   coarse node set + numbering (:619-664)  first-touch order over (fine id, 2x2x2 parent loop) == rank of the first
//                                           (i*8 + linear_idx) candidate that names a coarse coordinate: hash insert with
//                                           atomicMin(rank), flag first occurrences, exclusive scan  -> bit-exact ids
//   P (8 slots/row, trilinear, :445-466)    k_build_P, same slot layout and padding rule (:647-651)
//   R = P^T (SquareMatrix.h:573-607)        child table of each coarse node (<= 27 fine children), weights 1/.5 per axis
//   A_{l+1} = R (A_l P) (:526-571)          stencil collapse without hash maps: every level keeps the 125-slot stencil
//                                           layout, so  AP[i, J] (4^3 coarse window per fine row, k_ap) and
//                                           RAP[I, slot] (k_rap) are pure gathers — no atomics, deterministic
//   markColors (:582-605)                   4^3-node blocks, colour = parity bits, first-touch block ids per colour,
//                                           1-based index in block; the GS processing order is the radix-sorted
//                                           (colour, block, id) key, so smoothers walk contiguous segments
#include "hot_impl.h"
#include "hot_svd.h"
#include <rocprim/rocprim.hpp>

namespace hot {

__global__ void k_hash_clear2(HashMap h)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > h.mask) return;
    h.keys[i] = ~0ULL;
    h.minrank[i] = ~0ULL;
    h.id[i] = -1;
}
__global__ void k_coord_map_insert(HashMap h, const int32_t* __restrict__ coord, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = hash_insert_min(h, coord_key(coord[3 * i], coord[3 * i + 1], coord[3 * i + 2]), (unsigned long long)i);
    h.id[s] = i;
}

// candidate s = i*8 + linear_idx  (linear_idx = (dx*4 + dy*2 + dz), parent = coord/2 + d) with non-zero weight
__device__ __forceinline__ bool parent_of(const int32_t* __restrict__ coord, int s, int& px, int& py, int& pz)
{
    int i = s >> 3, l = s & 7;
    int x = coord[3 * i], y = coord[3 * i + 1], z = coord[3 * i + 2];
    int dx = l >> 2, dy = (l >> 1) & 1, dz = l & 1;
    if ((dx && !(x & 1)) || (dy && !(y & 1)) || (dz && !(z & 1))) return false; // weight 0 (even coordinate has one parent)
    px = x / 2 + dx, py = y / 2 + dy, pz = z / 2 + dz;
    return true;
}
__global__ void k_coarse_insert(HashMap h, const int32_t* __restrict__ coord, int n)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n * 8) return;
    int px, py, pz;
    if (parent_of(coord, s, px, py, pz)) hash_insert_min(h, coord_key(px, py, pz), (unsigned long long)s);
}
__global__ void k_coarse_flag(HashMap h, const int32_t* __restrict__ coord, int32_t* flags, int n)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n * 8) return;
    int px, py, pz;
    int f = 0;
    if (parent_of(coord, s, px, py, pz)) {
        int32_t slot = hash_find_slot(h, coord_key(px, py, pz));
        f = h.minrank[slot] == (unsigned long long)s;
    }
    flags[s] = f;
}
__global__ void k_coarse_assign(HashMap h, const int32_t* __restrict__ coord, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int32_t* ccoord, int n)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n * 8 || !flags[s]) return;
    int px, py, pz;
    parent_of(coord, s, px, py, pz);
    int id = scan[s];
    h.id[hash_find_slot(h, coord_key(px, py, pz))] = id;
    ccoord[3 * id] = px, ccoord[3 * id + 1] = py, ccoord[3 * id + 2] = pz;
}
// P row: 8 slots, padding repeats slot 0's column with weight 0 (MultigridPreconditioner.h:647-651)
template <class T>
__global__ void k_build_P(HashMap cmap, const int32_t* __restrict__ coord, int32_t* pcol, T* pw, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x = coord[3 * i], y = coord[3 * i + 1], z = coord[3 * i + 2];
    int first = hash_find_id(cmap, coord_key(x / 2, y / 2, z / 2));
    for (int l = 0; l < 8; ++l) {
        int dx = l >> 2, dy = (l >> 1) & 1, dz = l & 1;
        T wx = (x & 1) ? (T)0.5 : (dx ? (T)0 : (T)1), wy = (y & 1) ? (T)0.5 : (dy ? (T)0 : (T)1), wz = (z & 1) ? (T)0.5 : (dz ? (T)0 : (T)1);
        T w = wx * wy * wz;
        int c = first;
        if (w != (T)0) c = hash_find_id(cmap, coord_key(x / 2 + dx, y / 2 + dy, z / 2 + dz));
        pcol[8 * (int64_t)i + l] = c;
        pw[8 * (int64_t)i + l] = w;
    }
}
// children of coarse node I: fine coords 2I + (a-1, b-1, c-1)
__global__ void k_build_children(HashMap fmap, const int32_t* __restrict__ ccoord, int32_t* child, int nc)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nc * 27) return;
    int I = e / 27, q = e - I * 27;
    int a = q / 9 - 1, b = (q / 3) % 3 - 1, c = q % 3 - 1;
    int x = 2 * ccoord[3 * I] + a, y = 2 * ccoord[3 * I + 1] + b, z = 2 * ccoord[3 * I + 2] + c;
    child[e] = (x | y | z) < 0 ? -1 : hash_find_id(fmap, coord_key(x, y, z));
}
__global__ void k_coarse_cols(HashMap cmap, const int32_t* __restrict__ ccoord, int32_t* col, int nc)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)nc * 125) return;
    int n = (int)(e / 125), k = (int)(e - (int64_t)n * 125);
    int x = ccoord[3 * n] - (k / 25 - 2), y = ccoord[3 * n + 1] - ((k / 5) % 5 - 2), z = ccoord[3 * n + 2] - (k % 5 - 2);
    int j = (x | y | z) < 0 ? -1 : hash_find_id(cmap, coord_key(x, y, z));
    col[e] = j >= 0 ? j : (n > 0 ? 0 : 1);
}

// AP[i][(a,b,c) in the coarse window][9]: coarse window origin cb = (coord - 2) >> 1 ;  J = cb + (a,b,c)
// AP[i,J] = sum_{d in {-1,0,1}^3} w(d) A[i][slot(i - (2J + d))]
// Round 6: the window is PACKED.  Along an axis on which the fine node's coordinate is even the window holds 3 coarse nodes, not 4 (x = 2 m: the fine
// neighbours x - 2 .. x + 2 interpolate from m - 1, m, m + 1; position a = 3 is structurally zero), so a row has na nb nc = 27 .. 64 (42.9 on average)
// non-zero slots, stored first: slot p = (a nb + b) nc + c.  k_apmv_sub, which runs once per V-cycle and level, reads those only — a third fewer bytes;
// the rest of the 64-slot row is never written nor read.
__device__ __forceinline__ void ap_window(int x, int y, int z, int& na, int& nb, int& nc) { na = 3 + (x & 1), nb = 3 + (y & 1), nc = 3 + (z & 1); }
template <class T>
__global__ __launch_bounds__(256) void k_ap(const int32_t* __restrict__ coord, const T* __restrict__ val, T* ap, int n, const uint8_t* __restrict__ own)
{
    __shared__ T row[4][1125];
    int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int i = blockIdx.x * 4 + w;
    if (own && i < n && !own[i]) i = n; // sharded: rows of other ranks (wave-uniform; the barrier below is still reached)
    if (i < n) {
        const T* src = val + (int64_t)i * 1125;
        for (int e = lane; e < 1125; e += 64) row[w][e] = src[e];
    }
    __syncthreads();
    if (i >= n) return;
    int x = coord[3 * i], y = coord[3 * i + 1], z = coord[3 * i + 2];
    int cbx = (x - 2) >> 1, cby = (y - 2) >> 1, cbz = (z - 2) >> 1;
    int na, nb, nc;
    ap_window(x, y, z, na, nb, nc);
    const int cnt9 = na * nb * nc * 9;
    for (int o = lane; o < cnt9; o += 64) {
        int js = o / 9, comp = o - js * 9;
        int a = js / (nb * nc), b = (js / nc) % nb, c = js % nc;
        int Jx = cbx + a, Jy = cby + b, Jz = cbz + c;
        T sum = (T)0;
#pragma unroll
        for (int da = -1; da <= 1; ++da) {
            int ddx = x - (2 * Jx + da); // i - j
            if (ddx < -2 || ddx > 2) continue;
#pragma unroll
            for (int db = -1; db <= 1; ++db) {
                int ddy = y - (2 * Jy + db);
                if (ddy < -2 || ddy > 2) continue;
#pragma unroll
                for (int dc = -1; dc <= 1; ++dc) {
                    int ddz = z - (2 * Jz + dc);
                    if (ddz < -2 || ddz > 2) continue;
                    T wgt = (da ? (T)0.5 : (T)1) * (db ? (T)0.5 : (T)1) * (dc ? (T)0.5 : (T)1);
                    sum += wgt * row[w][((ddx + 2) * 25 + (ddy + 2) * 5 + (ddz + 2)) * 9 + comp];
                }
            }
        }
        ap[(int64_t)i * 576 + o] = sum;
    }
}
// coarse column ids of the AP window, by GEOMETRIC window position js = 16 a + 4 b + c: -1 at the positions that are structurally zero (not stored: the
// packed slot of a stored position is its rank among the row's stored positions, in this order), 0 where a coarse node of the window does not exist (its
// block is 0)
__global__ void k_ap_cols(HashMap cmap, const int32_t* __restrict__ coord, int32_t* apc, int n)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * 64) return;
    int i = (int)(e >> 6), js = (int)(e & 63);
    int na, nb, nc;
    ap_window(coord[3 * i], coord[3 * i + 1], coord[3 * i + 2], na, nb, nc);
    if ((js >> 4) >= na || ((js >> 2) & 3) >= nb || (js & 3) >= nc) {
        apc[e] = -1;
        return;
    }
    int Jx = ((coord[3 * i] - 2) >> 1) + (js >> 4), Jy = ((coord[3 * i + 1] - 2) >> 1) + ((js >> 2) & 3), Jz = ((coord[3 * i + 2] - 2) >> 1) + (js & 3);
    int j = (Jx | Jy | Jz) < 0 ? -1 : hash_find_id(cmap, coord_key(Jx, Jy, Jz));
    apc[e] = j >= 0 ? j : 0;
}
// RAP[I][slot k][9] = sum_{d} w(d) AP[child(I,d)][J - cb(child)] ,  J = I - Delta(k)
// One workgroup per coarse row.  The 27 children's A P rows (64 window slots x 9 scalars each) are staged in LDS nine at a time with
// coalesced loads (the first version gathered 8 bytes per lane and child straight from global memory: C2 2.29 ms per step for both levels, now 1.95;
// giving every XCD a contiguous run of coarse rows, for L2 reuse of the shared children, changed nothing); which window slot of child d holds coarse column k does not depend on the row:
// per axis, slot = 3 - k_axis + (d_axis < 0), valid if within 0..3.  The sum keeps the child order.
template <class T>
__global__ __launch_bounds__(256) void k_rap(const int32_t* __restrict__ ccoord, const int32_t* __restrict__ child, const T* __restrict__ ap, T* cval, int nc,
    const uint8_t* __restrict__ fine_own /*sharded: only the fine rows this rank owns contribute (partial sums), else null*/)
{
    __shared__ T sap[9 * 576];
    __shared__ int32_t sci[27];
    const int tid = threadIdx.x, I = blockIdx.x;
    if (tid < 27) {
        const int ci = child[I * 27 + tid];
        sci[tid] = (ci < 0 || (fine_own && !fine_own[ci])) ? -1 : ci;
    }
    int kx[5], ky[5], kz[5], comp[5];
    T sum[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int o = min(tid + 256 * t, 1124), k = o / 9;
        comp[t] = o - 9 * k, kx[t] = k / 25, ky[t] = (k / 5) % 5, kz[t] = k % 5, sum[t] = (T)0;
    }
    __syncthreads();
#pragma unroll
    for (int batch = 0; batch < 3; ++batch) {
        for (int e = tid; e < 9 * 576; e += 256) {
            const int q = 9 * batch + e / 576, ci = sci[q];
            const int cnt9 = (3 + (q / 9 != 1)) * (3 + ((q / 3) % 3 != 1)) * (3 + (q % 3 != 1)) * 9; // the child's packed window (child = 2 I + d: odd along the axes with d != 0)
            if (e % 576 < cnt9) sap[e] = ci >= 0 ? ap[(int64_t)ci * 576 + e % 576] : (T)0;
        }
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 9; ++qq) {
            const int q = 9 * batch + qq, da = q / 9 - 1, db = (q / 3) % 3 - 1, dc = q % 3 - 1;
            if (sci[q] < 0) continue; // workgroup-uniform
            const T wgt = (da ? (T)0.5 : (T)1) * (db ? (T)0.5 : (T)1) * (dc ? (T)0.5 : (T)1);
            const int na = 3 + (da != 0), nb = 3 + (db != 0), nc = 3 + (dc != 0); // the child's packed window (k_ap)
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int a = 3 - kx[t] + (da < 0), bb = 3 - ky[t] + (db < 0), c = 3 - kz[t] + (dc < 0);
                if ((unsigned)a >= (unsigned)na || (unsigned)bb >= (unsigned)nb || (unsigned)c >= (unsigned)nc) continue; // (position 3 of an even axis: structurally zero, not stored)
                sum[t] += wgt * sap[qq * 576 + ((a * nb + bb) * nc + c) * 9 + comp[t]];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 5; ++t)
        if (tid + 256 * t < 1125) cval[(int64_t)I * 1125 + tid + 256 * t] = sum[t];
}

// ---- colouring
__global__ void k_color_insert(HashMap h, const int32_t* __restrict__ coord, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hash_insert_min(h, coord_key(coord[3 * i] >> 2, coord[3 * i + 1] >> 2, coord[3 * i + 2] >> 2), (unsigned long long)i);
}
__device__ __forceinline__ int color_of(const int32_t* __restrict__ coord, int i)
{
    return (((coord[3 * i] >> 2) & 1) << 2) | (((coord[3 * i + 1] >> 2) & 1) << 1) | ((coord[3 * i + 2] >> 2) & 1);
}
__global__ void k_color_flag(HashMap h, const int32_t* __restrict__ coord, int32_t* flags, int n)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 8 * n) return;
    int c = e / n, i = e - c * n;
    int f = 0;
    if (color_of(coord, i) == c) {
        int32_t slot = hash_find_slot(h, coord_key(coord[3 * i] >> 2, coord[3 * i + 1] >> 2, coord[3 * i + 2] >> 2));
        f = h.minrank[slot] == (unsigned long long)i;
    }
    flags[e] = f;
}
__global__ void k_color_assign(HashMap h, const int32_t* __restrict__ coord, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int n)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 8 * n || !flags[e]) return;
    int c = e / n, i = e - c * n;
    int32_t slot = hash_find_slot(h, coord_key(coord[3 * i] >> 2, coord[3 * i + 1] >> 2, coord[3 * i + 2] >> 2));
    h.id[slot] = scan[e] - scan[c * n]; // block id inside its colour, first-touch order
}
__global__ void k_color_keys(HashMap h, const int32_t* __restrict__ coord, uint64_t* keys, uint32_t* vals, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t bid = hash_find_id(h, coord_key(coord[3 * i] >> 2, coord[3 * i + 1] >> 2, coord[3 * i + 2] >> 2));
    keys[i] = ((uint64_t)color_of(coord, i) << 56) | ((uint64_t)(uint32_t)bid << 32) | (uint32_t)i;
    vals[i] = (uint32_t)i;
}
__global__ void k_color_heads(const uint64_t* __restrict__ keys, int32_t* flags, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    flags[p] = (p == 0 || (keys[p] >> 32) != (keys[p - 1] >> 32)) ? 1 : 0;
}
__global__ void k_color_finish(const uint64_t* __restrict__ keys, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int32_t* gs_order, int32_t* block_start,
    int32_t* color_begin, int n, int nblocks)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    gs_order[p] = (int32_t)(keys[p] & 0xffffffffu);
    if (p == 0) block_start[nblocks] = n;
    if (flags[p]) {
        int b = scan[p];
        block_start[b] = p;
        int c = (int)(keys[p] >> 56);
        int cprev = p == 0 ? -1 : (int)(keys[p - 1] >> 56);
        for (int cc = cprev + 1; cc <= c; ++cc) color_begin[cc] = b;
    }
    if (p == n - 1) {
        int c = (int)(keys[p] >> 56);
        for (int cc = c + 1; cc <= 8; ++cc) color_begin[cc] = nblocks;
    }
}
__global__ void k_color_ckey(const uint64_t* __restrict__ keys, const int32_t* __restrict__ scan, const int32_t* __restrict__ block_start, uint32_t* ckey, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int node = (int)(keys[p] & 0xffffffffu);
    // block rank of position p = number of heads at or before p, minus one
    // (scan is the exclusive scan of head flags, so rank = scan[p] + flag[p] - 1; recomputed from block_start)
    int lo = 0, hi = 0;
    (void)lo, (void)hi;
    uint32_t color = (uint32_t)(keys[p] >> 56), bid = (uint32_t)((keys[p] >> 32) & 0xffffff);
    int b = scan[p]; // exclusive count of heads before p
    int start = block_start[b] == p ? p : block_start[b - 1];
    ckey[node] = (color << 28) | (bid << 7) | (uint32_t)(p - start + 1);
}

// the 26 colour blocks around block b (block coordinates +-1): global block id | colour << 28, or -1.  A row of b only
// couples to nodes of b and of these blocks (stencil reach 2 < block edge 4).
__global__ void k_block_neighbours(HashMap h, const int32_t* __restrict__ coord, const int32_t* __restrict__ gs_order, const int32_t* __restrict__ block_start,
    const int32_t* __restrict__ color_begin, int32_t* __restrict__ nbr, int nblocks)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nblocks * 26) return;
    const int b = e / 26;
    int q = e - b * 26;
    if (q >= 13) ++q; // skip the centre
    const int node = gs_order[block_start[b]];
    const int x = (coord[3 * node] >> 2) + q / 9 - 1, y = (coord[3 * node + 1] >> 2) + (q / 3) % 3 - 1, z = (coord[3 * node + 2] >> 2) + q % 3 - 1;
    int32_t out = -1;
    if ((x | y | z) >= 0) {
        const int32_t bid = hash_find_id(h, coord_key(x, y, z));
        if (bid >= 0) {
            const int c = ((x & 1) << 2) | ((y & 1) << 1) | (z & 1);
            out = (color_begin[c] + bid) | (c << 28);
        }
    }
    nbr[e] = out;
}

template <class T>
static void build_coord_map(Ctx<T>* ctx, Level<T>& L)
{
    uint32_t cap = 1024;
    while (cap < 2u * (uint32_t)L.n + 16u) cap <<= 1;
    L.hkeys.reserve(cap), L.hrank.reserve(cap), L.hid.reserve(cap);
    L.map.keys = L.hkeys.p, L.map.minrank = L.hrank.p, L.map.id = L.hid.p, L.map.mask = cap - 1;
    HOT_LAUNCH(ctx, "mg_hash_clear", k_hash_clear2, div_up(cap, 256), 256, 0, L.map);
    HOT_LAUNCH(ctx, "mg_coord_map", k_coord_map_insert, div_up(L.n, 256), 256, 0, L.map, L.coord.p, L.n);
}

// ---- sharded runs: owner of every colour block and the order of a colour's blocks (owner, then first touch), so that every rank owns a
// contiguous run of each colour's list whatever the ownership rule (hot_config.shard_owner):
//   1  the rank whose id prefix holds the block's lowest node, i.e. whose particles first touch it (rounds 2 - 4: the lower rank of a cut owned
//      every block the two share and received every partial matrix row);
//   0  finest level: the rank whose particle range (pages of the SPGrid page order) contains the block's own page, home_rank (hot_impl.h), if its
//      particles reach the block at all, else the reaching rank nearest to it in rank order — every rank owns the rows inside its own range, the boundary between the
//      ranks' rows is the boundary between their particles, both sides of a cut send the same amount of partial rows; coarser levels: the owner
//      of the first existing child of the block's lowest node.
// (Measured, DESIGN.md section 7: a rule that lets a rank own blocks none of its particles reach scatters single blocks over far ranks — the page
// order jumps — and a rank-local Gauss-Seidel sweep, hot_config.shard_gs = 1, then needs twice the iterations; alternating the owner along a cut
// in patches balances as well but lengthens the boundary: +13 .. +38 % iterations.)
struct RankPrefix {
    int v[65];
    uint64_t split[63];
    int R;
};
template <class T>
__global__ void k_color_owner_keys(HashMap h, const int32_t* __restrict__ coord, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, RankPrefix np, int mode /*0 first touch, 1 home page (finest level), 2 first child's owner*/,
    HashMap block_map, const uint64_t* __restrict__ sharers, const int32_t* __restrict__ child, const uint8_t* __restrict__ fine_owner, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int n)
{
    using G = Geo<T>;
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 8 * n || !flags[e]) return;
    const int c = e / n, i = e - c * n; // i: the block's lowest node
    const int x = coord[3 * i], y = coord[3 * i + 1], z = coord[3 * i + 2];
    const int32_t slot = hash_find_slot(h, coord_key(x >> 2, y >> 2, z >> 2));
    int r = 0;
    while (r + 1 < np.R && i >= np.v[r + 1]) ++r; // the first-touching rank
    if (mode == 1) {
        const int hr = home_rank<T>(np.split, np.R, x, y, z, 0);
        // does rank hr reach the block?  (sharers of an SPGrid block: the ranks whose particle tiles cover it or, fp64, its x-companion inside the colour block)
        int32_t b = hash_find_id(block_map, G::linear_offset(x & ~3, y & ~3, z & ~3) >> 12);
        if (b < 0 && G::BX == 2) b = hash_find_id(block_map, G::linear_offset((x & ~3) + 2, y & ~3, z & ~3) >> 12);
        if (b >= 0 && sharers[b]) { // the reaching rank nearest to hr in rank (= page) order, the lower one on a tie: hr itself wherever it reaches the block
            const uint64_t sh = sharers[b];
            for (int d = 0; d < np.R; ++d) {
                if (hr - d >= 0 && ((sh >> (hr - d)) & 1ULL)) {
                    r = hr - d;
                    break;
                }
                if (hr + d < np.R && ((sh >> (hr + d)) & 1ULL)) {
                    r = hr + d;
                    break;
                }
            }
        }
    }
    else if (mode == 2) {
        for (int q = 0; q < 27; ++q) {
            const int ci = child[(int64_t)i * 27 + q];
            if (ci >= 0) {
                r = fine_owner[ci];
                break;
            }
        }
    }
    keys[scan[e]] = ((uint64_t)c << 56) | ((uint64_t)r << 48) | (uint32_t)i;
    vals[scan[e]] = (uint32_t)slot;
}
__global__ void k_color_owner_assign(HashMap h, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ slots, const int32_t* __restrict__ scan, uint8_t* __restrict__ owner, int nb, int n)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nb) return;
    const int c = (int)(keys[p] >> 56);
    h.id[slots[p]] = p - scan[(size_t)c * n]; // block id inside its colour: position among the colour's blocks in (owner, first touch) order
    owner[p] = (uint8_t)((keys[p] >> 48) & 0xff);
}
template <class T>
static void mark_colors(Ctx<T>* ctx, Level<T>& L)
{
    int n = L.n;
    // block map (temporary)
    uint32_t cap = 1024;
    while (cap < 2u * (uint32_t)n + 16u) cap <<= 1;
    // (context-owned scratch: a local buffer would cost a hipMalloc and a synchronising hipFree per level and time step)
    DBuf<uint64_t>& hk = ctx->col_hk;
    DBuf<unsigned long long>& hr = ctx->col_hr;
    DBuf<int32_t>& hi = ctx->col_hi;
    hk.reserve(cap), hr.reserve(cap), hi.reserve(cap);
    HashMap h{ hk.p, hr.p, hi.p, cap - 1 };
    HOT_LAUNCH(ctx, "mg_hash_clear", k_hash_clear2, div_up(cap, 256), 256, 0, h);
    HOT_LAUNCH(ctx, "color_block_insert", k_color_insert, div_up(n, 256), 256, 0, h, L.coord.p, n);
    ctx->flags.reserve(8 * (size_t)n), ctx->scan.reserve(8 * (size_t)n);
    HOT_LAUNCH(ctx, "color_block_flag", k_color_flag, div_up(8 * (size_t)n, 256), 256, 0, h, L.coord.p, ctx->flags.p, n);
    const int nb_all = ctx->exclusive_scan_i32(ctx->flags.p, ctx->scan.p, 8 * (size_t)n);
    ctx->keys.reserve(n), ctx->keys2.reserve(n), ctx->vals.reserve(n), ctx->vals2.reserve(n);
    L.block_owner_h.clear();
    if (ctx->sharded() && (int)L.nstart.size() == ctx->comm.size + 1 && nb_all > 0) {
        // block ids inside a colour in (owner, first touch) order; the owners go to level_ownership
        RankPrefix np{};
        np.R = ctx->comm.size;
        for (int r = 0; r <= np.R; ++r) np.v[r] = L.nstart[r];
        int mode = 0;
        const uint8_t* fine_owner = nullptr;
        // 0 = by the sweep: page-range ownership under colour-synchronous sweeps; first touch under rank-local sweeps, which hold their iteration bound only under it
        const bool page_owner = ctx->cfg.shard_owner == 2 || (ctx->cfg.shard_owner == 0 && ctx->cfg.shard_gs == 0);
        if (page_owner) {
            if (L.id == 0 && (int)ctx->page_split.size() == np.R - 1 && ctx->halo_mode() && ctx->sharers.p) {
                mode = 1;
                for (int r = 0; r < np.R - 1; ++r) np.split[r] = ctx->page_split[r];
            }
            else if (L.id > 0 && L.id - 1 < (int)ctx->levels.size() && ctx->levels[L.id - 1]->part && L.child.p) {
                mode = 2;
                fine_owner = ctx->levels[L.id - 1]->owner.p;
            }
        }
        HOT_LAUNCH(ctx, "color_owner_keys", k_color_owner_keys<T>, div_up(8 * (size_t)n, 256), 256, 0, h, L.coord.p, ctx->flags.p, ctx->scan.p, np, mode, ctx->block_map, ctx->sharers.p, L.child.p, fine_owner, ctx->keys.p, ctx->vals.p, n);
        size_t sb = 0;
        HOT_HIP(rocprim::radix_sort_pairs(nullptr, sb, ctx->keys.p, ctx->keys2.p, ctx->vals.p, ctx->vals2.p, (size_t)nb_all, 0, 64, ctx->stream));
        if (sb > ctx->sort_tmp_bytes) {
            ctx->sort_tmp.reserve(sb);
            ctx->sort_tmp_bytes = ctx->sort_tmp.cap;
        }
        HOT_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, sb, ctx->keys.p, ctx->keys2.p, ctx->vals.p, ctx->vals2.p, (size_t)nb_all, 0, 64, ctx->stream));
        L.block_owner.reserve(nb_all);
        HOT_LAUNCH(ctx, "color_owner_assign", k_color_owner_assign, div_up(nb_all, 256), 256, 0, h, ctx->keys2.p, ctx->vals2.p, ctx->scan.p, L.block_owner.p, nb_all, n);
        L.block_owner_h.resize(nb_all);
        HOT_HIP(hipMemcpyAsync(L.block_owner_h.data(), L.block_owner.p, nb_all, hipMemcpyDeviceToHost, ctx->stream)); // (the sync at the end of this function)
    }
    else
        HOT_LAUNCH(ctx, "color_block_assign", k_color_assign, div_up(8 * (size_t)n, 256), 256, 0, h, L.coord.p, ctx->flags.p, ctx->scan.p, n);
    HOT_LAUNCH(ctx, "color_keys", k_color_keys, div_up(n, 256), 256, 0, h, L.coord.p, ctx->keys.p, ctx->vals.p, n);
    size_t bytes = 0;
    HOT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, ctx->keys.p, ctx->keys2.p, ctx->vals.p, ctx->vals2.p, (size_t)n, 0, 64, ctx->stream));
    if (bytes > ctx->sort_tmp_bytes) {
        ctx->sort_tmp.reserve(bytes);
        ctx->sort_tmp_bytes = ctx->sort_tmp.cap;
    }
    HOT_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp.p, bytes, ctx->keys.p, ctx->keys2.p, ctx->vals.p, ctx->vals2.p, (size_t)n, 0, 64, ctx->stream));
    HOT_LAUNCH(ctx, "color_heads", k_color_heads, div_up(n, 256), 256, 0, ctx->keys2.p, ctx->flags.p, n);
    L.nblocks = ctx->exclusive_scan_i32(ctx->flags.p, ctx->scan.p, n);
    L.gs_order.reserve(n), L.gs_block_start.reserve(L.nblocks + 1), L.ckey.reserve(n);
    DBuf<int32_t>& cb = ctx->col_cb;
    cb.reserve(16);
    HOT_LAUNCH(ctx, "color_finish", k_color_finish, div_up(n, 256), 256, 0, ctx->keys2.p, ctx->flags.p, ctx->scan.p, L.gs_order.p, L.gs_block_start.p, cb.p, n, L.nblocks);
    HOT_LAUNCH(ctx, "color_ckey", k_color_ckey, div_up(n, 256), 256, 0, ctx->keys2.p, ctx->scan.p, L.gs_block_start.p, L.ckey.p, n);
    L.gs_nbr.reserve(26 * (size_t)L.nblocks), L.gs_flag.reserve(4 * (size_t)L.nblocks);
    HOT_LAUNCH(ctx, "color_block_neighbours", k_block_neighbours, div_up(26 * (size_t)L.nblocks, 256), 256, 0, h, L.coord.p, L.gs_order.p, L.gs_block_start.p, cb.p, L.gs_nbr.p, L.nblocks);
    HOT_HIP(hipMemsetAsync(L.gs_flag.p, 0, 4 * (size_t)L.nblocks * sizeof(int), ctx->stream)); // sweep numbers start at 1
    HOT_HIP(hipMemcpyAsync(L.color_block_begin, cb.p, 9 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
}

// Regroup the slots of every row as [precede-off | precede-in | diagonal | follow-in | follow-off | structural zeros]
// w.r.t. the GS order ("in" = column inside the row's own 4^3 colour block), stable inside each class.  One wavefront
// per row, the row is staged in LDS and rewritten in place.  rowcnt[4 i ..] = the four off-diagonal class sizes.
template <class T>
__global__ __launch_bounds__(256) void k_gs_split_rows(int32_t* __restrict__ col, T* __restrict__ val, const uint32_t* __restrict__ ckey, int32_t* __restrict__ rowcnt, int n,
    const uint8_t* __restrict__ own, int32_t* __restrict__ gcol /*may be null: the tagged copy of col (in-block column -> -1 - its position in the colour block) the finest-level GS kernels read*/,
    int32_t* __restrict__ rowpn /*may be null: per row, class 0 (preceding off-block columns of the colour just before the row's) | class 6 (following ones of the colour just after) << 16*/)
{
    __shared__ T sval[4][1125];
    __shared__ int32_t scol[4][125];
    __shared__ int32_t sgcol[4][125];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * 4 + w;
    const bool valid = i0 < n && (!own || own[i0]); // sharded: only the rows this rank owns hold a matrix // no early return: the workgroup barrier below must be reached by all four waves
    const int i = valid ? i0 : n - 1;
    const uint32_t keyi = ckey[i];
    int32_t* c = col + (int64_t)i * 125;
    T* v = val + (int64_t)i * 1125;
    // 0 pre-off of the previous colour, 1 other pre-off, 2 pre-in, 3 diagonal, 4 follow-in, 5 follow-off, 6 follow-off of the
    // next colour, 7 structural zero ; 8 = lane has no slot.  The columns a chained sweep (k_gs_sweep) has to wait for — those
    // of the colour just before the row's own in sweep order — sit at the outer ends of the two halves, in the 64 slots the
    // sweep keeps in registers.
    constexpr int NCLS = 8;
    int cls[2], jj[2], tag[2];
    T bv[2][9];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int k = lane + 64 * r;
        cls[r] = NCLS;
        jj[r] = 0, tag[r] = 0;
        if (k < 125 && valid) {
            jj[r] = c[k];
            bool nz = false;
#pragma unroll
            for (int e = 0; e < 9; ++e) bv[r][e] = v[k * 9 + e], nz = nz || bv[r][e] != (T)0;
            if (!nz)
                cls[r] = 7;
            else if (jj[r] == i)
                cls[r] = 3;
            else {
                const uint32_t keyj = ckey[jj[r]];
                const bool in = (keyj >> 7) == (keyi >> 7);
                const int cj = (int)(keyj >> 28), ci = (int)(keyi >> 28);
                cls[r] = keyj < keyi ? (in ? 2 : (cj == ci - 1 ? 0 : 1)) : (in ? 4 : (cj == ci + 1 ? 6 : 5));
                tag[r] = in ? -(int)(keyj & 127u) : jj[r]; // -1 - (0-based position in the block)
            }
            if (cls[r] == 7 || cls[r] == 3) tag[r] = jj[r];
        }
    }
    // stable positions: class-major, then round, then lane
    int cnt[NCLS], base = 0, pos[2] = { 0, 0 };
#pragma unroll
    for (int cc = 0; cc < NCLS; ++cc) {
        unsigned long long m0 = __ballot(cls[0] == cc), m1 = __ballot(cls[1] == cc);
        unsigned long long below = (1ULL << lane) - 1ULL;
        if (cls[0] == cc) pos[0] = base + __popcll(m0 & below);
        if (cls[1] == cc) pos[1] = base + __popcll(m0) + __popcll(m1 & below);
        cnt[cc] = __popcll(m0) + __popcll(m1);
        base += cnt[cc];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (cls[r] < NCLS) {
            scol[w][pos[r]] = jj[r], sgcol[w][pos[r]] = tag[r];
#pragma unroll
            for (int e = 0; e < 9; ++e) sval[w][pos[r] * 9 + e] = bv[r][e];
        }
    }
    __syncthreads();
    if (!valid) return;
    for (int e = lane; e < 1125; e += 64) v[e] = sval[w][e];
    for (int k = lane; k < 125; k += 64) c[k] = scol[w][k];
    if (gcol)
        for (int k = lane; k < 125; k += 64) gcol[(int64_t)i * 125 + k] = sgcol[w][k];
    if (lane == 0) rowcnt[4 * i] = cnt[0] + cnt[1], rowcnt[4 * i + 1] = cnt[2], rowcnt[4 * i + 2] = cnt[4], rowcnt[4 * i + 3] = cnt[5] + cnt[6];
    if (lane == 0 && rowpn) rowpn[i] = cnt[0] | (cnt[6] << 16);
}

// per (colour block, position in block) record {node or -1, the node's four row-class counts}: one 32-byte load gives a GS
// workgroup everything it needs before it can start streaming rows (instead of block_start -> gs_order -> rowcnt)
__global__ void k_gs_pad(const int32_t* __restrict__ block_start, const int32_t* __restrict__ gs_order, const int32_t* __restrict__ rowcnt, int32_t* __restrict__ pad, int nblocks,
    const int32_t* __restrict__ rowpn /*may be null*/)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nblocks * 64) return;
    int b = e >> 6, t = e & 63;
    int p = block_start[b] + t;
    int node = p < block_start[b + 1] ? gs_order[p] : -1;
    int32_t* o = pad + 8 * (int64_t)e;
    o[0] = node;
    for (int k = 0; k < 4; ++k) o[1 + k] = node >= 0 ? rowcnt[4 * node + k] : 0;
    o[5] = o[6] = 0;
    o[7] = (node >= 0 && rowpn) ? rowpn[node] : 0; // the share of the two off-block runs that belongs to the neighbouring colour (k_gs_slot_fill2)
}

// Off-block slots for k_gs_offblock (mg_solve.hip): a row's preceding (forward sweep) / following (backward sweep) off-block columns are cut
// into runs of up to 16 stored entries, numbered over the level in (colour block, position) order — all forward slots first, then all
// backward ones — so that every colour's slots of a direction are one contiguous range and 16-lane groups of a wavefront take one slot
// each whatever the row lengths (3 to 98 entries a row on C2: one wavefront per row leaves more than half of its lanes without an entry).
//   count: flags[pos] = forward slots of the position, flags[npos + pos] = backward slots
//   fill : gs_pad[8 pos + 5 / 6] = first forward / backward slot of the position (record npos: the end sentinels), slot[s] = {first entry, count}
__global__ void k_gs_slot_count(const int32_t* __restrict__ pad, int32_t* __restrict__ flags, int npos)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npos) return;
    const int32_t* o = pad + 8 * (int64_t)e;
    const bool node = o[0] >= 0;
    flags[e] = node ? (o[1] + 15) >> 4 : 0, flags[npos + e] = node ? (o[4] + 15) >> 4 : 0;
}
struct GsColourStarts {
    int pos[2][8]; // [first / end][colour]: positions (64 per colour block) of the blocks whose rows are swept
};
__global__ void k_gs_slot_starts(const int32_t* __restrict__ scan, GsColourStarts cs, int npos, int total, int32_t* __restrict__ out)
{
    const int e = threadIdx.x;
    if (e >= 16) return;
    const int p = cs.pos[e >> 3][e & 7];
    out[e] = scan[p]; // forward ranges (scan[npos]: the first backward slot = the forward total)
    out[16 + e] = p < npos ? scan[npos + p] : total;
}
__global__ void k_gs_slot_fill(int32_t* __restrict__ pad, const int32_t* __restrict__ scan, int2* __restrict__ slot, int npos, int total)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > npos) return;
    int32_t* o = pad + 8 * (int64_t)e;
    if (e == npos) { // sentinel record: where the last colour's ranges end
        o[0] = -1, o[1] = o[2] = o[3] = o[4] = 0, o[5] = scan[npos], o[6] = total, o[7] = 0;
        return;
    }
    const int sf = scan[e], sb = scan[npos + e];
    o[5] = sf, o[6] = sb;
    const int node = o[0];
    if (node < 0) return;
    const int po = o[1], pi = o[2], fi = o[3], fo = o[4];
    const int base = node * 125; // entry index (< 2^31: 125 slots a row, up to 17 M rows)
    for (int q = 0; 16 * q < po; ++q) slot[sf + q] = make_int2(base + 16 * q, min(16, po - 16 * q));
    const int kb = po + pi + 1 + fi;
    for (int q = 0; 16 * q < fo; ++q) slot[sb + q] = make_int2(base + kb + 16 * q, min(16, fo - 16 * q));
}

// The same slots split by the AGE of what they read (k_gs_colour, mg_solve.hip): a row's off-block run of a direction is [columns of the colour
// swept just before the row's own | older colours] (forward; backward: [older | the colour just before in the backward order], k_gs_split_rows'
// classes 0 / 1 and 5 / 6).  The older part is final one colour pass earlier and streams beside the previous colour's substitutions; the part
// that reads the previous colour is summed by the row's own colour block.  Four lists, each numbered in (colour block, position) order:
// 0 forward older, 1 forward previous-colour, 2 backward older, 3 backward previous-colour; srec[pos] = the position's first slot in each
// (record npos: the ends).
__device__ __forceinline__ void gs_slot_runs(const int32_t* o, int (&first)[4], int (&cnt)[4]) // entry offsets inside the row and lengths of the four runs
{
    const int po = o[1], pi = o[2], fi = o[3], fo = o[4], c0 = o[7] & 0xffff, c6 = (o[7] >> 16) & 0xffff;
    const int kb = po + pi + 1 + fi;
    first[0] = c0, cnt[0] = po - c0; // forward older
    first[1] = 0, cnt[1] = c0; // forward, of the previous colour
    first[2] = kb, cnt[2] = fo - c6; // backward older
    first[3] = kb + fo - c6, cnt[3] = c6; // backward, of the colour walked just before
}
__global__ void k_gs_slot_count2(const int32_t* __restrict__ pad, int32_t* __restrict__ flags, int npos)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npos) return;
    const int32_t* o = pad + 8 * (int64_t)e;
    int first[4], cnt[4];
    gs_slot_runs(o, first, cnt);
    for (int k = 0; k < 4; ++k) flags[(size_t)k * npos + e] = o[0] >= 0 ? (cnt[k] + 15) >> 4 : 0;
}
struct GsColourStarts2 {
    int pos[2][8];
};
__global__ void k_gs_slot_starts2(const int32_t* __restrict__ scan, GsColourStarts2 cs, int npos, int total, int32_t* __restrict__ out)
{
    const int e = threadIdx.x; // [list][first / end][colour]
    if (e >= 64) return;
    const int k = e >> 4, p = cs.pos[(e >> 3) & 1][e & 7];
    const size_t at = (size_t)k * npos + p;
    out[e] = at < 4 * (size_t)npos ? scan[at] : total;
}
__global__ void k_gs_slot_fill2(const int32_t* __restrict__ pad, const int32_t* __restrict__ scan, int2* __restrict__ slot, int4* __restrict__ srec, int npos, int total)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > npos) return;
    if (e == npos) { // where the lists end
        srec[e] = make_int4(scan[(size_t)npos], scan[2 * (size_t)npos], scan[3 * (size_t)npos], total);
        return;
    }
    const int32_t* o = pad + 8 * (int64_t)e;
    int s[4];
    for (int k = 0; k < 4; ++k) s[k] = scan[(size_t)k * npos + e];
    srec[e] = make_int4(s[0], s[1], s[2], s[3]);
    const int node = o[0];
    if (node < 0) return;
    int first[4], cnt[4];
    gs_slot_runs(o, first, cnt);
    const int base = node * 125;
    for (int k = 0; k < 4; ++k)
        for (int q = 0; 16 * q < cnt[k]; ++q) slot[s[k] + q] = make_int2(base + first[k] + 16 * q, min(16, cnt[k] - 16 * q));
}

// In-block images for the finest-level GS kernels (layout: GsImg, hot_impl.h; consumer: k_gs_subst, mg_solve.hip).  One workgroup per
// colour block: (1) every row marks itself in the masks of the columns it couples to (LDS), (2) column offsets = running popcounts,
// (3) every entry -(D_r^-1 A_rc) goes to [offset of column c + rank of r among the column's rows]; the same index, per row and step of the walk, goes to the index table.
template <class T>
__global__ __launch_bounds__(512) void k_gs_images(const int32_t* __restrict__ gcol, const T* __restrict__ val, const T* __restrict__ diagBlockInv, const T* __restrict__ diagVal,
    const int32_t* __restrict__ gs_pad, T* __restrict__ img /*of block 0 (shifted for the colour on a row-partitioned level: Level::gs_img_shift)*/, uint16_t* __restrict__ imgi, int block0)
{
    using I = GsImg<T>;
    __shared__ int32_t nodes[64], rcl[64 * 4];
    __shared__ unsigned int mlo[2][64], mhi[2][64];
    __shared__ int32_t coff[2][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = block0 + blockIdx.x;
    T* hdr = img + (size_t)b * I::per_block;
    for (int e = tid; e < 2 * 64 * 9; e += 512) { // header: D and D^-1 by position
        const int pos = (e % 576) / 9;
        const int nd = gs_pad[8 * ((int64_t)b * 64 + pos)];
        hdr[e] = nd < 0 ? (T)0 : (e < 576 ? diagVal : diagBlockInv)[9 * (int64_t)nd + e % 9];
    }
    if (tid < 64) {
        const int32_t* rec = gs_pad + 8 * ((int64_t)b * 64 + tid);
        nodes[tid] = rec[0];
        rcl[4 * tid] = rec[1], rcl[4 * tid + 1] = rec[2], rcl[4 * tid + 2] = rec[3], rcl[4 * tid + 3] = rec[4];
    }
    if (tid < 128) mlo[tid >> 6][tid & 63] = 0u, mhi[tid >> 6][tid & 63] = 0u;
    if (tid < 18) hdr[I::hdr_elems + (tid / 9) * I::per_dir + tid % 9] = (T)0;
    __syncthreads();
    // this thread's entries: rows r = w + 8 t, both directions.  A wavefront has sixteen (direction, row) steps; their column positions are fetched in ONE
    // round trip and kept for the second pass, whose values come four steps at a time (rounds 4 - 5 walked the sixteen steps one dependent load chain
    // after the other, twice: 1.28 ms per C2 build, bound by those chains)
    int cpos[2][8], ek[2][8];
    bool act[2][8];
#pragma unroll
    for (int dir = 0; dir < 2; ++dir)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int r = w + 8 * t, i = nodes[r];
            const int po = rcl[4 * r], pi = rcl[4 * r + 1], fi = rcl[4 * r + 2];
            const int kbeg = dir == 0 ? po : po + pi + 1, kend = dir == 0 ? po + pi : po + pi + 1 + fi; // at most 63 in-block entries
            act[dir][t] = i >= 0 && kbeg + lane < kend;
            ek[dir][t] = act[dir][t] ? i * 125 + kbeg + lane : 0; // entry index (< 2^31: k_gs_slot_fill)
            cpos[dir][t] = -1 - gcol[ek[dir][t]]; // (an idle lane: entry 0 of the matrix, not used)
        }
#pragma unroll
    for (int dir = 0; dir < 2; ++dir)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int r = w + 8 * t;
            if (!act[dir][t]) continue;
            if (r < 32)
                atomicOr(&mlo[dir][cpos[dir][t]], 1u << r);
            else
                atomicOr(&mhi[dir][cpos[dir][t]], 1u << (r - 32));
        }
    __syncthreads();
    {
    if (tid < 2) { // offsets in the order the substitution walks the columns: forward ascending, backward descending
        int off = 1; // entry 0 of either direction is all zeros: what a lane without an entry in a column reads
        for (int s = 0; s < 64; ++s) {
            const int c = tid == 0 ? s : 63 - s;
            coff[tid][c] = off;
            off += __popc(mlo[tid][c]) + __popc(mhi[tid][c]);
        }
    }
    __syncthreads();
    { // the row's entry index at every step of the direction's walk (0: none, the all-zero entry): 64 x 16 bits = the 128 bytes a lane of k_gs_subst loads; a thread writes a quarter of a row's
        const int dir = tid >> 8, r = (tid >> 2) & 63, q4 = tid & 3;
        uint32_t* o = (uint32_t*)(imgi + ((size_t)b * 2 + dir) * I::idx_per_dir + r * 64) + 8 * q4;
        const unsigned long long below = (1ULL << r) - 1ULL;
        for (int s2 = 0; s2 < 8; ++s2) {
            uint32_t pair = 0;
            for (int h = 0; h < 2; ++h) {
                const int st = 16 * q4 + 2 * s2 + h, c = dir == 0 ? st : 63 - st;
                const unsigned long long m = ((unsigned long long)mhi[dir][c] << 32) | mlo[dir][c];
                const uint32_t idx = ((m >> r) & 1ULL) ? (uint32_t)(coff[dir][c] + __popcll(m & below)) : 0u;
                pair |= idx << (16 * h);
            }
            o[s2] = pair;
        }
    }
    }
    __syncthreads();
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        T* dbase = hdr + I::hdr_elems + dir * I::per_dir;
#pragma unroll
        for (int tg = 0; tg < 8; tg += 4) {
            T A[4][9], di[4][9];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = max(nodes[w + 8 * (tg + u)], 0);
#pragma unroll
                for (int e = 0; e < 9; ++e) A[u][e] = val[(int64_t)ek[dir][tg + u] * 9 + e], di[u][e] = diagBlockInv[9 * i + e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = tg + u, r = w + 8 * t;
                if (!act[dir][t]) continue;
                const int cp = cpos[dir][t];
                const unsigned long long m = ((unsigned long long)mhi[dir][cp] << 32) | mlo[dir][cp];
                const int rank = __popcll(m & ((1ULL << r) - 1ULL));
                T* dst = dbase + (size_t)(coff[dir][cp] + rank) * 9;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) dst[rr + 3 * c] = -(di[u][rr] * A[u][3 * c] + di[u][rr + 3] * A[u][3 * c + 1] + di[u][rr + 6] * A[u][3 * c + 2]); // gs_store_tri's product
            }
        }
    }
}

// hot_config.shard_gs = 2: the l1-scaled processor-block smoother (Baker, Falgout, Kolev, Yang, SIAM J. Sci. Comput. 33 (2011), section 6.2).  A rank that
// sweeps its own rows against its own rows only is the symmetric GS of its diagonal block of A, which need not converge; with the absolute row sums of a
// row's OFF-RANK couplings added to its diagonal, D' = D + diag(sum_j |A_ij| 1), it does for every symmetric positive definite A.  One wavefront per owned row.
template <class T>
__global__ __launch_bounds__(256) void k_l1_diag(const int32_t* __restrict__ col, const T* __restrict__ val, const uint8_t* __restrict__ owner, int me, const T* __restrict__ diagVal,
    const T* __restrict__ diagBlockInv, T* __restrict__ gsD, T* __restrict__ gsDinv, T* __restrict__ gsE, int n)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    T e0 = 0, e1 = 0, e2 = 0;
    if (owner[row] == me) {
        for (int k = lane; k < 125; k += 64) {
            const int j = col[(int64_t)row * 125 + k];
            if (owner[j] == me) continue;
            const T* b = val + ((int64_t)row * 125 + k) * 9; // column-major 3 x 3: entry (r, c) at r + 3 c
            e0 += habs(b[0]) + habs(b[3]) + habs(b[6]), e1 += habs(b[1]) + habs(b[4]) + habs(b[7]), e2 += habs(b[2]) + habs(b[5]) + habs(b[8]);
        }
    }
    e0 = wave_sum(e0), e1 = wave_sum(e1), e2 = wave_sum(e2);
    if (lane != 0) return;
    Mat3<T> D;
#pragma unroll
    for (int c = 0; c < 9; ++c) D.a[c] = diagVal[9 * (int64_t)row + c];
    gsE[3 * (int64_t)row] = e0, gsE[3 * (int64_t)row + 1] = e1, gsE[3 * (int64_t)row + 2] = e2;
    if (e0 == (T)0 && e1 == (T)0 && e2 == (T)0) { // an interior row: the smoother's blocks are the matrix's
#pragma unroll
        for (int c = 0; c < 9; ++c) gsD[9 * (int64_t)row + c] = D.a[c], gsDinv[9 * (int64_t)row + c] = diagBlockInv[9 * (int64_t)row + c];
        return;
    }
    D.a[0] += e0, D.a[4] += e1, D.a[8] += e2;
    T amax = (T)0;
#pragma unroll
    for (int c = 0; c < 9; ++c) amax = fmax(amax, habs(D.a[c]));
    const int ex = (amax > (T)0 && amax < (T)INFINITY) ? ilogb(amax) : 0; // (k_diag's scaling: the determinant of a very light node does not underflow in fp32)
    Mat3<T> Ds;
#pragma unroll
    for (int c = 0; c < 9; ++c) Ds.a[c] = scalbn(D.a[c], -ex);
    Mat3<T> Bi = m3_inverse(Ds);
#pragma unroll
    for (int c = 0; c < 9; ++c) gsD[9 * (int64_t)row + c] = D.a[c], gsDinv[9 * (int64_t)row + c] = scalbn(Bi.a[c], -ex);
}

template <class T>
static void split_rows(Ctx<T>* ctx, Level<T>& L)
{
    L.l1 = false;
    if (L.part && ctx->cfg.shard_gs == 2) {
        L.gsD.reserve(9 * (size_t)L.n), L.gsDinv.reserve(9 * (size_t)L.n), L.gsE.reserve(3 * (size_t)L.n);
        HOT_LAUNCH(ctx, "gs_l1_diag", k_l1_diag<T>, div_up(L.n, 4), 256, 0, L.col.p, L.val.p, L.owner.p, ctx->comm.rank, L.diagVal.p, L.diagBlockInv.p, L.gsD.p, L.gsDinv.p, L.gsE.p, L.n);
        L.l1 = true;
    }
    L.rowcnt.reserve(4 * (size_t)L.n);
    if (L.part) HOT_HIP(hipMemsetAsync(L.rowcnt.p, 0, 4 * (size_t)L.n * sizeof(int32_t), ctx->stream)); // rows of other ranks: no matrix, zero counts
    L.gs_col.reserve(125 * (size_t)L.n);
    L.gs_rowpn.reserve((size_t)L.n);
    HOT_LAUNCH(ctx, "gs_split_rows", k_gs_split_rows<T>, div_up(L.n, 4), 256, 0, L.col.p, L.val.p, L.ckey.p, L.rowcnt.p, L.n, L.mask(), L.gs_col.p, L.gs_rowpn.p);
    L.gs_pad.reserve(512 * (size_t)L.nblocks + 8); // + the sentinel record of k_gs_slot_fill
    HOT_LAUNCH(ctx, "gs_pad", k_gs_pad, div_up((size_t)L.nblocks * 64, 256), 256, 0, L.gs_block_start.p, L.gs_order.p, L.rowcnt.p, L.gs_pad.p, L.nblocks, L.gs_rowpn.p);
    L.split = true;
    // levels whose colours hold more blocks than the chip has compute units (smooth_dev: below that the chained single-launch sweep is as fast) run the off-block / substitution kernel
    // pair, which reads the in-block couplings from premultiplied images
    int max_nb = 0;
    for (int c = 0; c < 8; ++c) max_nb = std::max(max_nb, L.color_block_begin[c + 1] - L.color_block_begin[c]);
    // levels below that, swept by ONE chained launch per half sweep (smooth_dev: not row-partitioned, not forced to a launch per colour or to
    // sub-blocks): the inverses of the blocks' in-block triangles, so that a pass is a dense product instead of a 64-step substitution
    L.gs_w_ready = false;
    // (fp64 only: in fp32 the explicit inverse is formed and applied at 6e-8 per operation, and the fp32 configurations' chained levels are the small ones)
    {
        const bool baseline = ctx->cfg.useBaselineMultigrid != 0;
        const int splitLevel = ctx->cfg.topDownMGS ? 1 : ctx->cfg.levelCnt - 1;
        const int kind = L.id < splitLevel ? (baseline ? 5 : ctx->cfg.smoother) : (baseline ? 2 : ctx->cfg.coarseSolver); // what smooth_dev runs on this level
        // (not when the chained sweeps are switched off — a time-out, several ranks —, and never more than 2048 blocks: 290 KB of image per block in fp64,
        // gs_chain = 2 forces the chained launch on levels of any size, which then substitute)
        if (sizeof(T) == 8 && kind == 5 && !L.part && !ctx->gs_no_chain && ctx->cfg.gs_chain != 1 && (ctx->cfg.gs_sub_block == 0 || ctx->cfg.gs_sub_block == 64) && (max_nb <= 256 || ctx->cfg.gs_chain == 2) && L.nblocks <= 2048)
            ctx->build_gs_winv(L);
        if (!L.gs_w_ready && L.gs_w.p) { // the level stopped qualifying (it grew, the chain timed out): the images go
            HOT_HIP(hipStreamSynchronize(ctx->stream));
            HOT_HIP(hipFree(L.gs_w.p));
            L.gs_w.p = nullptr, L.gs_w.cap = 0;
        }
    }
    L.gs_img_ready = false;
    if (max_nb > 256 || ctx->cfg.gs_sub_block == 32) { // (row-partitioned levels too: the rows of other ranks have zero counts, hence no slots and empty images)
        // images only for the blocks this rank owns (a row-partitioned level: one contiguous run of every colour's list; 193 KB per block in fp64)
        {
            const int R1 = ctx->comm.size + 1, me = ctx->comm.rank;
            long long have = 0;
            for (int c = 0; c < 8; ++c) {
                const int b0 = L.color_block_begin[c] + (L.part ? L.csplit[c * R1 + me] : 0), b1 = L.part ? L.color_block_begin[c] + L.csplit[c * R1 + me + 1] : L.color_block_begin[c + 1];
                L.gs_img_shift[c] = have - b0;
                have += b1 - b0;
            }
            L.gs_img.reserve(GsImg<T>::per_block * (size_t)std::max<long long>(have, 1) + 16), // + one entry: k_gs_subst's unconditional loads
            L.gs_imgi.reserve(2 * GsImg<T>::idx_per_dir * (size_t)std::max<long long>(have, 1));
        }
        const int npos = 64 * L.nblocks;
        // one rank: the colour pass is ONE launch (k_gs_colour) that needs the slots split by the age of what they read (A/B build: HOT_GS_PAIR = the
        // kernel pair k_gs_offblock + k_gs_subst, which a row-partitioned level runs — a colour exchange sits between its passes)
        L.gs_fused_ready = false;
        if (!L.part && !ab_flag("HOT_GS_PAIR")) {
            ctx->flags.reserve(4 * (size_t)npos + 64), ctx->scan.reserve(4 * (size_t)npos);
            HOT_LAUNCH(ctx, "gs_slot_count", k_gs_slot_count2, div_up((size_t)npos, 256), 256, 0, L.gs_pad.p, ctx->flags.p, npos);
            L.gs_nslot = ctx->exclusive_scan_i32(ctx->flags.p, ctx->scan.p, 4 * (size_t)npos);
            L.gs_slot.reserve((size_t)L.gs_nslot + 8), L.gs_p1.reserve(3 * ((size_t)L.gs_nslot + 8)), L.gs_srec.reserve((size_t)npos + 1);
            HOT_LAUNCH(ctx, "gs_slot_fill", k_gs_slot_fill2, div_up((size_t)npos + 1, 256), 256, 0, L.gs_pad.p, ctx->scan.p, L.gs_slot.p, L.gs_srec.p, npos, L.gs_nslot);
            GsColourStarts2 cs;
            for (int c = 0; c < 8; ++c) cs.pos[0][c] = 64 * L.color_block_begin[c], cs.pos[1][c] = 64 * L.color_block_begin[c + 1];
            int32_t* d = (int32_t*)(ctx->flags.p); // (flags: consumed by the scan above)
            HOT_LAUNCH(ctx, "gs_slot_starts", k_gs_slot_starts2, 1, 64, 0, ctx->scan.p, cs, npos, L.gs_nslot, d);
            int32_t h[64];
            HOT_HIP(hipMemcpyAsync(h, d, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            ctx->sync();
            for (int k = 0; k < 4; ++k)
                for (int fe = 0; fe < 2; ++fe)
                    for (int c = 0; c < 8; ++c) L.gs_slot_rng2[k][fe][c] = h[16 * k + 8 * fe + c];
            HOT_LAUNCH(ctx, "gs_images", k_gs_images<T>, L.nblocks, 512, 0, L.gs_col.p, L.val.p, L.diagBlockInv.p, L.diagVal.p, L.gs_pad.p, L.gs_img.p, L.gs_imgi.p, 0);
            L.gs_img_ready = L.gs_fused_ready = true;
            return;
        }
        ctx->flags.reserve(2 * (size_t)npos), ctx->scan.reserve(2 * (size_t)npos);
        HOT_LAUNCH(ctx, "gs_slot_count", k_gs_slot_count, div_up((size_t)npos, 256), 256, 0, L.gs_pad.p, ctx->flags.p, npos);
        L.gs_nslot = ctx->exclusive_scan_i32(ctx->flags.p, ctx->scan.p, 2 * (size_t)npos);
        L.gs_slot.reserve((size_t)L.gs_nslot + 8), L.gs_p1.reserve(3 * ((size_t)L.gs_nslot + 8)); // + what the kernels' unconditional (clamped, dropped) loads may touch
        HOT_LAUNCH(ctx, "gs_slot_fill", k_gs_slot_fill, div_up((size_t)npos + 1, 256), 256, 0, L.gs_pad.p, ctx->scan.p, L.gs_slot.p, npos, L.gs_nslot);
        { // where each colour's slots begin, per direction, for the host: k_gs_offblock gets its range as launch arguments instead of starting with a dependent load
            GsColourStarts cs;
            const int R1 = ctx->comm.size + 1, me = ctx->comm.rank;
            for (int c = 0; c < 8; ++c) {
                const int b0 = L.color_block_begin[c], b1 = L.color_block_begin[c + 1];
                // a row-partitioned level: the run of the colour's block list this rank owns (Level::csplit, level_ownership)
                cs.pos[0][c] = 64 * (L.part ? b0 + L.csplit[c * R1 + me] : b0), cs.pos[1][c] = 64 * (L.part ? b0 + L.csplit[c * R1 + me + 1] : b1);
            }
            int32_t* d = (int32_t*)(ctx->flags.p); // (flags: consumed by the scan above)
            HOT_LAUNCH(ctx, "gs_slot_starts", k_gs_slot_starts, 1, 32, 0, ctx->scan.p, cs, npos, L.gs_nslot, d);
            int32_t h[32];
            HOT_HIP(hipMemcpyAsync(h, d, 32 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            ctx->sync();
            for (int dir = 0; dir < 2; ++dir)
                for (int fe = 0; fe < 2; ++fe)
                    for (int c = 0; c < 8; ++c) L.gs_slot_rng[dir][fe][c] = h[16 * dir + 8 * fe + c];
        }
        if (!L.part) // every block, one launch (all shifts are 0)
            HOT_LAUNCH(ctx, "gs_images", k_gs_images<T>, L.nblocks, 512, 0, L.gs_col.p, L.val.p, L.gs_dinv(), L.gs_d(), L.gs_pad.p, L.gs_img.p, L.gs_imgi.p, 0);
        for (int c = 0; c < 8 && L.part; ++c) {
            const int R1 = ctx->comm.size + 1, me = ctx->comm.rank;
            const int b0 = L.color_block_begin[c] + L.csplit[c * R1 + me], b1 = L.color_block_begin[c] + L.csplit[c * R1 + me + 1];
            if (b1 > b0)
                HOT_LAUNCH(ctx, "gs_images", k_gs_images<T>, b1 - b0, 512, 0, L.gs_col.p, L.val.p, L.gs_dinv(), L.gs_d(), L.gs_pad.p, L.gs_img.p + L.gs_img_shift[c] * (long long)GsImg<T>::per_block,
                    L.gs_imgi.p + L.gs_img_shift[c] * 2 * (long long)GsImg<T>::idx_per_dir, b0);
        }
        L.gs_img_ready = true;
    }
}

template <class T>
static void alloc_work(Level<T>& L)
{
    size_t m = 3 * (size_t)L.n;
    L.residual.reserve(m), L.initialResidual.reserve(m), L.sol.reserve(m), L.du.reserve(m), L.dAu.reserve(m), L.tmp.reserve(m);
    L.built = true;
}

template <class T>
void Ctx<T>::color_level(Level<T>& L)
{
    if (L.colored) return;
    mark_colors(this, L);
    L.colored = true;
}
// coarse rows for which this rank holds a partial Galerkin sum: at least one of the <= 27 fine children is a row it owns
__global__ void k_coarse_touched(const int32_t* __restrict__ child, const uint8_t* __restrict__ fine_own, uint8_t* touched, int nc)
{
    int I = blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= nc) return;
    bool t = false;
    for (int q = 0; q < 27; ++q) {
        const int ci = child[I * 27 + q];
        t = t || (ci >= 0 && fine_own[ci]);
    }
    touched[I] = t ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ baseline geometric multigrid
// MultigridSimulation::particlesToMultigrids (Projects/multigrid/MultigridSimulation.inl:345-456): coarse level l is a
// real MPM grid of spacing 2^l dx — particles re-sorted into it (:40-124), mass rasterised (:404-428), boundaries queried at
// its own nodes (buildMultigridBoundaries :126-161), and the system matrix re-assembled from the same per-particle
// dP/dF (buildMultigridMatrices :163-342; the reference caches dP/dF per particle, here it is recomputed from the same
// trial F, which gives the same numbers).  Each such grid is a whole context of this library; its level-0 matrix
// becomes level l of the hierarchy.
__global__ void k_invert_perm(const int32_t* __restrict__ slot2orig, int32_t* __restrict__ orig2slot, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) orig2slot[slot2orig[p]] = (int32_t)p;
}
// dst[c][i] = src[c][ orig2slot[ slot2orig_dst[i] ] ]: a per-particle SoA array of the fine context, in the coarse context's sorted order
template <class T>
__global__ void k_gather_via_orig(const T* __restrict__ src, T* __restrict__ dst, const int32_t* __restrict__ slot2orig_dst, const int32_t* __restrict__ orig2slot_src, int64_t n, int comps)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int64_t q = orig2slot_src[slot2orig_dst[p]];
    for (int c = 0; c < comps; ++c) dst[(int64_t)c * n + p] = src[(int64_t)c * n + q];
}
__global__ void k_count_missing_parents(const int32_t* __restrict__ pcol, int64_t n, int32_t* count)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && pcol[e] < 0) atomicAdd(count, 1);
}

template <class T>
Ctx<T>* Ctx<T>::build_gmg_grid(int level)
{
    while ((int)gmg.size() < level) gmg.push_back(nullptr);
    Ctx<T>*& g = gmg[level - 1];
    if (!g) {
        hot_config c = cfg;
        c.dx = cfg.dx * (double)(1 << level);
        c.levelCnt = 1, c.useBaselineMultigrid = 0, c.profile = 0, c.debug_store = 0;
        g = new Ctx<T>(c);
    }
    g->cfg.project = cfg.project, g->cfg.systemBCProject = cfg.systemBCProject, g->cfg.Ainv = cfg.Ainv, g->cfg.boundaryType = cfg.boundaryType;
    const int64_t n = Np;
    // particles in this context's sorted order, carrying their original indices (the sort key's tie break, :69-71)
    g->Np = n;
    g->reserve_particles(n);
    HOT_HIP(hipMemsetAsync(g->pGid.p, 0, (size_t)n * sizeof(int32_t), stream)); // ids are carried along by the sort; unused here
    auto give = [&](DBuf<T>& dst, const DBuf<T>& src, int comps) { HOT_HIP(hipMemcpyAsync(dst.p, src.p, (size_t)n * comps * sizeof(T), hipMemcpyDeviceToDevice, stream)); };
    give(g->pX, pX, 3), give(g->pV, pV, 3), give(g->pM, pM, 1), give(g->pC, pC, 9), give(g->pF, pFn, 9), give(g->pVol, pVol, 1), give(g->pMu, pMu, 1), give(g->pLam, pLam, 1), give(g->pJp, pJp, 1);
    HOT_HIP(hipMemcpyAsync(g->slot2orig.p, slot2orig.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    orig2slot.reserve(n);
    HOT_LAUNCH(this, "gmg_invert_perm", k_invert_perm, div_up(n, 256), 256, 0, slot2orig.p, orig2slot.p, n);
    sync();
    g->Ng = g->Nb = g->Nn = 0;
    // boundaries: the collision objects are queried again at the coarse grid's nodes
    if (!cobjs.empty())
        g->set_collision_objects((int32_t)cobjs.size(), cobjs.data());
    else if (!hs_origin.empty())
        g->set_halfspaces((int32_t)hs_origin.size() / 3, hs_origin.data(), hs_normal.data());
    else {
        need(Nc == 0, "useBaselineMultigrid: the boundaries must be given as hot_set_collision_objects / hot_set_sticky_halfspaces (every coarse grid queries them at its own nodes; an explicit node list only describes level 0)");
        g->set_collision_objects(0, nullptr); // the cached coarse context must not keep boundaries the caller has cleared since
        g->set_halfspaces(0, nullptr, nullptr);
    }
    g->sort();
    g->p2g();
    g->begin_step((double)dt);
    // trial F of every particle (the linearisation point of dP/dF), in the coarse context's order
    HOT_LAUNCH(this, "gmg_gather_F", k_gather_via_orig<T>, div_up(n, 256), 256, 0, pFt.p, g->pFt.p, g->slot2orig.p, orig2slot.p, n, 9);
    sync();
    g->build_hessian();
    return g;
}

__global__ void k_mark_tiles(const int32_t* __restrict__ tileDof, int64_t n, const uint8_t* __restrict__ own, uint8_t* __restrict__ need)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int32_t j = tileDof[e];
    if (j >= 0 && !own[j]) need[j] = 1;
}
__global__ void k_mark_table_rows(const int32_t* __restrict__ ids, int width, int nrows, const uint8_t* __restrict__ row_own, const uint8_t* __restrict__ own, uint8_t* __restrict__ need)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)nrows * width) return;
    if (!row_own[e / width]) return;
    const int32_t j = ids[e];
    if (j >= 0 && !own[j]) need[j] = 1;
}
// Halo mode: (re)builds the exchange lists of level `level` from everything that reads its vectors on this rank as the hierarchy stands:
// the 125-stencil of the owned rows (always), the nodes of the particle tiles (level 0), the 27 children of the coarse rows this rank
// owns on the next level (restriction, when that level is partitioned) and the 4^3 coarse windows of the rows it owns on the finer level
// (prolongation and r -= (A P) e).
template <class T>
static void rebuild_halo(Ctx<T>* ctx, int level)
{
    Level<T>& L = *ctx->levels[level];
    if (!L.part) return;
    ctx->build_halo(L, [&](uint8_t* need) {
        if (level == 0) {
            constexpr int TILE = (Geo<T>::BX + 2) * (Geo<T>::BY + 2) * (Geo<T>::BZ + 2);
            const int64_t nt = (int64_t)ctx->Ng * TILE;
            HOT_LAUNCH(ctx, "halo_mark", k_mark_tiles, div_up((size_t)nt, 256), 256, 0, ctx->tileDof.p, nt, L.own.p, need);
        }
        if (level + 1 < (int)ctx->levels.size() && ctx->levels[level + 1]->part) {
            Level<T>& C = *ctx->levels[level + 1];
            HOT_LAUNCH(ctx, "halo_mark", k_mark_table_rows, div_up((size_t)C.n * 27, 256), 256, 0, C.child.p, 27, C.n, C.own.p, L.own.p, need);
        }
        if (level > 0) {
            Level<T>& F = *ctx->levels[level - 1];
            HOT_LAUNCH(ctx, "halo_mark", k_mark_table_rows, div_up((size_t)F.n * 64, 256), 256, 0, F.apc.p, 64, F.n, F.own.p, L.own.p, need);
        }
    });
}

// Halo mode, end of hot_p2g: level 0 of the hierarchy is created right away — coordinates, coordinate map, colouring, row ownership and
// exchange lists depend on the numbering only — because every vector operation of the step (begin_step, the exit test, the state pass)
// already works on partitioned vectors.  hot_build_hessian fills in the matrix later.
template <class T>
void Ctx<T>::level0_ownership()
{
    release_levels();
    Level<T>* L = acquire_level(0);
    levels.push_back(L);
    L->n = Nn;
    L->coord.reserve(3 * (size_t)Nn);
    HOT_HIP(hipMemcpyAsync(L->coord.p, id2coord.p, 3 * (size_t)Nn * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    L->nstart = nstart0;
    {
        IndexPhase ip(this);
        build_coord_map(this, *L);
        color_level(*L);
        level_ownership(*L);
    }
    rebuild_halo(this, 0);
    vmask = L->own.p;
    gs_no_chain = true; // a chained sweep's timeout would be rank-local and desynchronise the collectives
}

template <class T>
void Ctx<T>::build_mg()
{
    need(!levels.empty(), "hot_build_mg before hot_build_hessian");
    need(!(!cfg.systemBCProject && cfg.levelCnt > 1), "levelCnt > 1 requires systemBCProject (ImplicitSolver.h:339)");
    need(cfg.levelCnt >= 1 && cfg.levelCnt <= 10, "levelCnt must be in [1,10] (MultigridPreconditioner.h:369)");
    for (int k : { cfg.smoother, cfg.coarseSolver })
        need(k == 0 || k == 1 || k == 2 || k == 5 || k == 6 || k == 7, "smoother/coarseSolver must be 0, 1, 2, 5, 6 or (coarseSolver only) 7; 3/4 are not selectable in the reference either");
    need(!(cfg.smoother == 7 && cfg.levelCnt > 1), "Do not use IC solver as smoother in the multigrid! (MultigridPreconditioner.h:614,686)");
    need(!(cfg.coarseSolver == 7 && cfg.useBaselineMultigrid), "coarseSolver 7 with useBaselineMultigrid: the baseline fixes its own top solver (MultigridSimulation.inl:447-454)");
    const bool baseline = cfg.useBaselineMultigrid != 0;
    if (baseline) {
        need(!sharded(), "useBaselineMultigrid is a single-rank mode: its coarse levels are whole MPM grids re-rasterised from ALL particles (hot_set_comm with size > 1 is not supported with it)");
        need(cfg.Ainv == 1, "useBaselineMultigrid scales with the inverse diagonal blocks (MultigridSimulation.inl:446): set Ainv = 1");
        need(!cfg.topDownMGS, "useBaselineMultigrid fixes the V-cycle schedule (MultigridSimulation.inl:447-454); topDownMGS does not apply");
    }
    double t0 = wall_ms();
    release_levels(1);
    bool colors = baseline || cfg.smoother == 5 || cfg.coarseSolver == 5 || cfg.coarseSolver == 7 || sharded(); // baseline: GS smoother, PCG on top (:447-448); sharded: row ownership follows the colour blocks
    Level<T>& L0 = *levels[0];
    alloc_work(L0);
    if (colors) color_level(L0);
    if (!baseline && ((cfg.coarseSolver == 6 && cfg.levelCnt == 1) || (cfg.smoother == 6 && cfg.levelCnt > 1))) estimate_2norm(L0, 1e-6); // MultigridPreconditioner.h:610-611
    for (int level = 0; level < cfg.levelCnt - 1; ++level) {
        Level<T>& F = *levels[level];
        int n = F.n;
        build_coord_map(this, F);
        Level<T>* Cp = acquire_level(level + 1);
        levels.push_back(Cp);
        Level<T>& C = *Cp;
        Ctx<T>* grid = baseline ? build_gmg_grid(level + 1) : nullptr;
        size_t nc;
        if (baseline) {
            // ---- coarse node set and matrix = the coarse grid's own DOFs and re-rasterised system
            Level<T>& G0 = *grid->levels[0];
            C.n = G0.n;
            nc = C.n;
            C.coord.reserve(3 * nc), C.col.reserve(125 * nc), C.val.reserve(1125 * nc), C.child.reserve(27 * nc);
            HOT_HIP(hipMemcpyAsync(C.coord.p, G0.coord.p, 3 * nc * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
            HOT_HIP(hipMemcpyAsync(C.col.p, G0.col.p, 125 * nc * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
            HOT_HIP(hipMemcpyAsync(C.val.p, G0.val.p, 1125 * nc * sizeof(T), hipMemcpyDeviceToDevice, stream));
            build_coord_map(this, C);
        }
        else {
            // ---- coarse node set with first-touch numbering
            uint32_t cap = 1024;
            while (cap < 2u * (uint32_t)n + 16u) cap <<= 1;
            C.hkeys.reserve(cap), C.hrank.reserve(cap), C.hid.reserve(cap);
            C.map.keys = C.hkeys.p, C.map.minrank = C.hrank.p, C.map.id = C.hid.p, C.map.mask = cap - 1;
            HOT_LAUNCH(this, "mg_hash_clear", k_hash_clear2, div_up(cap, 256), 256, 0, C.map);
            size_t cand = 8 * (size_t)n;
            flags.reserve(cand), scan.reserve(cand);
            HOT_LAUNCH(this, "mg_coarse_insert", k_coarse_insert, div_up(cand, 256), 256, 0, C.map, F.coord.p, n);
            HOT_LAUNCH(this, "mg_coarse_flag", k_coarse_flag, div_up(cand, 256), 256, 0, C.map, F.coord.p, flags.p, n);
            C.n = exclusive_scan_i32(flags.p, scan.p, cand);
            nc = C.n;
            C.coord.reserve(3 * nc), C.col.reserve(125 * nc), C.val.reserve(1125 * nc), C.child.reserve(27 * nc);
            HOT_LAUNCH(this, "mg_coarse_assign", k_coarse_assign, div_up(cand, 256), 256, 0, C.map, F.coord.p, flags.p, scan.p, C.coord.p, n);
            if (sharded()) { // a rank's coarse id prefix: the coarse nodes first touched by fine ids below its fine prefix (first-touch numbering keeps prefixes)
                C.nstart.assign(comm.size + 1, C.n);
                for (int r = 0; r < comm.size; ++r)
                    if (F.nstart[r] < n) HOT_HIP(hipMemcpyAsync(&C.nstart[r], scan.p + 8 * (size_t)F.nstart[r], sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                sync();
            }
        }
        // ---- transfer tables
        F.pcol.reserve(8 * (size_t)n), F.pw.reserve(8 * (size_t)n);
        HOT_LAUNCH(this, "mg_build_P", k_build_P<T>, div_up(n, 256), 256, 0, C.map, F.coord.p, F.pcol.p, F.pw.p, n);
        HOT_LAUNCH(this, "mg_build_children", k_build_children, div_up(27 * nc, 256), 256, 0, F.map, C.coord.p, C.child.p, C.n);
        if (baseline) { // ZIRAN_ASSERT(new_coord2id->find(...)) of buildMultigridMatrices (:207)
            HOT_HIP(hipMemsetAsync(dscal.p + 210, 0, sizeof(double), stream));
            HOT_LAUNCH(this, "gmg_check_parents", k_count_missing_parents, div_up(8 * (size_t)n, 256), 256, 0, F.pcol.p, 8 * (int64_t)n, (int32_t*)(dscal.p + 210));
            int32_t missing = 0;
            HOT_HIP(hipMemcpyAsync(&missing, dscal.p + 210, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            sync();
            need(missing == 0, "useBaselineMultigrid: a fine node has a trilinear parent that is not a DOF of the coarse grid");
        }
        else
            HOT_LAUNCH(this, "mg_coarse_cols", k_coarse_cols, div_up(125 * nc, 256), 256, 0, C.map, C.coord.p, C.col.p, C.n);
        // ---- A_c = R (A P)
        F.apv.reserve(576 * (size_t)n), F.apc.reserve(64 * (size_t)n);
        HOT_LAUNCH(this, "mg_AP", k_ap<T>, div_up(n, 4), 256, 0, F.coord.p, F.val.p, F.apv.p, n, F.mask());
        HOT_LAUNCH(this, "mg_AP_cols", k_ap_cols, div_up(64 * (size_t)n, 256), 256, 0, C.map, F.coord.p, F.apc.p, n);
        if (!baseline) HOT_LAUNCH(this, "mg_RAP", k_rap<T>, nc, 256, 0, C.coord.p, C.child.p, F.apv.p, C.val.p, C.n, F.mask());
        if (F.part) {
            // every rank has summed its own fine rows into ALL coarse rows (zeros where it owns no child).  Large coarse levels
            // stay partitioned: partial rows go to their owners; small ones are replicated: one all-reduce of the whole matrix
            const int minrows = comm.partition_min_rows > 0 ? comm.partition_min_rows : 4096;
            if (C.n >= minrows) {
                color_level(C);
                level_ownership(C);
                DBuf<uint8_t> touched;
                touched.reserve(nc);
                HOT_LAUNCH(this, "shard_coarse_touched", k_coarse_touched, div_up(nc, 256), 256, 0, C.child.p, F.own.p, touched.p, C.n);
                exchange_rows(C, touched.p);
                if (halo_mode()) rebuild_halo(this, level + 1), rebuild_halo(this, level); // C: stencil + the windows of this rank's fine rows; F: + the children of its coarse rows
            }
            else {
                CommTag tag(this, "coarse_matrix_allreduce");
                c_allreduce(C.val.p, (int64_t)nc * 1125, REAL, HOT_COMM_SUM, true);
            }
        }
        build_diagonal(C);
        C.nnzb = -1; // counted on request (hot_get_level_nnzb)
        alloc_work(C);
        if (colors) color_level(C);
        if (!baseline && ((cfg.coarseSolver == 6 && level + 2 == cfg.levelCnt) || (cfg.smoother == 6 && level + 2 < cfg.levelCnt))) estimate_2norm(C, 1e-6); // :682-683
        if (colors) split_rows(this, F); // level `level` is no longer needed in stencil-slot order
    }
    if (cfg.coarseSolver == 7) { // incomplete-Cholesky top solver: factor while the top level is still in stencil-slot order, then regroup the factor like the matrix
        Level<T>& Top = *levels.back();
        build_ic(Top);
        Top.ic_rowcnt.reserve(4 * (size_t)Top.n), Top.ic_pad.reserve(512 * (size_t)Top.nblocks);
        HOT_LAUNCH(this, "gs_split_rows", k_gs_split_rows<T>, div_up(Top.n, 4), 256, 0, Top.ic_col.p, Top.ic_val.p, Top.ckey.p, Top.ic_rowcnt.p, Top.n, (const uint8_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
        HOT_LAUNCH(this, "gs_pad", k_gs_pad, div_up((size_t)Top.nblocks * 64, 256), 256, 0, Top.gs_block_start.p, Top.gs_order.p, Top.ic_rowcnt.p, Top.ic_pad.p, Top.nblocks, (const int32_t*)nullptr);
    }
    if (colors) split_rows(this, *levels.back());
    sync();
    stats.ms_mg_build += wall_ms() - t0;
}

template <class T>
void Ctx<T>::get_level(int32_t level, int32_t* nrows, int32_t* colsize, int32_t* ic)
{
    need(level >= 0 && level < (int)levels.size(), "level out of range");
    Level<T>& L = *levels[level];
    if (nrows) *nrows = L.n;
    if (colsize) *colsize = 125;
    download(ic, L.coord.p, 3 * (size_t)L.n);
    sync();
}
template <class T>
void Ctx<T>::get_matrix(int32_t level, int32_t* entryCol, void* entryVal)
{
    need(level >= 0 && level < (int)levels.size(), "level out of range");
    Level<T>& L = *levels[level];
    download(entryCol, L.col.p, 125 * (size_t)L.n);
    download(entryVal, L.val.p, 1125 * (size_t)L.n);
    sync();
}
template <class T>
void Ctx<T>::get_prolongation(int32_t level, int32_t* entryCol, void* weight)
{
    need(level >= 0 && level + 1 < (int)levels.size(), "level out of range");
    Level<T>& L = *levels[level];
    download(entryCol, L.pcol.p, 8 * (size_t)L.n);
    download(weight, L.pw.p, 8 * (size_t)L.n);
    sync();
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
