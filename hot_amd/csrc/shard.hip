// libhotmi355x — one connected body over several ranks (one context per rank = per GPU): the data-path side of the
// decomposition described in include/hot_mi355x.h (hot_comm) and DESIGN.md §7.  The reference is a single process; the
// hazards this has to respect are the reference's own: the 8-colouring of the scatters (Lib/MPM/MpmSimulationBase.h:251-264)
// becomes "partial tiles, summed over the ranks", the colour order of gs_smooth (Projects/multigrid/MultigridPreconditioner.h:
// 266-318) becomes "owner computes a colour, hands it to every rank, next colour", the serial Set_Page / getNumNodes /
// first-touch numberings (MpmSimulationBase.cpp:1099-1125, MpmGrid.h:148-161, MultigridPreconditioner.h:619-664) become
// "rank lists concatenated in rank order", which is the serial order because a rank's shard is a contiguous range of the
// globally sorted particle groups.
//
// The library itself never communicates: the three collectives of hot_comm are callbacks (RCCL through torch.distributed in
// hot_amd/dist.py).  They are called with device pointers after the context's stream has been synchronised.
#include "hot_impl.h"
#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace hot {

template <class T>
void Ctx<T>::set_comm(const hot_comm* c)
{
    if (c && c->size > 1) {
        need(c->allreduce && c->allgather && c->alltoallv, "hot_set_comm: all three collectives are required");
        need(c->rank >= 0 && c->rank < c->size && c->size <= 64, "hot_set_comm: 0 <= rank < size <= 64");
        need(!cfg.useBaselineMultigrid, "hot_set_comm: the --baseline geometric multigrid is single-rank only");
        comm = *c;
        gs_no_chain = true; // a chained coarse-level sweep that timed out would make ONE rank redo its solve and desynchronise the collectives: launch-per-pass sweeps only
    }
    else {
        comm = hot_comm{};
        gs_no_chain = gs_chain_timed_out; // back to one rank: chained sweeps again, unless one has timed out on this context
    }
    block_first.clear();
    vmask = nullptr;
}
template <class T>
void Ctx<T>::c_allreduce(void* buf, int64_t n, int dtype, int op, bool on_device)
{
    if (!sharded() || n <= 0) return;
    if (on_device && !comm.stream_ordered) HOT_HIP(hipStreamSynchronize(stream));
    prof.count(on_device ? "comm_allreduce" : "comm_allreduce_scalars");
    account(n * (dtype == HOT_COMM_F32 || dtype == HOT_COMM_I32 ? 4 : 8));
    HOT_CHECK(comm.allreduce(comm.user, buf, n, dtype, op, on_device ? 1 : 0) == 0, HOT_ERR_DEVICE, "hot_comm.allreduce failed");
}
template <class T>
void Ctx<T>::c_allgather(const void* send, void* recv, int64_t bytes, bool on_device)
{
    if (on_device && !comm.stream_ordered) HOT_HIP(hipStreamSynchronize(stream));
    prof.count("comm_allgather");
    account(bytes);
    HOT_CHECK(comm.allgather(comm.user, send, recv, bytes, on_device ? 1 : 0) == 0, HOT_ERR_DEVICE, "hot_comm.allgather failed");
}
template <class T>
void Ctx<T>::c_alltoallv(const void* send, const int64_t* soff, const int64_t* sbytes, void* recv, const int64_t* roff, const int64_t* rbytes)
{
    if (!comm.stream_ordered) HOT_HIP(hipStreamSynchronize(stream));
    prof.count("comm_alltoallv");
    {
        int64_t b = 0;
        for (int q = 0; q < comm.size; ++q) b += sbytes[q];
        account(b);
    }
    HOT_CHECK(comm.alltoallv(comm.user, send, soff, sbytes, recv, roff, rbytes, 1) == 0, HOT_ERR_DEVICE, "hot_comm.alltoallv failed");
}
template <class T>
void Ctx<T>::allreduce_tiles(T* tiles, int q)
{
    c_allreduce(tiles, (int64_t)q * Nb * EPB, REAL, HOT_COMM_SUM, true);
}

// ------------------------------------------------------------------------------------------------ global block list
struct RankCounts {
    int v[64];
};
__global__ void k_hash_clear4(HashMap h)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > h.mask) return;
    h.keys[i] = ~0ULL;
    h.minrank[i] = ~0ULL;
    h.id[i] = -1;
}
// candidate s = rank * maxn + position in that rank's first-touch list
__global__ void k_merge_insert(HashMap h, const uint64_t* __restrict__ lists, RankCounts cnt, int maxn, int total)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total || s % maxn >= cnt.v[s / maxn]) return;
    hash_insert_min(h, lists[s] >> 12, (unsigned long long)s);
}
__global__ void k_merge_flag(HashMap h, const uint64_t* __restrict__ lists, RankCounts cnt, int maxn, int total, int32_t* flags)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total) return;
    int f = 0;
    if (s % maxn < cnt.v[s / maxn]) f = h.minrank[hash_find_slot(h, lists[s] >> 12)] == (unsigned long long)s;
    flags[s] = f;
}
__global__ void k_merge_assign(HashMap h, const uint64_t* __restrict__ lists, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, uint64_t* blocks, int total)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total || !flags[s]) return;
    h.id[hash_find_slot(h, lists[s] >> 12)] = scan[s];
    blocks[scan[s]] = lists[s];
}

// The global block list is the concatenation, in rank order, of the ranks' own first-touch lists with repeated pages
// dropped: exactly the serial Set_Page order of the whole body (MpmSimulationBase.cpp:1099-1125), because rank r's particle
// groups all precede rank r+1's in the global sort.  The same min-sequence-number hashing as the single-rank list.
template <class T>
void Ctx<T>::merge_block_lists()
{
    const int R = comm.size;
    std::vector<int64_t> counts(R, 0);
    int64_t mine = Nb;
    c_allgather(&mine, counts.data(), sizeof(int64_t), false);
    int64_t maxn = 0, sum = 0;
    RankCounts rc{};
    for (int r = 0; r < R; ++r) maxn = std::max(maxn, counts[r]), sum += counts[r], rc.v[r] = (int)counts[r];
    const int total = (int)(maxn * R);
    xsend.reserve((size_t)maxn * 8), xrecv.reserve((size_t)total * 8);
    HOT_HIP(hipMemsetAsync(xsend.p, 0, (size_t)maxn * 8, stream));
    HOT_HIP(hipMemcpyAsync(xsend.p, blocks.p, (size_t)Nb * 8, hipMemcpyDeviceToDevice, stream));
    c_allgather(xsend.p, xrecv.p, maxn * 8, true);
    const uint64_t* lists = (const uint64_t*)xrecv.p;
    uint32_t capn = 1024;
    while (capn < 2 * (uint32_t)sum + 16u) capn <<= 1;
    bh_keys.reserve(capn), bh_rank.reserve(capn), bh_id.reserve(capn);
    block_map.keys = bh_keys.p, block_map.minrank = bh_rank.p, block_map.id = bh_id.p, block_map.mask = capn - 1;
    flags.reserve(total), scan.reserve(total);
    HOT_LAUNCH(this, "hash_clear", k_hash_clear4, div_up(capn, 256), 256, 0, block_map);
    HOT_LAUNCH(this, "block_merge_insert", k_merge_insert, div_up(total, 256), 256, 0, block_map, lists, rc, (int)maxn, total);
    HOT_LAUNCH(this, "block_merge_flag", k_merge_flag, div_up(total, 256), 256, 0, block_map, lists, rc, (int)maxn, total, flags.p);
    Nb = exclusive_scan_i32(flags.p, scan.p, total);
    blocks.reserve(Nb);
    HOT_LAUNCH(this, "block_merge_assign", k_merge_assign, div_up(total, 256), 256, 0, block_map, lists, flags.p, scan.p, blocks.p, total);
    block_first.assign(R + 1, Nb);
    for (int r = 0; r < R; ++r) HOT_HIP(hipMemcpyAsync(&block_first[r], scan.p + (size_t)r * maxn, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    sync();
}

// ------------------------------------------------------------------------------------------------ row ownership
struct Int9 {
    int v[9];
};
__global__ void k_node_owner(const uint32_t* __restrict__ ckey, const uint8_t* __restrict__ owner_b, Int9 cb, uint8_t* owner, uint8_t* own, int me, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = ckey[i];
    const uint8_t o = owner_b[cb.v[k >> 28] + (int)((k >> 7) & 0x1fffffu)];
    owner[i] = o, own[i] = o == (uint8_t)me;
}

// The owner of a 4^3 colour block comes from the colouring (mark_colors, mg_build.hip: hot_config.shard_owner — balanced between the two sides of
// a cut by default), which also orders the blocks of a colour by (owner, first touch): every rank owns a contiguous run of each colour's block
// list, and its nodes of colour c are one contiguous position range of gs_order.
template <class T>
void Ctx<T>::level_ownership(Level<T>& L)
{
    const int R = comm.size, me = comm.rank, nb = L.nblocks;
    HOT_CHECK((int)L.nstart.size() == R + 1 && L.colored && nb > 0, HOT_ERR_INVALID, "level_ownership: prefixes / colouring missing");
    std::vector<int32_t> hstart(nb + 1);
    HOT_HIP(hipMemcpyAsync(hstart.data(), L.gs_block_start.p, (size_t)(nb + 1) * 4, hipMemcpyDeviceToHost, stream));
    sync();
    std::vector<uint8_t> ob(nb);
    L.csplit.assign(8 * (R + 1), 0);
    L.xbeg.assign(R * 8, 0), L.xcnt.assign(R * 8, 0);
    for (int c = 0; c < 8; ++c) {
        const int b0 = L.color_block_begin[c], b1 = L.color_block_begin[c + 1];
        HOT_CHECK((int)L.block_owner_h.size() == nb, HOT_ERR_INVALID, "level_ownership: the colouring carries no block owners");
        for (int b = b0; b < b1; ++b) {
            ob[b] = L.block_owner_h[b];
            HOT_CHECK(ob[b] < R && (b == b0 || ob[b] >= ob[b - 1]), HOT_ERR_INVALID, "level_ownership: colour blocks are not ordered by owner");
        }
        // split[c][r] = first block (relative to b0) owned by a rank >= r
        int b = b0;
        for (int q = 0; q <= R; ++q) {
            while (b < b1 && (int)ob[b] < q) ++b;
            L.csplit[c * (R + 1) + q] = (q == R ? b1 : b) - b0;
        }
        for (int q = 0; q < R; ++q) {
            const int lo = b0 + L.csplit[c * (R + 1) + q], hi = b0 + L.csplit[c * (R + 1) + q + 1];
            L.xbeg[q * 8 + c] = hstart[lo], L.xcnt[q * 8 + c] = hstart[hi] - hstart[lo];
        }
    }
    L.xmax_full = 0;
    for (int c = 0; c < 8; ++c) L.xmax_col[c] = 0;
    for (int q = 0; q < R; ++q) {
        int tot = 0;
        for (int c = 0; c < 8; ++c) tot += L.xcnt[q * 8 + c], L.xmax_col[c] = std::max(L.xmax_col[c], L.xcnt[q * 8 + c]);
        L.xmax_full = std::max(L.xmax_full, tot);
    }
    DBuf<uint8_t> dob;
    dob.reserve(nb);
    L.owner.reserve(L.n), L.own.reserve(L.n), L.dxtab.reserve(2 * R * 8);
    std::vector<int32_t> tab(2 * R * 8);
    for (int k = 0; k < R * 8; ++k) tab[k] = L.xbeg[k], tab[R * 8 + k] = L.xcnt[k];
    HOT_HIP(hipMemcpyAsync(dob.p, ob.data(), nb, hipMemcpyHostToDevice, stream));
    HOT_HIP(hipMemcpyAsync(L.dxtab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, stream));
    Int9 cb;
    for (int c = 0; c < 9; ++c) cb.v[c] = L.color_block_begin[c];
    HOT_LAUNCH(this, "shard_node_owner", k_node_owner, div_up(L.n, 256), 256, 0, L.ckey.p, dob.p, cb, L.owner.p, L.own.p, me, L.n);
    sync(); // ob / tab / dob go out of scope
    L.part = true;
}

// ------------------------------------------------------------------------------------------------ vector exchange
// position in gs_order of the k-th exchanged node of rank r (all colours: the rank's eight colour ranges back to back), or -1
__device__ __forceinline__ int xchg_pos(const int32_t* __restrict__ tab, int R, int r, int colour, int k)
{
    const int32_t* beg = tab + r * 8;
    const int32_t* cnt = tab + R * 8 + r * 8;
    if (colour >= 0) return k < cnt[colour] ? beg[colour] + k : -1;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (k < cnt[c]) return beg[c] + k;
        k -= cnt[c];
    }
    return -1;
}
template <class T>
__global__ void k_xchg_pack(const T* __restrict__ x, const int32_t* __restrict__ gs_order, const int32_t* __restrict__ tab, int R, int me, int colour, T* __restrict__ out, int maxc, int ncomp)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= maxc) return;
    const int p = xchg_pos(tab, R, me, colour, k);
    const int64_t i = p >= 0 ? gs_order[p] : 0;
    for (int d = 0; d < ncomp; ++d) out[ncomp * (int64_t)k + d] = p >= 0 ? x[ncomp * i + d] : (T)0;
}
template <class T>
__global__ void k_xchg_unpack(T* __restrict__ x, const int32_t* __restrict__ gs_order, const int32_t* __restrict__ tab, int R, int me, int colour, const T* __restrict__ in, int maxc, int ncomp)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * maxc) return;
    const int r = e / maxc, k = e - r * maxc;
    if (r == me) return;
    const int p = xchg_pos(tab, R, r, colour, k);
    if (p < 0) return;
    const int64_t i = gs_order[p];
    for (int d = 0; d < ncomp; ++d) x[ncomp * i + d] = in[ncomp * (int64_t)e + d];
}

// Every rank receives the owners' entries of x: after a row-partitioned operator (all colours) or after one colour of a
// Gauss-Seidel sweep.  One all-gather of equally sized (padded) slots; the slot layout follows gs_order, so packing and
// unpacking need no index lists beyond the per-(rank, colour) ranges.
template <class T>
void Ctx<T>::exchange(Level<T>& L, T* x, int colour, int ncomp)
{
    if (!L.part) return;
    if (halo_mode()) {
        halo_gather(L, x, colour, ncomp);
        return;
    }
    gather_colour(L, x, colour, ncomp);
}
template <class T>
void Ctx<T>::gather_all(Level<T>& L, T* x, int ncomp)
{
    if (L.part) gather_colour(L, x, -1, ncomp);
}
template <class T>
void Ctx<T>::gather_colour(Level<T>& L, T* x, int colour, int ncomp)
{
    CommTag tag(this, "allgather_vectors");
    const int R = comm.size, me = comm.rank;
    const int maxc = colour < 0 ? L.xmax_full : L.xmax_col[colour];
    if (maxc == 0) return;
    const size_t slot = (size_t)maxc * ncomp * sizeof(T);
    xsend.reserve(slot), xrecv.reserve(slot * R);
    HOT_LAUNCH(this, "xchg_pack", k_xchg_pack<T>, div_up(maxc, 256), 256, 0, x, L.gs_order.p, L.dxtab.p, R, me, colour, (T*)xsend.p, maxc, ncomp);
    c_allgather(xsend.p, xrecv.p, (int64_t)slot, true);
    HOT_LAUNCH(this, "xchg_unpack", k_xchg_unpack<T>, div_up((size_t)R * maxc, 256), 256, 0, x, L.gs_order.p, L.dxtab.p, R, me, colour, (const T*)xrecv.p, maxc, ncomp);
}

// ================================================================================================ halo mode
// hot_config.shard_replicated == 0.  The INDEX structure of the grid stays replicated (block list, node numbering, coordinates,
// colouring, coarse numbering, transfer tables: integers every rank derives from the same small all-gathered inputs), so every
// exchange list below is computed locally or with one handshake per step.  FIELD data is not replicated:
//   * node tiles (P2G, force, CN quantity, matrix-free products): the ranks whose particle groups cover a block exchange their
//     partial tiles pairwise and add them in ascending rank order — every sharer ends with the same bits, nobody else gets anything;
//   * DOF vectors: valid on the rows a rank owns and, after halo_gather, on the entries it reads (the 125-stencil of its rows, the
//     nodes of its particle tiles, the transfer-operator partners of its rows);
//   * reductions: local sums over owned rows, one all-reduce per batch of scalars (reduce_scalars).
// Bytes per exchange scale with the cut surface, not with the body.

// ------------------------------------------------------------------------------------------------ shared node tiles
template <class T>
__global__ void k_touch_blocks(const int32_t* __restrict__ group_nb, int ng, uint8_t* __restrict__ touch)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < ng * 8 && group_nb[e] >= 0) touch[group_nb[e]] = 1;
}
// fp64: a 4^3 colour block (the unit of row ownership) is two SPGrid blocks (2 x 4 x 4) side by side in x; a rank that covers one of
// them takes part in the sums of the other as well, so the owner of a colour block always holds the node data of all its rows
template <class T>
__global__ void k_touch_companions(HashMap h, const uint64_t* __restrict__ blocks, const uint8_t* __restrict__ touch, uint8_t* __restrict__ out, int nb)
{
    using G = Geo<T>;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb || !touch[b]) return;
    out[b] = 1;
    if (G::BX == 4) return;
    int i, j, k;
    G::linear_to_coord(blocks[b], i, j, k);
    const int32_t c = hash_find_id(h, G::linear_offset(i ^ 2, j, k) >> 12);
    if (c >= 0) out[c] = 1;
}
struct PeerSegs { // per peer: [begin, begin + count) of a list, and where the peer's packed payload starts (in list entries)
    int64_t lbeg[64], cnt[64], obeg[64];
};
// out[((k * q + a) * EPB) + e] = arrays[a][list[k] * EPB + e]
template <class T>
struct TileArrays {
    T* a[9];
};
template <class T>
__global__ void k_tile_pack(TileArrays<T> arr, int q, const int32_t* __restrict__ list, int64_t nlist, T* __restrict__ out)
{
    constexpr int EPB = Geo<T>::EPB;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nlist * q * EPB) return;
    const int64_t k = e / (q * EPB);
    const int r = (int)(e - k * q * EPB), a = r / EPB, l = r - a * EPB;
    out[e] = arr.a[a][(int64_t)list[k] * EPB + l];
}
// every block this rank shares: the sharers' partial tiles added in ascending rank order (its own at its rank's position)
template <class T>
__global__ void k_tile_sum(TileArrays<T> arr, int q, const uint64_t* __restrict__ sharers, const int32_t* __restrict__ tpos, const T* __restrict__ recv, PeerSegs seg, int R, int me, int nb)
{
    constexpr int EPB = Geo<T>::EPB;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)nb * q * EPB) return;
    const int b = (int)(e / (q * EPB));
    const uint64_t sh = sharers[b];
    if (!((sh >> me) & 1ULL) || (sh & (sh - 1)) == 0) return; // not mine, or nobody else's
    const int r = (int)(e - (int64_t)b * q * EPB), a = r / EPB, l = r - a * EPB;
    T* cell = arr.a[a] + (int64_t)b * EPB + l;
    T acc = (T)0;
    bool first = true;
    for (int p = 0; p < R; ++p) {
        if (!((sh >> p) & 1ULL)) continue;
        const T v = p == me ? *cell : recv[((seg.obeg[p] + tpos[(int64_t)p * nb + b]) * q + a) * EPB + l];
        acc = first ? v : acc + v;
        first = false;
    }
    *cell = acc;
}

template <class T>
void Ctx<T>::build_tile_plan()
{
    IndexPhase ip(this);
    const int R = comm.size, me = comm.rank;
    touch.reserve(2 * (size_t)Nb);
    HOT_HIP(hipMemsetAsync(touch.p, 0, 2 * (size_t)Nb, stream));
    HOT_LAUNCH(this, "shard_touch", k_touch_blocks<T>, div_up((size_t)Ng * 8, 256), 256, 0, group_nb.p, Ng, touch.p);
    HOT_LAUNCH(this, "shard_touch", k_touch_companions<T>, div_up(Nb, 256), 256, 0, block_map, blocks.p, touch.p, touch.p + Nb, Nb);
    std::vector<uint8_t> mine(Nb), all((size_t)Nb * R);
    HOT_HIP(hipMemcpyAsync(mine.data(), touch.p + Nb, Nb, hipMemcpyDeviceToHost, stream));
    HOT_HIP(hipMemcpyAsync(touch.p, touch.p + Nb, Nb, hipMemcpyDeviceToDevice, stream));
    sync();
    c_allgather(mine.data(), all.data(), Nb, false);
    std::vector<uint64_t> sh(Nb, 0);
    for (int r = 0; r < R; ++r)
        for (int b = 0; b < Nb; ++b)
            if (all[(size_t)r * Nb + b]) sh[b] |= 1ULL << r;
    std::vector<int32_t> pos((size_t)R * Nb, -1), list;
    tcnt.assign(R, 0), toff.assign(R, 0);
    for (int q = 0; q < R; ++q) {
        toff[q] = (int64_t)list.size();
        if (q == me) continue;
        for (int b = 0; b < Nb; ++b)
            if (((sh[b] >> q) & 1ULL) && ((sh[b] >> me) & 1ULL)) pos[(size_t)q * Nb + b] = (int32_t)tcnt[q]++, list.push_back(b);
    }
    sharers.reserve(Nb), tpos.reserve((size_t)R * Nb), tlist.reserve(std::max<size_t>(list.size(), 1));
    HOT_HIP(hipMemcpyAsync(sharers.p, sh.data(), (size_t)Nb * 8, hipMemcpyHostToDevice, stream));
    HOT_HIP(hipMemcpyAsync(tpos.p, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, stream));
    if (!list.empty()) HOT_HIP(hipMemcpyAsync(tlist.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, stream));
    sync(); // the host vectors go out of scope
}

// q slot arrays (Nb * EPB each) hold this rank's partial node sums; afterwards every block this rank covers holds the body's
template <class T>
void Ctx<T>::tile_exchange(T* const* arrays, int q)
{
    CommTag tag(this, "tiles");
    const int R = comm.size, me = comm.rank;
    HOT_CHECK(q >= 1 && q <= 9 && (int)tcnt.size() == R, HOT_ERR_INVALID, "tile_exchange: no tile plan (hot_sort)");
    TileArrays<T> arr{};
    for (int a = 0; a < q; ++a) arr.a[a] = arrays[a];
    const int64_t per = (int64_t)q * EPB; // values per shared block
    int64_t tot = 0;
    PeerSegs seg{};
    std::vector<int64_t> off(R, 0), bytes(R, 0);
    for (int p = 0; p < R; ++p) seg.lbeg[p] = toff[p], seg.cnt[p] = tcnt[p], seg.obeg[p] = tot, off[p] = tot * per * (int64_t)sizeof(T), bytes[p] = tcnt[p] * per * (int64_t)sizeof(T), tot += tcnt[p];
    // a rank that shares no block (an isolated body, an interior-only range) still enters the collective, with zero byte counts: "all ranks make
    // the same sequence of calls" is the hot_comm contract, and a true collective (MPI_Alltoallv, all_to_all_single) would otherwise wait for it
    xsend.reserve(std::max<size_t>((size_t)tot * per * sizeof(T), 16)), xrecv.reserve(std::max<size_t>((size_t)tot * per * sizeof(T), 16));
    if (tot > 0) HOT_LAUNCH(this, "tile_pack", k_tile_pack<T>, div_up((size_t)tot * per, 256), 256, 0, arr, q, tlist.p, tot, (T*)xsend.p);
    c_alltoallv(xsend.p, off.data(), bytes.data(), xrecv.p, off.data(), bytes.data()); // symmetric lists: what goes to a peer and what comes from it have the same layout
    if (tot > 0) HOT_LAUNCH(this, "tile_sum", k_tile_sum<T>, div_up((size_t)Nb * per, 256), 256, 0, arr, q, sharers.p, tpos.p, (const T*)xrecv.p, seg, R, me, Nb);
}

// ------------------------------------------------------------------------------------------------ DOF-vector halos
__global__ void k_mark_stencil(HashMap h, const int32_t* __restrict__ coord, const uint8_t* __restrict__ own, uint8_t* __restrict__ need, int n)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * 125) return;
    const int i = (int)(e / 125), k = (int)(e - (int64_t)i * 125);
    if (!own[i]) return;
    const int x = coord[3 * i] + k / 25 - 2, y = coord[3 * i + 1] + (k / 5) % 5 - 2, z = coord[3 * i + 2] + k % 5 - 2;
    if ((x | y | z) < 0) return;
    const int32_t j = hash_find_id(h, coord_key(x, y, z));
    if (j >= 0 && !own[j]) need[j] = 1;
}
// flags over the POSITIONS of gs_order (colour-major): the halo entries owned by rank q
__global__ void k_need_flags(const int32_t* __restrict__ gs_order, const uint8_t* __restrict__ need, const uint8_t* __restrict__ owner, int q, int32_t* __restrict__ flags, int n)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = gs_order[p];
    flags[p] = (need[i] && owner[i] == (uint8_t)q) ? 1 : 0;
}
__global__ void k_need_compact(const int32_t* __restrict__ gs_order, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int32_t* __restrict__ out, int n)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && flags[p]) out[scan[p]] = gs_order[p];
}
template <class T>
__global__ void k_halo_pack(const T* __restrict__ x, const int32_t* __restrict__ list, int64_t cnt, int ncomp, T* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cnt * ncomp) return;
    const int64_t k = e / ncomp;
    out[e] = x[(int64_t)list[k] * ncomp + (e - k * ncomp)];
}
template <class T>
__global__ void k_halo_unpack(T* __restrict__ x, const int32_t* __restrict__ list, int64_t cnt, int ncomp, const T* __restrict__ in)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cnt * ncomp) return;
    const int64_t k = e / ncomp;
    x[(int64_t)list[k] * ncomp + (e - k * ncomp)] = in[e];
}

template <class T>
void Ctx<T>::mark_stencil(Level<T>& L, uint8_t* need)
{
    HOT_LAUNCH(this, "halo_mark", k_mark_stencil, div_up((size_t)L.n * 125, 256), 256, 0, L.map, L.coord.p, L.own.p, need, L.n);
}

// The exchange lists of one level.  `need` = entries read here and owned elsewhere: the 125-stencil of the owned rows (needs L.map)
// plus whatever mark_extra adds.  The readers tell the owners once per step (counts, then ids); both sides keep the lists in
// (peer, colour, gs_order position) order, so a colour of a Gauss-Seidel sweep is one contiguous piece per peer.
template <class T>
void Ctx<T>::build_halo(Level<T>& L, const std::function<void(uint8_t*)>& mark_extra)
{
    IndexPhase ip(this);
    HOT_CHECK(L.part && L.colored, HOT_ERR_INVALID, "build_halo: the level is not partitioned");
    const int R = comm.size, me = comm.rank, n = L.n;
    auto& H = L.halo;
    H.need.reserve(n);
    HOT_HIP(hipMemsetAsync(H.need.p, 0, n, stream));
    mark_stencil(L, H.need.p);
    if (mark_extra) mark_extra(H.need.p);
    // positions of gs_order where every colour starts
    std::vector<int32_t> bstart(L.nblocks + 1);
    HOT_HIP(hipMemcpyAsync(bstart.data(), L.gs_block_start.p, (size_t)(L.nblocks + 1) * 4, hipMemcpyDeviceToHost, stream));
    sync();
    int cpos[9];
    for (int c = 0; c <= 8; ++c) cpos[c] = bstart[L.color_block_begin[c]];
    flags.reserve(n + 1), scan.reserve(n + 1);
    H.rcnt.assign(R, 0), H.roff.assign(R, 0), H.rcol.assign((size_t)R * 9, 0);
    H.recv.reserve(std::max(n, 1));
    int64_t rt = 0;
    std::vector<int32_t> hscan(9);
    for (int q = 0; q < R; ++q) {
        H.roff[q] = rt;
        if (q == me) continue;
        HOT_LAUNCH(this, "halo_flags", k_need_flags, div_up(n, 256), 256, 0, L.gs_order.p, H.need.p, L.owner.p, q, flags.p, n);
        const int c = exclusive_scan_i32(flags.p, scan.p, n);
        if (c > 0) {
            HOT_LAUNCH(this, "halo_compact", k_need_compact, div_up(n, 256), 256, 0, L.gs_order.p, flags.p, scan.p, H.recv.p + rt, n);
            for (int k = 0; k < 8; ++k)
                if (cpos[k] < n)
                    HOT_HIP(hipMemcpyAsync(&hscan[k], scan.p + cpos[k], 4, hipMemcpyDeviceToHost, stream));
                else
                    hscan[k] = c;
            sync();
            for (int k = 0; k < 8; ++k) H.rcol[(size_t)q * 9 + k] = hscan[k];
            H.rcol[(size_t)q * 9 + 8] = c;
        }
        H.rcnt[q] = c, rt += c;
    }
    H.rtot = rt;
    // handshake: every rank learns, per colour, how many of its rows each peer reads; then which
    std::vector<int64_t> mine((size_t)R * 9), all((size_t)R * R * 9);
    for (size_t k = 0; k < mine.size(); ++k) mine[k] = H.rcol[k];
    c_allgather(mine.data(), all.data(), (int64_t)R * 9 * sizeof(int64_t), false);
    H.scnt.assign(R, 0), H.soff.assign(R, 0), H.scol.assign((size_t)R * 9, 0);
    int64_t stot = 0;
    for (int p = 0; p < R; ++p) {
        H.soff[p] = stot;
        if (p == me) continue;
        for (int k = 0; k < 9; ++k) H.scol[(size_t)p * 9 + k] = all[((size_t)p * R + me) * 9 + k];
        H.scnt[p] = H.scol[(size_t)p * 9 + 8], stot += H.scnt[p];
    }
    H.stot = stot;
    H.send.reserve(std::max<int64_t>(stot, 1));
    {
        std::vector<int64_t> so(R), sb(R), ro(R), rb(R);
        for (int p = 0; p < R; ++p) so[p] = H.roff[p] * 4, sb[p] = H.rcnt[p] * 4, ro[p] = H.soff[p] * 4, rb[p] = H.scnt[p] * 4;
        c_alltoallv(H.recv.p, so.data(), sb.data(), H.send.p, ro.data(), rb.data()); // my read lists go out, the peers' read lists of my rows come in
    }
    sync();
    H.built = true;
}

template <class T>
void Ctx<T>::halo_gather(Level<T>& L, T* x, int colour, int ncomp)
{
    if (!L.part) return;
    CommTag tag(this, "halos");
    auto& H = L.halo;
    HOT_CHECK(H.built, HOT_ERR_INVALID, "halo_gather: the level has no exchange lists");
    const int R = comm.size;
    std::vector<int64_t> so(R, 0), sb(R, 0), ro(R, 0), rb(R, 0), sl(R, 0), rl(R, 0), sc(R, 0), rc(R, 0);
    int64_t st = 0, rt = 0;
    const int64_t eb = (int64_t)ncomp * sizeof(T);
    for (int p = 0; p < R; ++p) {
        const int64_t s0 = colour < 0 ? 0 : H.scol[(size_t)p * 9 + colour], s1 = colour < 0 ? H.scnt[p] : H.scol[(size_t)p * 9 + colour + 1];
        const int64_t r0 = colour < 0 ? 0 : H.rcol[(size_t)p * 9 + colour], r1 = colour < 0 ? H.rcnt[p] : H.rcol[(size_t)p * 9 + colour + 1];
        sl[p] = H.soff[p] + s0, sc[p] = s1 - s0, so[p] = st * eb, sb[p] = sc[p] * eb, st += sc[p];
        rl[p] = H.roff[p] + r0, rc[p] = r1 - r0, ro[p] = rt * eb, rb[p] = rc[p] * eb, rt += rc[p];
    }
    xsend.reserve((size_t)std::max<int64_t>(st, 1) * eb), xrecv.reserve((size_t)std::max<int64_t>(rt, 1) * eb);
    for (int p = 0; p < R; ++p)
        if (sc[p] > 0) HOT_LAUNCH(this, "halo_pack", k_halo_pack<T>, div_up((size_t)sc[p] * ncomp, 256), 256, 0, x, H.send.p + sl[p], sc[p], ncomp, (T*)(xsend.p + so[p]));
    c_alltoallv(xsend.p, so.data(), sb.data(), xrecv.p, ro.data(), rb.data());
    for (int p = 0; p < R; ++p)
        if (rc[p] > 0) HOT_LAUNCH(this, "halo_unpack", k_halo_unpack<T>, div_up((size_t)rc[p] * ncomp, 256), 256, 0, x, H.recv.p + rl[p], rc[p], ncomp, (const T*)(xrecv.p + ro[p]));
}

// ------------------------------------------------------------------------------------------------ partial matrix rows
__global__ void k_rows_flag(const uint8_t* __restrict__ touched, const uint8_t* __restrict__ owner, int q, int32_t* flags, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (touched[i] && owner[i] == (uint8_t)q) ? 1 : 0;
}
__global__ void k_rows_compact(const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int32_t* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) out[scan[i]] = i;
}
template <class T>
__global__ void k_rows_pack(const T* __restrict__ val, const int32_t* __restrict__ rows, int64_t nrows, T* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nrows * 1125) return;
    const int64_t k = e / 1125;
    out[e] = val[(int64_t)rows[k] * 1125 + (e - k * 1125)];
}
template <class T>
__global__ void k_rows_add(T* __restrict__ val, const int32_t* __restrict__ rows, int64_t nrows, const T* __restrict__ in)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nrows * 1125) return;
    const int64_t k = e / 1125;
    val[(int64_t)rows[k] * 1125 + (e - k * 1125)] += in[e];
}

// Rows of L's matrix for which this rank holds a partial sum (`touched`) but which another rank owns are sent to the owner and
// added there, source ranks in ascending order (deterministic).  Used for the level-0 Hessian (rows near the shard boundary get
// contributions from both sides' particles) and for the Galerkin products (a coarse row sums over fine rows of several owners).
template <class T>
void Ctx<T>::exchange_rows(Level<T>& L, const uint8_t* touched)
{
    CommTag tag(this, L.id == 0 ? "rows_level0" : "rows_coarse");
    const int R = comm.size, me = comm.rank, n = L.n;
    flags.reserve(n), scan.reserve(n);
    DBuf<int32_t> sendrows, recvrows;
    sendrows.reserve(n);
    std::vector<int64_t> scnt(R, 0), soff(R, 0);
    int64_t off = 0;
    for (int q = 0; q < R; ++q) {
        soff[q] = off;
        if (q == me) continue;
        HOT_LAUNCH(this, "rows_flag", k_rows_flag, div_up(n, 256), 256, 0, touched, L.owner.p, q, flags.p, n);
        const int c = exclusive_scan_i32(flags.p, scan.p, n);
        if (c > 0) HOT_LAUNCH(this, "rows_compact", k_rows_compact, div_up(n, 256), 256, 0, flags.p, scan.p, sendrows.p + off, n);
        scnt[q] = c, off += c;
    }
    std::vector<int64_t> all((size_t)R * R, 0);
    c_allgather(scnt.data(), all.data(), (int64_t)R * sizeof(int64_t), false);
    std::vector<int64_t> rcnt(R, 0), roff(R, 0);
    int64_t rtot = 0;
    for (int s = 0; s < R; ++s) roff[s] = rtot, rcnt[s] = all[(size_t)s * R + me], rtot += rcnt[s];
    recvrows.reserve(std::max<int64_t>(rtot, 1));
    auto scaled = [&](const std::vector<int64_t>& v, int64_t f) {
        std::vector<int64_t> o(v);
        for (auto& x : o) x *= f;
        return o;
    };
    {
        auto so = scaled(soff, 4), sb = scaled(scnt, 4), ro = scaled(roff, 4), rb = scaled(rcnt, 4);
        c_alltoallv(sendrows.p, so.data(), sb.data(), recvrows.p, ro.data(), rb.data());
    }
    const int64_t rowbytes = 1125 * (int64_t)sizeof(T);
    xsend.reserve((size_t)std::max<int64_t>(off, 1) * rowbytes), xrecv.reserve((size_t)std::max<int64_t>(rtot, 1) * rowbytes);
    if (off > 0) HOT_LAUNCH(this, "rows_pack", k_rows_pack<T>, div_up((size_t)off * 1125, 256), 256, 0, L.val.p, sendrows.p, off, (T*)xsend.p);
    {
        auto so = scaled(soff, rowbytes), sb = scaled(scnt, rowbytes), ro = scaled(roff, rowbytes), rb = scaled(rcnt, rowbytes);
        c_alltoallv(xsend.p, so.data(), sb.data(), xrecv.p, ro.data(), rb.data());
    }
    for (int s = 0; s < R; ++s)
        if (rcnt[s] > 0)
            HOT_LAUNCH(this, "rows_add", k_rows_add<T>, div_up((size_t)rcnt[s] * 1125, 256), 256, 0, L.val.p, recvrows.p + roff[s], rcnt[s], (const T*)xrecv.p + roff[s] * 1125);
    sync(); // the local buffers go out of scope
}

// ------------------------------------------------------------------------------------------------ particle migration
__global__ void k_idkeys(const int32_t* __restrict__ gid, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) keys[p] = (uint64_t)(uint32_t)gid[p], vals[p] = (uint32_t)p;
}
// SPGrid page id of every particle (the high part of the sort key, MpmSimulationBase.cpp:1080-1085)
template <class T>
__global__ void k_page_ids(const T* __restrict__ X, uint64_t* __restrict__ page, int64_t n, T one_over_dx)
{
    using G = Geo<T>;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int b0 = base_node_of<T>(one_over_dx, X[p]), b1 = base_node_of<T>(one_over_dx, X[n + p]), b2 = base_node_of<T>(one_over_dx, X[2 * n + p]);
    page[p] = G::linear_offset(b0, b1, b2) >> 12;
}
__global__ void k_page_sample(const uint64_t* __restrict__ page, int64_t n, uint64_t* __restrict__ out, int S)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) out[i] = page[(int64_t)(((__int128)i * n) / S)];
}
struct Splitters {
    uint64_t v[63]; // rank r holds the pages [v[r-1], v[r]) ; v[-1] = 0, v[size-1] = inf
    int n;
};
__global__ void k_dest_flags(const uint64_t* __restrict__ page, int64_t n, Splitters sp, int q, int32_t* __restrict__ flags)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int d = 0;
    while (d < sp.n && page[p] >= sp.v[d]) ++d;
    flags[p] = d == q ? 1 : 0;
}
// particle-major records of `comps` scalars: out[k * comps + c] = a[c * n + list[k]]
template <class U>
__global__ void k_pack_attr(const U* __restrict__ a, int64_t n, int comps, const int32_t* __restrict__ list, int64_t cnt, U* __restrict__ out, int stride, int off)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cnt * comps) return;
    const int64_t k = e / comps;
    const int c = (int)(e - k * comps);
    out[k * stride + off + c] = a[(int64_t)c * n + list[k]];
}
// new SoA array: [0, kept) gathered from the old one through `keep`, [kept, nnew) from the received records
template <class U>
__global__ void k_build_attr(const U* __restrict__ old, int64_t nold, int comps, const int32_t* __restrict__ keep, int64_t kept, const U* __restrict__ recv, int stride, int off, U* __restrict__ out, int64_t nnew)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnew * comps) return;
    const int c = (int)(e / nnew);
    const int64_t p = e - (int64_t)c * nnew;
    out[e] = p < kept ? old[(int64_t)c * nold + keep[p]] : recv[(p - kept) * stride + off + c];
}
__global__ void k_rank_of(const uint32_t* __restrict__ sorted_slot, int32_t* __restrict__ slot2orig, int64_t n)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) slot2orig[sorted_slot[k]] = (int32_t)k; // hot_get_particles order = ascending global id
}

// Every hot_sort of a sharded context first re-shards the particles: rank r gets the r-th of `size` nearly equal runs of
// the SPGrid page order (splitters = quantiles of a strided sample of all ranks' page ids, whole pages only), so a shard stays
// spatially compact however the body moves, the ranks' group lists stay contiguous ranges of the global sort, and the rows a
// rank's particles touch stay (mostly) the rows it owns.  Particles travel as records of their 29 scalars + their global id.
template <class T>
void Ctx<T>::migrate_particles()
{
    CommTag tag(this, "migration");
    const int R = comm.size, me = comm.rank;
    constexpr int S = 1024, NC = 29;
    const int64_t n = Np;
    keys.reserve(std::max<int64_t>(n, S)), flags.reserve(std::max<size_t>(n, 64)), scan.reserve(std::max<size_t>(n, 64));
    uint64_t* page = keys.p;
    HOT_LAUNCH(this, "migrate_page_ids", k_page_ids<T>, div_up(n, 256), 256, 0, pX.p, page, n, (T)1 / dx);
    // ---- splitters from a strided sample of every rank's pages; a rank's particle count rides behind its samples (one all-gather, not two)
    xsend.reserve((size_t)(S + 1) * 8), xrecv.reserve((size_t)(S + 1) * 8 * R);
    HOT_LAUNCH(this, "migrate_sample", k_page_sample, div_up(S, 256), 256, 0, page, n, (uint64_t*)xsend.p, S);
    const uint64_t my_count = (uint64_t)n; // (lives until the collective has synchronised the stream)
    HOT_HIP(hipMemcpyAsync((uint64_t*)xsend.p + S, &my_count, 8, hipMemcpyHostToDevice, stream));
    c_allgather(xsend.p, xrecv.p, (int64_t)(S + 1) * 8, true);
    std::vector<uint64_t> gathered((size_t)(S + 1) * R), samp((size_t)S * R);
    HOT_HIP(hipMemcpyAsync(gathered.data(), xrecv.p, gathered.size() * 8, hipMemcpyDeviceToHost, stream));
    sync();
    // every rank contributes S samples whatever it holds: a sample of rank r stands for Np_r / S particles.  Splitters = weighted
    // quantiles, so that a skewed distribution (a body drifting out of some ranks' page ranges) is re-balanced instead of re-derived
    std::vector<int64_t> cnts(R, 0);
    for (int r = 0; r < R; ++r) {
        std::copy(gathered.begin() + (size_t)r * (S + 1), gathered.begin() + (size_t)r * (S + 1) + S, samp.begin() + (size_t)r * S);
        cnts[r] = (int64_t)gathered[(size_t)r * (S + 1) + S];
    }
    std::vector<std::pair<uint64_t, int64_t>> ws; // (page, weight = particles of the sample's rank; every rank has S samples) — integers: exact quantiles
    ws.reserve(samp.size());
    __int128 wtot = 0;
    for (int r = 0; r < R; ++r)
        for (int k = 0; k < S; ++k) ws.emplace_back(samp[(size_t)r * S + k], cnts[r]), wtot += cnts[r];
    std::sort(ws.begin(), ws.end(), [](const std::pair<uint64_t, int64_t>& a, const std::pair<uint64_t, int64_t>& b) { return a.first < b.first; });
    Splitters sp{};
    sp.n = R - 1;
    {
        __int128 acc = 0;
        size_t k = 0;
        uint64_t prev = 0;
        for (int r = 1; r < R; ++r) {
            const __int128 want = wtot * r / R;
            while (k < ws.size() && acc + ws[k].second <= want) acc += ws[k].second, ++k;
            uint64_t v = k < ws.size() ? ws[k].first : ws.back().first + 1;
            if (r > 1 && v <= prev) { // equal splitters (few pages, many ranks): advance to the next distinct page so that no range is empty by construction
                size_t j = k;
                while (j < ws.size() && ws[j].first <= prev) ++j;
                v = j < ws.size() ? ws[j].first : prev + 1;
            }
            sp.v[r - 1] = prev = v;
        }
    }
    page_split.assign(sp.v, sp.v + sp.n);
    // ---- who goes where: one compacted list per destination (ascending slot order), the kept particles included
    DBuf<int32_t> lists;
    lists.reserve(n);
    std::vector<int64_t> cnt(R, 0), off(R, 0);
    int64_t o = 0;
    for (int q = 0; q < R; ++q) {
        off[q] = o;
        HOT_LAUNCH(this, "migrate_flags", k_dest_flags, div_up(n, 256), 256, 0, page, n, sp, q, flags.p);
        const int c = exclusive_scan_i32(flags.p, scan.p, n);
        if (c > 0) HOT_LAUNCH(this, "rows_compact", k_rows_compact, div_up(n, 256), 256, 0, flags.p, scan.p, lists.p + o, (int)n);
        cnt[q] = c, o += c;
    }
    std::vector<int64_t> all((size_t)R * R, 0);
    c_allgather(cnt.data(), all.data(), (int64_t)R * sizeof(int64_t), false);
    int64_t moving = 0, incoming = 0;
    for (int a = 0; a < R; ++a)
        for (int b = 0; b < R; ++b)
            if (a != b) moving += all[(size_t)a * R + b];
    if (moving == 0) return; // nobody changes rank this step
    const int64_t kept = cnt[me];
    std::vector<int64_t> scnt(cnt), soff(off), rcnt(R, 0), roff(R, 0);
    scnt[me] = 0;
    for (int src = 0; src < R; ++src) {
        roff[src] = incoming;
        if (src != me) rcnt[src] = all[(size_t)src * R + me], incoming += rcnt[src];
    }
    const int64_t nnew = kept + incoming, nout = n - kept;
    HOT_CHECK(nnew > 0, HOT_ERR_INVALID, "particle migration left this rank without particles (fewer occupied SPGrid pages than ranks: use fewer ranks for a body this small)");
    constexpr int index_bits = 32 - G::block_bits;
    HOT_CHECK(nnew < (1LL << index_bits), HOT_ERR_CAPACITY, "particle count of this rank exceeds 2^(32-block_bits) after migration");
    // ---- outgoing records: [NC scalars] per particle in list order (the kept run of the list is skipped by the offsets), ids apart
    DBuf<T> sendT, recvT;
    DBuf<int32_t> sendI, recvI;
    sendT.reserve((size_t)std::max<int64_t>(n, 1) * NC), recvT.reserve((size_t)std::max<int64_t>(incoming, 1) * NC);
    sendI.reserve(std::max<int64_t>(n, 1)), recvI.reserve(std::max<int64_t>(incoming, 1));
    struct Attr {
        DBuf<T>* a;
        int comps;
    } attrs[9] = { { &pX, 3 }, { &pV, 3 }, { &pM, 1 }, { &pVol, 1 }, { &pMu, 1 }, { &pLam, 1 }, { &pJp, 1 }, { &pC, 9 }, { &pF, 9 } };
    {
        int col = 0;
        for (auto& at : attrs) {
            HOT_LAUNCH(this, "migrate_pack", k_pack_attr<T>, div_up((size_t)n * at.comps, 256), 256, 0, at.a->p, n, at.comps, lists.p, n, sendT.p, NC, col);
            col += at.comps;
        }
        HOT_LAUNCH(this, "migrate_pack", k_pack_attr<int32_t>, div_up(n, 256), 256, 0, pGid.p, n, 1, lists.p, n, sendI.p, 1, 0);
    }
    auto scaled = [&](const std::vector<int64_t>& v, int64_t f) {
        std::vector<int64_t> r(v);
        for (auto& x : r) x *= f;
        return r;
    };
    {
        auto so = scaled(soff, NC * (int64_t)sizeof(T)), sb = scaled(scnt, NC * (int64_t)sizeof(T)), ro = scaled(roff, NC * (int64_t)sizeof(T)), rb = scaled(rcnt, NC * (int64_t)sizeof(T));
        c_alltoallv(sendT.p, so.data(), sb.data(), recvT.p, ro.data(), rb.data());
        auto so4 = scaled(soff, 4), sb4 = scaled(scnt, 4), ro4 = scaled(roff, 4), rb4 = scaled(rcnt, 4);
        c_alltoallv(sendI.p, so4.data(), sb4.data(), recvI.p, ro4.data(), rb4.data());
    }
    (void)nout;
    // ---- the new particle set: kept particles (in their old order) followed by the arrivals (by source rank)
    const int32_t* keep = lists.p + off[me];
    {
        std::vector<DBuf<T>> fresh(9);
        int col = 0, k = 0;
        for (auto& at : attrs) {
            fresh[k].reserve((size_t)nnew * at.comps, 1.25);
            HOT_LAUNCH(this, "migrate_build", k_build_attr<T>, div_up((size_t)nnew * at.comps, 256), 256, 0, at.a->p, n, at.comps, keep, kept, recvT.p, NC, col, fresh[k].p, nnew);
            col += at.comps, ++k;
        }
        DBuf<int32_t> gid;
        gid.reserve(nnew, 1.25);
        HOT_LAUNCH(this, "migrate_build", k_build_attr<int32_t>, div_up(nnew, 256), 256, 0, pGid.p, n, 1, keep, kept, recvI.p, 1, 0, gid.p, nnew);
        sync();
        k = 0;
        for (auto& at : attrs) std::swap(at.a->p, fresh[k].p), std::swap(at.a->cap, fresh[k].cap), ++k;
        std::swap(pGid.p, gid.p), std::swap(pGid.cap, gid.cap);
    }
    Np = nnew;
    reserve_particles(nnew); // the scratch / derived per-particle buffers (the nine attribute arrays above are large enough already)
    // ---- hot_get_particles order: ascending global id
    keys.reserve(nnew), keys2.reserve(nnew), vals.reserve(nnew), vals2.reserve(nnew);
    {
        // keys = id (as 64-bit), vals = slot
        HOT_LAUNCH(this, "migrate_idkeys", k_idkeys, div_up(nnew, 256), 256, 0, pGid.p, keys.p, vals.p, nnew);
        size_t bytes = 0;
        HOT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys.p, keys2.p, vals.p, vals2.p, (size_t)nnew, 0, 32, stream));
        if (bytes > sort_tmp_bytes) {
            sort_tmp.reserve(bytes);
            sort_tmp_bytes = sort_tmp.cap;
        }
        HOT_HIP(rocprim::radix_sort_pairs(sort_tmp.p, bytes, keys.p, keys2.p, vals.p, vals2.p, (size_t)nnew, 0, 32, stream));
        HOT_LAUNCH(this, "migrate_rank_of", k_rank_of, div_up(nnew, 256), 256, 0, vals2.p, slot2orig.p, nnew);
    }
    sync();
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
