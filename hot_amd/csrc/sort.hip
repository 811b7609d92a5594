// libhotmi355x — particle storage, sort/bin, block list, neighbour tables.
//
// Replaces MpmSimulationBase::sortParticlesAndPolluteGrid (reference Lib/MPM/MpmSimulationBase.cpp:1066-1137):
//   key build (:1080-1085)           -> k_make_keys (same float arithmetic: X * (1/dx), baseNode, Linear_Offset)
//   tbb::parallel_sort (:1087)       -> LSD radix sort of the 64-bit keys (rocPRIM device primitive)
//   serial group scan (:1089-1097)   -> head flags + exclusive scan
//   serial Set_Page loop (:1099-1125)-> hash insert with atomicMin(sequence rank): a page's position in the
//                                       block list is the rank of its FIRST Set_Page call, which is what the
//                                       reference's insertion-ordered std::vector records (SPGrid_Page_Map.h:61-70)
//   serial memset (:1126-1136)       -> tile clears
// MI355X design: the particle arrays are physically permuted into sorted order every step (one gather pass),
// so every later particle kernel streams them fully coalesced instead of chasing particle_order.
#include "hot_impl.h"
#include <rocprim/rocprim.hpp>

namespace hot {

template <class T>
__global__ void k_aos_to_soa(const T* __restrict__ aos, T* __restrict__ soa, int64_t n, int comps)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * comps) return;
    int64_t p = i / comps;
    int c = (int)(i - p * comps);
    soa[(int64_t)c * n + p] = aos[i];
}
// aos[orig*comps + c] = soa[c*n + slot], orig = slot2orig[slot]
template <class T>
__global__ void k_soa_to_aos(const T* __restrict__ soa, T* __restrict__ aos, const int32_t* __restrict__ slot2orig, int64_t n, int comps)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * comps) return;
    int c = (int)(i / n);
    int64_t p = i - (int64_t)c * n;
    aos[(int64_t)slot2orig[p] * comps + c] = soa[i];
}
template <class T>
__global__ void k_fill(T* a, int64_t n, T v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}
__global__ void k_iota(int32_t* a, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (int32_t)i;
}
template <class T>
__global__ void k_identity9(T* a, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 9) return;
    int c = (int)(i / n);
    a[i] = (c == 0 || c == 4 || c == 8) ? (T)1 : (T)0;
}

template <class T>
__global__ void k_make_keys(const T* __restrict__ X, const int32_t* __restrict__ slot2orig, uint64_t* keys, uint32_t* vals, int64_t n, T one_over_dx)
{
    using G = Geo<T>;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int b0 = base_node_of<T>(one_over_dx, X[p]), b1 = base_node_of<T>(one_over_dx, X[n + p]), b2 = base_node_of<T>(one_over_dx, X[2 * n + p]);
    uint64_t offset = G::linear_offset(b0, b1, b2);
    constexpr int index_bits = 32 - G::block_bits;
    keys[p] = ((offset >> G::data_bits) << index_bits) + (uint64_t)(uint32_t)slot2orig[p];
    vals[p] = (uint32_t)p;
}

// Sharded runs break ties inside a cell by the GLOBAL particle id, which need not fit the 32 - block_bits index field of the key (64 M
// particles per GPU of a weak-scaled body exceed it on the second rank): two passes instead — by id, then by (cell, rank of the id on
// this rank) — give the same order for ids of any size.
__global__ void k_id_keys(const int32_t* __restrict__ gid, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) keys[p] = (uint64_t)(uint32_t)gid[p], vals[p] = (uint32_t)p;
}
template <class T>
__global__ void k_make_keys_in_order(const T* __restrict__ X, const uint32_t* __restrict__ visit, uint64_t* keys, uint32_t* vals, int64_t n, T one_over_dx)
{
    using G = Geo<T>;
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t p = visit[k];
    int b0 = base_node_of<T>(one_over_dx, X[p]), b1 = base_node_of<T>(one_over_dx, X[n + p]), b2 = base_node_of<T>(one_over_dx, X[2 * n + p]);
    constexpr int index_bits = 32 - G::block_bits;
    keys[k] = ((G::linear_offset(b0, b1, b2) >> G::data_bits) << index_bits) + (uint64_t)k; // k = the particle's rank by id on this rank (< Np < 2^index_bits)
    vals[k] = (uint32_t)p;
}

template <class U>
__global__ void k_gather(const U* __restrict__ src, U* __restrict__ dst, const uint32_t* __restrict__ perm, int64_t n, int comps)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int64_t q = perm[p];
    for (int c = 0; c < comps; ++c) dst[(int64_t)c * n + p] = src[(int64_t)c * n + q];
}

__global__ void k_group_heads(const uint64_t* __restrict__ keys, int32_t* flags, int64_t n)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    flags[p] = (p == 0 || (keys[p] >> 32) != (keys[p - 1] >> 32)) ? 1 : 0;
}
template <class T>
__global__ void k_fill_groups(const uint64_t* __restrict__ keys, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, int32_t* group_first,
    uint64_t* group_page, int32_t* group_origin, int64_t n, int ng)
{
    using G = Geo<T>;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (p == 0) group_first[ng] = (int32_t)n;
    if (!flags[p]) return;
    int g = scan[p];
    group_first[g] = (int32_t)p;
    uint64_t page = keys[p] >> 32;
    group_page[g] = page;
    int i, j, k;
    G::linear_to_coord(page << 12, i, j, k);
    group_origin[3 * g] = i, group_origin[3 * g + 1] = j, group_origin[3 * g + 2] = k;
}

__global__ void k_hash_clear(HashMap h)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > h.mask) return;
    h.keys[i] = ~0ULL;
    h.minrank[i] = ~0ULL;
    h.id[i] = -1;
}
template <class T>
__device__ inline uint64_t nb_page(const int32_t* origin, int g, int n)
{
    using G = Geo<T>;
    int a = n >> 2, b = (n >> 1) & 1, c = n & 1;
    return G::linear_offset(origin[3 * g] + a * G::BX, origin[3 * g + 1] + b * G::BY, origin[3 * g + 2] + c * G::BZ) >> 12;
}
template <class T>
__global__ void k_block_insert(HashMap h, const int32_t* __restrict__ origin, int ng)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ng * 8) return;
    hash_insert_min(h, nb_page<T>(origin, s >> 3, s & 7), (unsigned long long)s);
}
template <class T>
__global__ void k_block_flag(HashMap h, const int32_t* __restrict__ origin, int32_t* flags, int ng)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ng * 8) return;
    int32_t slot = hash_find_slot(h, nb_page<T>(origin, s >> 3, s & 7));
    flags[s] = (h.minrank[slot] == (unsigned long long)s) ? 1 : 0;
}
template <class T>
__global__ void k_block_assign(HashMap h, const int32_t* __restrict__ origin, const int32_t* __restrict__ flags, const int32_t* __restrict__ scan, uint64_t* blocks, int ng)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ng * 8 || !flags[s]) return;
    uint64_t page = nb_page<T>(origin, s >> 3, s & 7);
    int32_t slot = hash_find_slot(h, page);
    h.id[slot] = scan[s];
    blocks[scan[s]] = page << 12;
}
template <class T>
__global__ void k_group_nb(HashMap h, const int32_t* __restrict__ origin, int32_t* group_nb, int ng)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ng * 8) return;
    group_nb[s] = hash_find_id(h, nb_page<T>(origin, s >> 3, s & 7));
}
template <class T>
__global__ void k_base_offsets(const T* __restrict__ X, const int32_t* __restrict__ slot2orig, uint64_t* out, int32_t* order, int64_t n, T one_over_dx)
{
    using G = Geo<T>;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int b0 = base_node_of<T>(one_over_dx, X[p]), b1 = base_node_of<T>(one_over_dx, X[n + p]), b2 = base_node_of<T>(one_over_dx, X[2 * n + p]);
    uint64_t offset = G::linear_offset(b0, b1, b2);
    offset = (offset >> G::data_bits) << G::data_bits;
    int32_t o = slot2orig[p];
    out[o] = offset;
    order[p] = o;
}
__global__ void k_block_group(const int32_t* __restrict__ group_nb, int32_t* block_group, int ng)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ng) block_group[group_nb[g * 8]] = g; // slot 0 of a group's neighbour table is its own page
}
template <class T>
__global__ void k_block_rev(HashMap h, const uint64_t* __restrict__ blocks, const int32_t* __restrict__ block_group, int32_t* block_rev, int nb)
{
    using G = Geo<T>;
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nb * 8) return;
    int b = s >> 3, n = s & 7;
    int i, j, k;
    G::linear_to_coord(blocks[b], i, j, k);
    i -= (n >> 2) * G::BX, j -= ((n >> 1) & 1) * G::BY, k -= (n & 1) * G::BZ;
    int r = -1;
    if ((i | j | k) >= 0) {
        int32_t bid = hash_find_id(h, G::linear_offset(i, j, k) >> 12);
        if (bid >= 0) r = block_group[bid];
    }
    block_rev[s] = r;
}

// out_q[slot] = sum over the <= 8 particle groups whose partial tile covers the node, fixed order n = 0..7
template <class T>
__global__ void k_tile_reduce(const T* __restrict__ part, int Q, const int32_t* __restrict__ block_rev, T* o0, T* o1, T* o2, T* o3, T* o4, int nb)
{
    using G = Geo<T>;
    constexpr int TX = G::BX + 2, TY = G::BY + 2, TZ = G::BZ + 2, TILE = TX * TY * TZ;
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (int64_t)nb * G::EPB) return;
    int b = (int)(s / G::EPB), e = (int)(s - (int64_t)b * G::EPB);
    int ez = e & (G::BZ - 1), ey = (e >> G::zb) & (G::BY - 1), ex = e >> (G::zb + G::yb);
    T acc[5] = { 0, 0, 0, 0, 0 };
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        int g = block_rev[b * 8 + n];
        if (g < 0) continue;
        int tx = ex + (n >> 2) * G::BX, ty = ey + ((n >> 1) & 1) * G::BY, tz = ez + (n & 1) * G::BZ;
        if (tx >= TX || ty >= TY || tz >= TZ) continue;
        const T* p = part + (int64_t)g * Q * TILE + (tx * TY + ty) * TZ + tz;
        for (int q = 0; q < Q; ++q) acc[q] += p[q * TILE];
    }
    o0[s] = acc[0];
    if (Q > 1) o1[s] = acc[1];
    if (Q > 2) o2[s] = acc[2];
    if (Q > 3) o3[s] = acc[3];
    if (Q > 4) o4[s] = acc[4];
}

__global__ void k_group_ranges(const int32_t* __restrict__ first, int32_t* out, int ng)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    out[2 * g] = first[g];
    out[2 * g + 1] = first[g + 1] - 1;
}

// ------------------------------------------------------------------------------------------------ Ctx
template <class T>
Ctx<T>::Ctx(const hot_config& c)
{
    cfg = c;
    dx = (T)c.dx;
    HOT_HIP(hipSetDevice(c.device));
    HOT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    prof.on = c.profile != 0;
    keep_debug = c.debug_store != 0;
    dscal.reserve(1024); // [0,256) solver scalars, [512, 1024) L-BFGS two-loop: dot batches and the Gram matrix (solve.hip)
    red_part.reserve(4096), red_count.reserve(4);
    HOT_HIP(hipMemset(red_count.p, 0, 4 * sizeof(unsigned)));
    HOT_HIP(hipHostMalloc((void**)&hscal, 256 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)); // fine-grained: kernels publish results and tickets here while they run (wait_ticket)
    std::memset(hscal, 0, 256 * sizeof(double)); // hscal[250] doubles as the device-written k_gs_sweep wait-timeout flag
    std::memset(&stats, 0, sizeof(stats));
}
template <class T>
Ctx<T>::~Ctx()
{
    for (auto* g : gmg) delete g;
    for (auto* l : levels) delete l;
    for (auto& pool : level_pool)
        for (auto* l : pool) delete l;
    if (hscal) (void)hipHostFree(hscal);
    if (stream) (void)hipStreamDestroy(stream);
}

template <class T>
int32_t Ctx<T>::exclusive_scan_i32(const int32_t* in, int32_t* out, size_t n)
{
    if (n == 0) return 0;
    size_t bytes = 0;
    HOT_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), stream));
    if (bytes > scan_tmp_bytes) {
        scan_tmp.reserve(bytes);
        scan_tmp_bytes = scan_tmp.cap;
    }
    HOT_HIP(rocprim::exclusive_scan(scan_tmp.p, bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), stream));
    int32_t last_in = 0, last_out = 0;
    HOT_HIP(hipMemcpyAsync(&last_in, in + n - 1, 4, hipMemcpyDeviceToHost, stream));
    HOT_HIP(hipMemcpyAsync(&last_out, out + n - 1, 4, hipMemcpyDeviceToHost, stream));
    sync();
    return last_in + last_out;
}

template <class T>
void Ctx<T>::set_particles(int64_t n, const void* X, const void* V, const void* m, const void* C, const void* F, const void* vol, const void* mu, const void* lam, const void* Jp)
{
    constexpr int index_bits = 32 - G::block_bits;
    HOT_CHECK(n > 0 && n < (1LL << index_bits), HOT_ERR_CAPACITY, "particle count must be in (0, 2^(32-block_bits)) (MpmSimulationBase.cpp:1071-1072)");
    need(X && V && m && vol && mu && lam, "X, V, mass, vol, mu, lambda are required");
    Np = n;
    reserve_particles(n);
    auto put = [&](DBuf<T>& dst, const void* src, int comps) {
        if (comps == 1) {
            HOT_HIP(hipMemcpyAsync(dst.p, src, n * sizeof(T), hipMemcpyDefault, stream));
            return;
        }
        HOT_HIP(hipMemcpyAsync(spare9.p, src, (size_t)n * comps * sizeof(T), hipMemcpyDefault, stream));
        HOT_LAUNCH(this, "aos_to_soa", k_aos_to_soa<T>, div_up(n * comps, 256), 256, 0, spare9.p, dst.p, n, comps);
    };
    put(pX, X, 3), put(pV, V, 3), put(pM, m, 1), put(pVol, vol, 1), put(pMu, mu, 1), put(pLam, lam, 1);
    if (C)
        put(pC, C, 9);
    else
        HOT_HIP(hipMemsetAsync(pC.p, 0, 9 * n * sizeof(T), stream));
    if (F)
        put(pF, F, 9);
    else
        HOT_LAUNCH(this, "identity9", k_identity9<T>, div_up(9 * n, 256), 256, 0, pF.p, n);
    if (Jp)
        put(pJp, Jp, 1);
    else
        HOT_LAUNCH(this, "fill", k_fill<T>, div_up(n, 256), 256, 0, pJp.p, n, (T)1);
    HOT_LAUNCH(this, "iota", k_iota, div_up(n, 256), 256, 0, slot2orig.p, n);
    HOT_LAUNCH(this, "iota", k_iota, div_up(n, 256), 256, 0, pGid.p, n); // global particle ids: the caller's order unless hot_set_particle_ids says otherwise
    Ng = Nb = Nn = 0;
    sync();
}

// every per-particle buffer for n particles (contents are not preserved on growth)
template <class T>
void Ctx<T>::reserve_particles(int64_t n)
{
    const double slack = sharded() ? 1.25 : 1.0; // shards grow and shrink as particles migrate
    // (+ 4: k_p2g_stream's 16-byte DMA lanes may straddle the end of an array's last component)
    pX.reserve(3 * n + 4, slack), pV.reserve(3 * n + 4, slack), pM.reserve(n + 4, slack), pC.reserve(9 * n + 4, slack), pF.reserve(9 * n, slack), pVol.reserve(n, slack), pMu.reserve(n, slack), pLam.reserve(n, slack),
        pJp.reserve(n, slack);
    pFn.reserve(9 * n, slack), pFt.reserve(9 * n, slack), pStress.reserve(9 * n, slack), pGradV.reserve(9 * n, slack);
    spare1.reserve(n + 4, slack), spare3.reserve(3 * n + 4, slack), spare9.reserve(9 * n + 4, slack), sparei.reserve(n, slack), slot2orig.reserve(n, slack), pGid.reserve(n, slack);
}
template <class T>
void Ctx<T>::set_particle_ids(const int32_t* ids)
{
    need(Np > 0 && ids, "hot_set_particle_ids after hot_set_particles");
    HOT_HIP(hipMemcpyAsync(pGid.p, ids, (size_t)Np * sizeof(int32_t), hipMemcpyDefault, stream));
    sync();
}
// ids of the particles in the order hot_get_particles returns them
template <class T>
void Ctx<T>::get_particle_ids(int32_t* ids)
{
    need(Np > 0 && ids, "no particles");
    HOT_LAUNCH(this, "soa_to_aos", k_soa_to_aos<int32_t>, div_up(Np, 256), 256, 0, pGid.p, sparei.p, slot2orig.p, Np, 1);
    HOT_HIP(hipMemcpyAsync(ids, sparei.p, (size_t)Np * sizeof(int32_t), hipMemcpyDefault, stream));
    sync();
}

template <class T>
void Ctx<T>::get_particles(void* X, void* V, void* C, void* F, void* mu, void* lam, void* Jp)
{
    need(Np > 0, "no particles");
    int64_t n = Np;
    auto get = [&](const DBuf<T>& src, void* dst, int comps) {
        if (!dst) return;
        HOT_LAUNCH(this, "soa_to_aos", k_soa_to_aos<T>, div_up(n * comps, 256), 256, 0, src.p, spare9.p, slot2orig.p, n, comps);
        HOT_HIP(hipMemcpyAsync(dst, spare9.p, (size_t)n * comps * sizeof(T), hipMemcpyDefault, stream));
        sync();
    };
    get(pX, X, 3), get(pV, V, 3), get(pC, C, 9), get(pF, F, 9), get(pMu, mu, 1), get(pLam, lam, 1), get(pJp, Jp, 1);
}

template <class T>
void Ctx<T>::sort()
{
    need(Np > 0, "hot_sort: no particles");
    double t0 = wall_ms();
    comm_calls = comm_calls_index = comm_bytes_index = comm_bytes_data = 0; // hot_stats.comm_*: since this call
    vmask = nullptr; // no row ownership until hot_p2g has numbered the nodes
    if (sharded()) migrate_particles(); // every particle to the rank that holds its SPGrid page range (changes Np)
    int64_t n = Np;
    keys.reserve(n), keys2.reserve(n), vals.reserve(n), vals2.reserve(n), flags.reserve(std::max<size_t>(n, 64)), scan.reserve(std::max<size_t>(n, 64));
    T one_over_dx = (T)1 / dx;
    // tie-break inside a cell: the caller's particle index — in a sharded run the global particle id, so that the order inside
    // a cell is the single-rank one whatever the shard looks like
    size_t bytes = 0;
    HOT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys.p, keys2.p, vals.p, vals2.p, (size_t)n, 0, 64, stream));
    if (bytes > sort_tmp_bytes) {
        sort_tmp.reserve(bytes);
        sort_tmp_bytes = sort_tmp.cap;
    }
    constexpr int index_bits = 32 - G::block_bits;
    if (sharded()) { // (cell, global id) order in two passes: ids of any size (see k_id_keys)
        HOT_LAUNCH(this, "make_keys", k_id_keys, div_up(n, 256), 256, 0, pGid.p, keys.p, vals.p, n);
        prof.begin("radix_sort_pairs", stream);
        HOT_HIP(rocprim::radix_sort_pairs(sort_tmp.p, bytes, keys.p, keys2.p, vals.p, vals2.p, (size_t)n, 0, 64, stream));
        prof.end(stream);
        HOT_LAUNCH(this, "make_keys", k_make_keys_in_order<T>, div_up(n, 256), 256, 0, pX.p, vals2.p, keys.p, vals.p, n, one_over_dx);
        prof.begin("radix_sort_pairs", stream);
        HOT_HIP(rocprim::radix_sort_pairs(sort_tmp.p, bytes, keys.p, keys2.p, vals.p, vals2.p, (size_t)n, 0, 64, stream));
        prof.end(stream);
    }
    else {
        HOT_LAUNCH(this, "make_keys", k_make_keys<T>, div_up(n, 256), 256, 0, pX.p, slot2orig.p, keys.p, vals.p, n, one_over_dx);
        prof.begin("radix_sort_pairs", stream);
        HOT_HIP(rocprim::radix_sort_pairs(sort_tmp.p, bytes, keys.p, keys2.p, vals.p, vals2.p, (size_t)n, 0, 64, stream));
        prof.end(stream);
    }
    // physical reorder of every per-particle array into sorted order
    auto reorder = [&](DBuf<T>& a, DBuf<T>& spare, int comps) {
        HOT_LAUNCH(this, "reorder_gather", k_gather<T>, div_up(n, 256), 256, 0, a.p, spare.p, vals2.p, n, comps);
        std::swap(a.p, spare.p);
        std::swap(a.cap, spare.cap);
    };
    reorder(pX, spare3, 3), reorder(pV, spare3, 3), reorder(pM, spare1, 1), reorder(pVol, spare1, 1), reorder(pMu, spare1, 1), reorder(pLam, spare1, 1), reorder(pJp, spare1, 1);
    reorder(pC, spare9, 9), reorder(pF, spare9, 9);
    HOT_LAUNCH(this, "reorder_gather", k_gather<int32_t>, div_up(n, 256), 256, 0, slot2orig.p, sparei.p, vals2.p, n, 1);
    std::swap(slot2orig.p, sparei.p);
    std::swap(slot2orig.cap, sparei.cap);
    HOT_LAUNCH(this, "reorder_gather", k_gather<int32_t>, div_up(n, 256), 256, 0, pGid.p, sparei.p, vals2.p, n, 1);
    std::swap(pGid.p, sparei.p);
    std::swap(pGid.cap, sparei.cap);
    // groups
    HOT_LAUNCH(this, "group_heads", k_group_heads, div_up(n, 256), 256, 0, keys2.p, flags.p, n);
    Ng = exclusive_scan_i32(flags.p, scan.p, n);
    group_first.reserve(Ng + 1), group_page.reserve(Ng), group_origin.reserve(3 * (size_t)Ng), group_nb.reserve(8 * (size_t)Ng);
    HOT_LAUNCH(this, "fill_groups", k_fill_groups<T>, div_up(n, 256), 256, 0, keys2.p, flags.p, scan.p, group_first.p, group_page.p, group_origin.p, n, Ng);
    // block list in Set_Page insertion order
    size_t cand = (size_t)Ng * 8;
    uint32_t capn = 1024;
    while (capn < 2 * cand) capn <<= 1;
    bh_keys.reserve(capn), bh_rank.reserve(capn), bh_id.reserve(capn);
    block_map.keys = bh_keys.p, block_map.minrank = bh_rank.p, block_map.id = bh_id.p, block_map.mask = capn - 1;
    flags.reserve(cand), scan.reserve(cand);
    HOT_LAUNCH(this, "hash_clear", k_hash_clear, div_up(capn, 256), 256, 0, block_map);
    HOT_LAUNCH(this, "block_insert", k_block_insert<T>, div_up(cand, 256), 256, 0, block_map, group_origin.p, Ng);
    HOT_LAUNCH(this, "block_flag", k_block_flag<T>, div_up(cand, 256), 256, 0, block_map, group_origin.p, flags.p, Ng);
    Nb = exclusive_scan_i32(flags.p, scan.p, cand);
    blocks.reserve(Nb);
    HOT_LAUNCH(this, "block_assign", k_block_assign<T>, div_up(cand, 256), 256, 0, block_map, group_origin.p, flags.p, scan.p, blocks.p, Ng);
    if (sharded()) {
        IndexPhase ip(this);
        merge_block_lists(); // the global Set_Page order: block ids, Nb and block_map are global from here on
    }
    HOT_LAUNCH(this, "group_nb", k_group_nb<T>, div_up(cand, 256), 256, 0, block_map, group_origin.p, group_nb.p, Ng);
    if (halo_mode()) build_tile_plan(); // which ranks cover which block: the pairwise tile exchanges of the step
    // node tiles
    size_t slots = (size_t)Nb * EPB;
    gM.reserve(slots, 1.25), gMV.reserve(3 * slots, 1.25), gF.reserve(3 * slots, 1.25), gCN.reserve(slots, 1.25), gIdx.reserve(slots, 1.25), block_count.reserve(Nb + 1, 1.25);
    block_group.reserve(Nb, 1.25), block_rev.reserve(8 * (size_t)Nb, 1.25), gPart.reserve((size_t)Ng * 5 * TILE, 1.25);
    HOT_HIP(hipMemsetAsync(block_group.p, 0xff, (size_t)Nb * sizeof(int32_t), stream));
    HOT_LAUNCH(this, "block_group", k_block_group, div_up(Ng, 256), 256, 0, group_nb.p, block_group.p, Ng);
    HOT_LAUNCH(this, "block_rev", k_block_rev<T>, div_up((size_t)Nb * 8, 256), 256, 0, block_map, blocks.p, block_group.p, block_rev.p, Nb);
    build_cell_table();
    Nn = 0;
    Nc = 0;
    updated = false;
    stats.ms_sort = wall_ms() - t0;
}

template <class T>
void Ctx<T>::reduce_tiles(int Q, T* o0, T* o1, T* o2, T* o3, T* o4, const char* name)
{
    HOT_LAUNCH(this, name, k_tile_reduce<T>, div_up((size_t)Nb * EPB, 256), 256, 0, gPart.p, Q, block_rev.p, o0, o1, o2, o3, o4, Nb);
}

template <class T>
void Ctx<T>::get_counts(int64_t* np, int32_t* ng, int32_t* nb, int32_t* nn)
{
    if (np) *np = Np;
    if (ng) *ng = Ng;
    if (nb) *nb = Nb;
    if (nn) *nn = Nn;
}

template <class T>
void Ctx<T>::get_indexing(int32_t* order, uint64_t* base_offset, int32_t* group, uint64_t* block_offset, uint64_t* blk)
{
    need(Ng > 0, "hot_get_indexing before hot_sort");
    int64_t n = Np;
    if (order || base_offset) {
        DBuf<uint64_t> off;
        DBuf<int32_t> ord;
        off.reserve(n), ord.reserve(n);
        HOT_LAUNCH(this, "base_offsets", k_base_offsets<T>, div_up(n, 256), 256, 0, pX.p, slot2orig.p, off.p, ord.p, n, (T)1 / dx);
        download(order, ord.p, n);
        download(base_offset, off.p, n);
        sync();
    }
    if (group) {
        DBuf<int32_t> r;
        r.reserve(2 * (size_t)Ng);
        HOT_LAUNCH(this, "group_ranges", k_group_ranges, div_up(Ng, 256), 256, 0, group_first.p, r.p, Ng);
        download(group, r.p, 2 * (size_t)Ng);
        sync();
    }
    download(block_offset, group_page.p, Ng);
    download(blk, blocks.p, Nb);
    sync();
}

template struct Ctx<float>;
template struct Ctx<double>;

CtxBase* make_ctx_f32(const hot_config& cfg) { return new Ctx<float>(cfg); }
CtxBase* make_ctx_f64(const hot_config& cfg) { return new Ctx<double>(cfg); }

} // namespace hot
