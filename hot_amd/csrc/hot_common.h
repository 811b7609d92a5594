// libhotmi355x — common device/host utilities (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <map>
#include <chrono>
#include "../../include/hot_mi355x.h"

namespace hot {

struct Error {
    int code;
    std::string msg;
};

#define HOT_HIP(expr)                                                                                             \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) throw ::hot::Error{ HOT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e) }; \
    } while (0)

#define HOT_CHECK(cond, code, text)                       \
    do {                                                  \
        if (!(cond)) throw ::hot::Error{ (code), (text) }; \
    } while (0)

// ------------------------------------------------------------------ device buffers (grow-only)
template <class T>
struct DBuf {
    T* p = nullptr;
    size_t cap = 0;
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf()
    {
        if (p) (void)hipFree(p);
    }
    // ensure capacity >= n elements; contents are NOT preserved on growth
    void reserve(size_t n, double slack = 1.0)
    {
        if (n <= cap) return;
        if (p) HOT_HIP(hipFree(p));
        p = nullptr;
        cap = (size_t)(n * slack) + 64;
        HOT_HIP(hipMalloc((void**)&p, cap * sizeof(T)));
    }
    operator T*() const { return p; }
};

// ------------------------------------------------------------------ SPGrid address arithmetic
// Device restatement of SPGrid_Mask<log2_struct, log2_struct, 3, 12> (reference Lib/SPGrid/Core/SPGrid_Mask.h:21-57,
// 76-80,119-123,150-157,237-245): low 12 bits = element inside the 4 KiB page (x | y | z, z lowest, times the
// struct size), high bits = block coordinates Morton-interleaved starting at the "left-over" axis.  Instead of
// the reference's generic software pdep (SPGrid_Utilities.h:80-343) the two concrete layouts are written in
// closed form with magic-number bit spreads; tests pin them bit-exactly to the compiled reference.
template <int LOG2_STRUCT>
struct SpMask {
    static constexpr int data_bits = LOG2_STRUCT;
    static constexpr int block_bits = 12 - LOG2_STRUCT;
    static constexpr int zb = block_bits / 3 + (block_bits % 3 > 0);
    static constexpr int yb = block_bits / 3 + (block_bits % 3 > 1);
    static constexpr int xb = block_bits / 3;
    static constexpr int EPB = 1 << block_bits; // elements (nodes) per block
    static constexpr int BX = 1 << xb, BY = 1 << yb, BZ = 1 << zb;
    // position of the lowest page bit of each axis: page masks are (0x9249.. << s) with s = 3 - block_bits % 3,
    // i.e. z at bits {s, s+3, ..} >= 12 etc.
    static constexpr int sh = 3 - block_bits % 3;
    // first bit index >= 12 for each axis
    static constexpr int first_at_or_above_12(int start)
    {
        int b = start;
        while (b < 12) b += 3;
        return b;
    }
    static constexpr int zlo = first_at_or_above_12(sh + 0), ylo = first_at_or_above_12(sh + 1), xlo = first_at_or_above_12(sh + 2);

    // spread the low 21 bits of v so that bit i lands at bit 3*i
    __host__ __device__ static inline uint64_t spread3(uint64_t v)
    {
        v &= 0x1fffffULL;
        v = (v | (v << 32)) & 0x1f00000000ffffULL;
        v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
        v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
        v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
        v = (v | (v << 2)) & 0x1249249249249249ULL;
        return v;
    }
    __host__ __device__ static inline uint32_t compact3(uint64_t v)
    {
        v &= 0x1249249249249249ULL;
        v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ULL;
        v = (v ^ (v >> 4)) & 0x100f00f00f00f00fULL;
        v = (v ^ (v >> 8)) & 0x1f0000ff0000ffULL;
        v = (v ^ (v >> 16)) & 0x1f00000000ffffULL;
        v = (v ^ (v >> 32)) & 0x1fffffULL;
        return (uint32_t)v;
    }
    __host__ __device__ static inline uint64_t linear_offset(int i, int j, int k)
    {
        uint64_t ux = (uint64_t)(int64_t)i, uy = (uint64_t)(int64_t)j, uz = (uint64_t)(int64_t)k;
        uint64_t elem = ((ux & (BX - 1)) << (data_bits + zb + yb)) | ((uy & (BY - 1)) << (data_bits + zb)) | ((uz & (BZ - 1)) << data_bits);
        uint64_t page = (spread3(ux >> xb) << xlo) | (spread3(uy >> yb) << ylo) | (spread3(uz >> zb) << zlo);
        return page | elem;
    }
    __host__ __device__ static inline void linear_to_coord(uint64_t o, int& i, int& j, int& k)
    {
        int ex = (int)((o >> (data_bits + zb + yb)) & (BX - 1)), ey = (int)((o >> (data_bits + zb)) & (BY - 1)), ez = (int)((o >> data_bits) & (BZ - 1));
        i = (int)(compact3(o >> xlo) << xb) | ex;
        j = (int)(compact3(o >> ylo) << yb) | ey;
        k = (int)(compact3(o >> zlo) << zb) | ez;
    }
};

// per-dtype grid geometry: GridState<float,3> = 64 B, GridState<double,3> = 128 B (reference Lib/MPM/MpmGrid.h:15-34)
template <class T>
struct Geo;
template <>
struct Geo<float> : SpMask<6> {
};
template <>
struct Geo<double> : SpMask<7> {
};

// reference Lib/Ziran/Math/MathTools.h:21-25 and Lib/Ziran/Math/Splines/BSplines.h:16-20
template <class T>
__host__ __device__ inline int int_floor(T x)
{
    int i = (int)x;
    return i - (i > x);
}
template <class T>
__host__ __device__ inline int base_node(T x_index_space) { return int_floor<T>(x_index_space - (T)0.5); }

// The index-space coordinate X / dx: the reference stores the rounded product one_over_dx * X, takes floor(. - 0.5) for the base node and
// subtracts the base node from it (BSplineWeights::compute, BSplines.h:16-29, MpmGrid.h:55-78).  Built as the reference is (-O3
// -march=native, CMakeLists.txt:28; GCC's default -ffp-contract=fast) both the subtraction of 0.5 and the subtraction of the base node are
// contracted with the multiply, i.e. both see the EXACT product — which is also 40x closer to the truth in fp32 (measured against an fp64
// run of the same inputs: 6.5e-7 against 2.6e-5 in the trial F; at X / dx ~ 500 a rounded float product has lost 3e-5 of a cell).  Whether
// a compiler fuses here is its choice, so both roundings are spelled out as fma on the device (sort keys, transfers, state pass; the CPU
// restatement the tests compare with does the same): base = floor(fma(1/dx, X, -0.5)), fraction = fma(1/dx, X, -base).
template <class T>
__host__ __device__ inline int base_node_of(T one_over_dx, T x) { return int_floor<T>(fma(one_over_dx, x, -(T)0.5)); }

// quadratic B-spline weights and derivatives of one axis: reference BSplines.h:55-81
template <class T>
__device__ inline void bspline(T one_over_dx, T xw, int& base, T (&w)[3], T (&dw)[3])
{
    base = base_node_of<T>(one_over_dx, xw);
    T d0 = fma(one_over_dx, xw, -(T)base);
    T z = ((T)1.5 - d0);
    w[0] = (T)0.5 * z * z;
    T d1 = d0 - (T)1;
    w[1] = (T)0.75 - d1 * d1;
    T d2 = (T)1 - d1;
    T zz = (T)1.5 - d2;
    w[2] = (T)0.5 * zz * zz;
    dw[0] = -z;
    dw[1] = -(T)2 * d1;
    dw[2] = zz;
}

// ------------------------------------------------------------------ wave / block reductions (wave = 64)
// Sum over the 64 lanes, returned in every lane.  DPP data movement (quad_perm, row_shr, row_bcast) instead of
// ds_bpermute shuffles: no LDS traffic and a dependent latency of a few VALU ops per step.  The order of the additions
// is fixed, so the result is deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float wave_last(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ double wave_last(double v)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <class T>
__device__ inline T wave_sum(T v)
{
    v += dpp_move<0xb1, 0xf>(v); // quad_perm [1,0,3,2]
    v += dpp_move<0x4e, 0xf>(v); // quad_perm [2,3,0,1]: every lane holds its quad's sum
    v += dpp_move<0x114, 0xf>(v); // row_shr:4  (lanes without a source add 0)
    v += dpp_move<0x118, 0xf>(v); // row_shr:8  : lane 15 of every row holds the row's sum
    v += dpp_move<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
    return wave_last(v);
}
// block-wide sum for blockDim.x == 256; result valid in thread 0
template <class T>
__device__ inline T block_sum_256(T v, T* sm4)
{
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm4[w] = v;
    __syncthreads();
    T r = 0;
    if (threadIdx.x == 0) r = sm4[0] + sm4[1] + sm4[2] + sm4[3];
    __syncthreads();
    return r;
}

// NV block-wide sums at once (blockDim.x == 256): the same additions in the same order as NV calls of block_sum_256, with one barrier
// pair instead of NV; results valid in thread 0.  use(k): whether sum k is wanted (workgroup-uniform).  smn: NV * 4 doubles of LDS.
template <int NV, class Use>
__device__ inline void block_sum_256_n(double (&v)[NV], Use use, double* smn)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (use(k)) {
            v[k] = wave_sum(v[k]);
            if (lane == 0) smn[4 * k + w] = v[k];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (use(k)) v[k] = smn[4 * k] + smn[4 * k + 1] + smn[4 * k + 2] + smn[4 * k + 3];
    }
    __syncthreads();
}

// Order-deterministic grid-wide sum (blockDim.x == 256), one launch: every workgroup deposits its total, the one that
// arrives last adds the deposits in index order and STORES the result (no same-address floating-point atomics, whose
// arrival order changes the rounding from run to run and, between ranks of a sharded solve, from rank to rank).
// Deposits and the arrival counter are RELAXED agent-scope atomics: on gfx950 those are performed at the device-coherent level
// (sc1 write-through stores / L2-bypassing loads), past the per-XCD L2s, so no cache write-back or invalidate is needed; the
// deposit is ordered before the counter increment by s_waitcnt vmcnt(0) (the store has been acknowledged by then).  A
// release / acquire pair would be the portable spelling, but at agent scope it compiles to buffer_wbl2 + buffer_inv, which
// flush and invalidate the XCD's whole L2 once per workgroup (measured: k_state 0.25 -> 0.65 ms at 37 k workgroups).
// The same hardware-level hand-off is used by k_gs_sweep (mg_solve.hip).  The counter is left at 0 for the next launch on
// the stream.  t0 / t1: block totals, valid in thread 0 (block_sum_256).
struct GridRed {
    double* part; // >= 2 * gridDim.x
    unsigned* count;
    double* mirror; // optional: pinned host memory that also receives the result(s), so that the host needs no copy
    double* ticket; // optional (with mirror): pinned host word that receives ticket_val after the results — the host spins on it
    double ticket_val; //   instead of paying a hipStreamSynchronize wake-up (Ctx::wait_ticket)
    double* mirror1 = nullptr; // optional: where the SECOND result of a two-sum launch goes instead of mirror[1]
};
// results first, then the ticket.  A system-scope release is needed here: s_waitcnt vmcnt(0) only waits for the L2's acknowledgement
// of the result stores, and the ticket (another cache line, another channel) can overtake them on the way to host memory.
__device__ __forceinline__ void host_ticket_store(double* ticket, double val)
{
    __hip_atomic_store(ticket, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void grid_sum_store(double t0, double t1, int nv, GridRed gr, double* o0, double* o1, double* sm4)
{
    __shared__ int s_last;
    const unsigned nb = gridDim.x;
    if (threadIdx.x == 0) {
        __hip_atomic_store(gr.part + blockIdx.x, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nv > 1) __hip_atomic_store(gr.part + nb + blockIdx.x, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef HOT_GRIDSUM_DEPOSIT_ONLY // (timing experiment, tools/variant.sh: what the acknowledged deposit + the counter's round trip cost a workgroup; no result)
        s_last = blockIdx.x == nb - 1u;
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(gr.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == nb - 1u;
#endif
    }
    __syncthreads();
    if (!s_last) return; // workgroup-uniform
    // The deposits come from past the L2 (a few microseconds per dependent round): eight rounds of loads are in flight per thread before the
    // first addition; the additions keep the order i = thread, thread + 256, ...
    double a = 0, b = 0;
    for (unsigned base = threadIdx.x; base < nb; base += 8 * 256) {
        double va[8], vb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned i = base + 256u * j;
            va[j] = i < nb ? __hip_atomic_load(gr.part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            vb[j] = (nv > 1 && i < nb) ? __hip_atomic_load(gr.part + nb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) a += va[j], b += vb[j];
    }
    a = block_sum_256<double>(a, sm4);
    if (nv > 1) b = block_sum_256<double>(b, sm4);
    if (threadIdx.x == 0) {
        *o0 = a;
        if (nv > 1) *o1 = b;
        if (gr.mirror) {
            gr.mirror[0] = a;
            if (nv > 1) (gr.mirror1 ? *gr.mirror1 : gr.mirror[1]) = b;
            if (gr.ticket) host_ticket_store(gr.ticket, gr.ticket_val);
        }
        __hip_atomic_store(gr.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The same for up to NVMAX sums per launch (the L-BFGS dot batches): `tot` holds the block totals in thread 0, slot(k) is the deposit
// row of total k (-1: not used; the used ones fill rows 0 .. nv - 1), deposits are laid out [row][workgroup].  gr.part must hold
// nv * gridDim.x doubles (Ctx::gred_n), smn NVMAX * 4 doubles of LDS.  The last workgroup adds all nv rows in one pass (loads of every
// row in flight together, one barrier pair).
template <int NVMAX, class Slot>
__device__ inline void grid_sum_store_n(const double (&tot)[NVMAX], Slot slot, int nv, GridRed gr, double* out, double* smn)
{
    __shared__ int s_last_n;
    const unsigned nb = gridDim.x;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NVMAX; ++k) {
            const int row = slot(k);
            if (row >= 0) __hip_atomic_store(gr.part + (size_t)row * nb + blockIdx.x, tot[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(gr.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last_n = prev == nb - 1u;
    }
    __syncthreads();
    if (!s_last_n) return; // workgroup-uniform
    double a[NVMAX];
#pragma unroll
    for (int k = 0; k < NVMAX; ++k) a[k] = 0;
    constexpr int J = NVMAX > 8 ? 2 : 4; // rounds of loads in flight per thread (see grid_sum_store); same order of additions as one round at a time
    for (unsigned base = threadIdx.x; base < nb; base += J * 256) {
        double v[NVMAX][J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const unsigned i = base + 256u * j;
#pragma unroll
            for (int k = 0; k < NVMAX; ++k)
                v[k][j] = (k < nv && i < nb) ? __hip_atomic_load(gr.part + (size_t)k * nb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int k = 0; k < NVMAX; ++k) a[k] += v[k][j];
    }
    block_sum_256_n<NVMAX>(a, [nv](int k) { return k < nv; }, smn);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NVMAX; ++k)
            if (k < nv) {
                out[k] = a[k];
                if (gr.mirror) gr.mirror[k] = a[k];
            }
        if (gr.mirror && gr.ticket) host_ticket_store(gr.ticket, gr.ticket_val);
        __hip_atomic_store(gr.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// broadcast lane `src` (wave-uniform) of v to every lane through SGPRs (v_readlane_b32), no LDS round trip
__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <class T>
__device__ inline void atomic_add(T* p, T v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// accumulator type of the LDS tiles: double also for the fp32 build (see k_force_cells)
template <class T>
using AccT = double;

template <class T>
__device__ inline void lds_atomic_add(T* p, T v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ------------------------------------------------------------------ open-addressing hash map u64 -> (min rank, id)
struct HashMap {
    uint64_t* keys = nullptr; // ~0 == empty
    unsigned long long* minrank = nullptr;
    int32_t* id = nullptr;
    uint32_t mask = 0; // capacity - 1
};
__host__ __device__ inline uint64_t hash_mix(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
__device__ inline uint32_t hash_insert_min(const HashMap& h, uint64_t key, unsigned long long rank)
{
    uint32_t s = (uint32_t)hash_mix(key) & h.mask;
    while (true) {
        unsigned long long prev = atomicCAS((unsigned long long*)&h.keys[s], ~0ULL, (unsigned long long)key);
        if (prev == ~0ULL || prev == key) {
            atomicMin(&h.minrank[s], rank);
            return s;
        }
        s = (s + 1) & h.mask;
    }
}
__device__ inline int32_t hash_find_slot(const HashMap& h, uint64_t key)
{
    uint32_t s = (uint32_t)hash_mix(key) & h.mask;
    while (true) {
        uint64_t k = h.keys[s];
        if (k == key) return (int32_t)s;
        if (k == ~0ULL) return -1;
        s = (s + 1) & h.mask;
    }
}
__device__ inline int32_t hash_find_id(const HashMap& h, uint64_t key)
{
    int32_t s = hash_find_slot(h, key);
    return s < 0 ? -1 : h.id[s];
}
// pack non-negative 3-D integer coordinates (< 2^21) into a map key
__host__ __device__ inline uint64_t coord_key(int x, int y, int z) { return ((uint64_t)(uint32_t)x << 42) | ((uint64_t)(uint32_t)y << 21) | (uint64_t)(uint32_t)z; }

// First-generation kernels and launch-structure alternatives are compiled only into the A/B build (-DHOT_AB_KERNELS,
// libhotmi355x_ab.so), where environment variables select them for tests/test_gpu_variants.py.  The product library has
// neither the kernels nor any getenv.
#ifdef HOT_AB_KERNELS
inline bool ab_flag(const char* name) { return getenv(name) != nullptr; }
inline int ab_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr bool ab_flag(const char*) { return false; }
constexpr int ab_int(const char*, int dflt) { return dflt; }
#endif

inline double wall_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace hot
