// libhotmi355x — operator applications: block-ELL SpMV, restriction / prolongation, smoothers, V-cycle.
//
//   k_spmv          SquareMatrix::multiply (reference Projects/multigrid/SquareMatrix.h:477-487) — THE bandwidth consumer
//                   (SURVEY §8a row 17/21).  One wavefront per block row: a row is 125 contiguous 3x3 blocks (9000 B fp64),
//                   lane l owns slots l and l+64, so the 64 lanes stream the row in two fully coalesced sweeps; x is
//                   gathered per slot, the three row sums are reduced with __shfl_xor.
//   k_apmv_sub      the r -= A (P e) of the V-cycle as r -= (A P) e with the A P kept from the Galerkin build (half the bytes).
//   k_gs_block      MultigridOperator::gs_smooth (Projects/multigrid/MultigridPreconditioner.h:266-318): symmetric coloured
//                   block Gauss–Seidel in the reference's exact node order (colour, first-touch block, id), one launch per
//                   colour, its sub-blocks walked inside: streaming phase + LDS-triangle substitution (see the kernel).
//   k_gs_sweep      the same passes chained inside one launch by per-block flags (coarse levels).   k_gs_color: the simple one-wavefront-
//                   per-block version kept as the A/B reference (HOT_SIMPLE_GS).
//   restrict/prolong SparseMPMMatrix::transposeMultiply / multiply on the transfer matrices (MPMMultigridMatrix.h:63-70) as
//                   pure gathers over the child / parent tables with scalar weights (the reference stores 3x3 w*I blocks).
//   smooth_dev      jacobi_smooth :160-173, optimal_jacobi_smooth :174-189, cg_smooth :190-226, chebyshev_smooth :227-264
//                   (+ SquareMatrix::estimate2norm), gs_smooth :266-318
//   vcycle_dev      MultigridOperator::operator() :362-421 with setup_parameters :525-551
#include "hot_impl.h"
#include "hot_svd.h"

namespace hot {

// ------------------------------------------------------------------------------------------------ vector ops
template <class T>
__global__ void k_axpy(size_t n, T a, const T* __restrict__ x, T* y)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
template <class T>
__global__ void k_axpy_dev(size_t n, const double* a, double sign, const T* __restrict__ x, T* y)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    T s = (T)(sign * (*a));
    if (i < n) y[i] += s * x[i];
}
// y = x + (*a) * y
template <class T>
__global__ void k_xpay_dev(size_t n, const double* a, const T* __restrict__ x, T* y)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    T s = (T)(*a);
    if (i < n) y[i] = x[i] + s * y[i];
}
template <class T>
__global__ __launch_bounds__(256) void k_dot(size_t n, const T* __restrict__ x, const T* __restrict__ y, double* out, GridRed gr, const uint8_t* __restrict__ mask)
{
    __shared__ double red[4];
    double s = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * stride) { // four strided elements per trip in flight
        T xv[4], yv[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride, ic = i < n ? i : i0;
            on[u] = i < n && (!mask || mask[ic / 3]); // sharded, halo mode: the rows this rank owns
            xv[u] = on[u] ? x[ic] : (T)0, yv[u] = on[u] ? y[ic] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (on[u]) s += (double)(xv[u] * yv[u]);
    }
    double t = block_sum_256<double>(s, red);
    grid_sum_store(t, 0.0, 1, gr, out, nullptr, red);
}
template <class T>
void Ctx<T>::axpy(size_t n, T a, const T* x, T* y)
{
    HOT_LAUNCH(this, "axpy", k_axpy<T>, div_up(n, 256), 256, 0, n, a, x, y);
}
template <class T>
void Ctx<T>::axpy_dev(size_t n, const double* a, double sign, const T* x, T* y)
{
    HOT_LAUNCH(this, "axpy", k_axpy_dev<T>, div_up(n, 256), 256, 0, n, a, sign, x, y);
}
// a plain kernel: a device-to-device hipMemcpyAsync between two kernels leaves the GPU idle for ~20 us on either side of the blit
template <class T>
__global__ __launch_bounds__(256) void k_copy(size_t n, const T* __restrict__ x, T* __restrict__ y)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i];
}
// y := x, y2 := x (if not null), z := 0: the head of a V-cycle
template <class T>
__global__ __launch_bounds__(256) void k_vcycle_start(size_t n, const T* __restrict__ x, T* __restrict__ y, T* __restrict__ y2, T* __restrict__ z)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = x[i];
        y[i] = v, z[i] = (T)0;
        if (y2) y2[i] = v;
    }
}
template <class T>
void Ctx<T>::copy(size_t n, const T* x, T* y)
{
    if (n) HOT_LAUNCH(this, "copy", k_copy<T>, (int)std::min<size_t>(div_up(n, 256), 2048), 256, 0, n, x, y);
}
template <class T>
void Ctx<T>::zero(size_t n, T* y)
{
    HOT_HIP(hipMemsetAsync(y, 0, n * sizeof(T), stream));
}
template <class T>
void Ctx<T>::dot_to(size_t n, const T* x, const T* y, double* out, double* mirror)
{
    const int grid = std::min(div_up(n, 1024), 256);
    if (vmask) { // partitioned vectors: local sum over the owned rows, summed over the ranks, then (if asked for) handed to the host slot
        HOT_LAUNCH(this, "dot", k_dot<T>, grid, 256, 0, n, x, y, out, gred(grid), vmask);
        reduce_scalars(out, 1);
        if (mirror) HOT_HIP(hipMemcpyAsync(mirror, out, sizeof(double), hipMemcpyDeviceToHost, stream));
        return;
    }
    HOT_LAUNCH(this, "dot", k_dot<T>, grid, 256, 0, n, x, y, out, gred(grid, mirror), (const uint8_t*)nullptr); // <= 256 deposits, summed in index order
}
template <class T>
double Ctx<T>::dot_host(size_t n, const T* x, const T* y)
{
    const int grid = std::min(div_up(n, 1024), 256);
    HOT_LAUNCH(this, "dot", k_dot<T>, grid, 256, 0, n, x, y, dscal.p + 100, gred(grid, hscal + 100, true), vmask); // the summing workgroup also writes the pinned host slot
    wait_ticket();
    if (vmask) c_allreduce(hscal + 100, 1, HOT_COMM_F64, HOT_COMM_SUM, false); // partitioned vectors: the ranks' sums over their own rows
    return hscal[100];
}

// ------------------------------------------------------------------------------------------------ SpMV
// Row-per-wavefront kernels deal their workgroups to the XCDs in eight contiguous runs of rows (workgroup b runs on XCD b % 8: observed
// placement, used for speed only — any placement computes the same rows): DOF ids follow the blocks in first-touch (page) order, so an
// XCD's rows gather x from one region of the grid and that part of x stays in ITS 4 MB L2, instead of every XCD pulling all of x
// through its own L2 (k_gs_offblock: 46 MB of 155 MB per launch were those eight copies, profiles/r04_pmc_summary.json).  Launch with
// xcd_grid(number of workgroups).
__device__ __forceinline__ int xcd_block() { return (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)); }
static inline int xcd_grid(int nwg) { return 8 * div_up(nwg, 8); }
template <class T>
__global__ __launch_bounds__(256) void k_spmv(const int32_t* __restrict__ col, const T* __restrict__ val, const T* __restrict__ x, T* __restrict__ y, int n, const uint8_t* __restrict__ own)
{
    const int lane = threadIdx.x & 63;
    const int row = xcd_block() * 4 + (threadIdx.x >> 6);
    if (row >= n || (own && !own[row])) return; // sharded: rows of other ranks (wave-uniform)
    const int32_t* c = col + (int64_t)row * 125;
    const T* v = val + (int64_t)row * 1125;
    T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int k = lane + 64 * r;
        if (k < 125) {
            int j = c[k];
            const T* b = v + k * 9;
            T x0 = x[3 * (int64_t)j], x1 = x[3 * (int64_t)j + 1], x2 = x[3 * (int64_t)j + 2];
            s0 += b[0] * x0 + b[3] * x1 + b[6] * x2;
            s1 += b[1] * x0 + b[4] * x1 + b[7] * x2;
            s2 += b[2] * x0 + b[5] * x1 + b[8] * x2;
        }
    }
    s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
    if (lane == 0) y[3 * (int64_t)row] = s0, y[3 * (int64_t)row + 1] = s1, y[3 * (int64_t)row + 2] = s2;
}
// A load of data a kernel reads ONCE (matrix values, column ids, images).  An ORDINARY load: the non-temporal hint, which helps a 16-byte-per-lane copy
// (tools/micro/copy_bw.hip: 6.2 -> 6.5 TB/s), makes these 8-byte pieces — several lanes and instructions per cache line — miss on every piece: measured
// with __builtin_nontemporal_load here, C2: the colour pass 46 -> 77 us per launch, k_gs_residual 260 -> 327 us, k_apmv_sub 183 -> 329 us.
template <class U>
__device__ __forceinline__ U nt_load(const U* p)
{
    return *p;
}
// "not written yet" marks of the chained GS sweeps (k_gs_sweep): signalling-NaN payloads that no arithmetic result carries
template <class T>
struct GsUnset;
template <>
struct GsUnset<double> {
    static constexpr unsigned long long bits = 0x7ff4dead0badf00dull;
    static __device__ __forceinline__ bool is(double v) { return (unsigned long long)__double_as_longlong(v) == bits; }
};
template <>
struct GsUnset<float> {
    static constexpr unsigned bits = 0x7fa0f00du;
    static __device__ __forceinline__ bool is(float v) { return (unsigned)__float_as_int(v) == bits; }
};
template <class T>
__device__ __forceinline__ void gs_store_unset(T* p)
{
    if constexpr (sizeof(T) == 8)
        *(unsigned long long*)p = GsUnset<double>::bits;
    else
        *(unsigned*)p = GsUnset<float>::bits;
}
// r_i -= sum_J AP[i][J] e_J : the residual update after the coarse-grid correction.  A (P e) and (A P) e are the same
// vector; A P is a by-product of the Galerkin build with 64 window slots per row instead of the 125 of A, i.e. half
// the bytes of the SpMV the reference performs here (MultigridPreconditioner.h:362-421).
template <class T>
__global__ __launch_bounds__(256) void k_apmv_sub(const int32_t* __restrict__ apc, const T* __restrict__ apv, const T* __restrict__ e, T* __restrict__ r, int n, const uint8_t* __restrict__ own,
    T* unset /*not null: the level's GS forward target, marked "not written yet" here (see smooth_dev)*/)
{
    const int lane = threadIdx.x & 63;
    const int row = xcd_block() * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (unset && lane < 3) gs_store_unset(unset + 3 * (int64_t)row + lane);
    if (own && !own[row]) return;
    // lane = geometric window position; -1 where the window is structurally zero (an even coordinate's fourth coarse node): nothing is stored nor loaded
    // there, and the stored positions are packed in this order (k_ap): a lane's slot is its rank among the row's stored positions.  The wavefront sum
    // pairs the same positions as before the packing (the skipped ones used to add exact zeros): bit-identical results.
    const int j = nt_load(apc + (int64_t)row * 64 + lane);
    const int slot = __popcll(__ballot(j >= 0) & ((1ull << lane) - 1ull));
    T s0 = (T)0, s1 = (T)0, s2 = (T)0;
    if (j >= 0) {
        const T* bp = apv + ((int64_t)row * 64 + slot) * 9;
        T b[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) b[t] = nt_load(bp + t);
        const T x0 = e[3 * (int64_t)j], x1 = e[3 * (int64_t)j + 1], x2 = e[3 * (int64_t)j + 2];
        s0 = b[0] * x0 + b[3] * x1 + b[6] * x2;
        s1 = b[1] * x0 + b[4] * x1 + b[7] * x2;
        s2 = b[2] * x0 + b[5] * x1 + b[8] * x2;
    }
    s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
    if (lane == 0) r[3 * (int64_t)row] -= s0, r[3 * (int64_t)row + 1] -= s1, r[3 * (int64_t)row + 2] -= s2;
}
// ---- cg_smooth (MultigridPreconditioner.h:190-226) in three launches per iteration instead of nine.  Device scalars:
// s[0] z'r of the current iterate, s[1] du'A du, s[4] z'r of the next one, s[6] "s[4] is to become s[0]", s[7] z'r of the initial
// residual, s[8] the tolerance 0.25 s[7] (cgratio 0.5 squared, :203-209), s[9] iterations done.  The loop test `z'r < tol -> stop` is
// evaluated on the device: an iteration launched after convergence does nothing, so the host can enqueue a group of iterations and
// look at the outcome once (k_cg_direction leaves the latest z'r and the count in the pinned host slots `hm`).
__device__ __forceinline__ bool cg_active(double zTr, double tol) { return !(zTr < tol); }
template <class T>
__global__ void k_cg_setup(double* s, double* hm)
{
    const double tol = (double)(T)(s[7] * 0.25);
    s[8] = tol;
    s[1] = s[2] = s[3] = s[4] = s[5] = s[6] = s[9] = 0.0;
    hm[0] = s[0], hm[1] = 0.0, hm[2] = tol;
}
template <class T>
__global__ __launch_bounds__(256) void k_cg_spmv_dot(const int32_t* __restrict__ col, const T* __restrict__ val, const T* __restrict__ du, T* __restrict__ dAu, int n, double* s, GridRed gr)
{
    __shared__ double red[4];
    const bool roll = s[6] != 0.0;
    const double cur = roll ? s[4] : s[0]; // s[4] is not written in this launch; s[0] is read only where it is not
    if (blockIdx.x == 0 && threadIdx.x == 0 && roll) s[0] = s[4]; // also when converged: the two kernels that follow test s[0]
    if (!cg_active(cur, s[8])) return; // converged before this iteration
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    double part = 0;
    if (row < n) {
        const int32_t* c = col + (int64_t)row * 125;
        const T* v = val + (int64_t)row * 1125;
        T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int k = lane + 64 * r;
            if (k < 125) {
                int j = c[k];
                const T* b = v + k * 9;
                T x0 = du[3 * (int64_t)j], x1 = du[3 * (int64_t)j + 1], x2 = du[3 * (int64_t)j + 2];
                s0 += b[0] * x0 + b[3] * x1 + b[6] * x2;
                s1 += b[1] * x0 + b[4] * x1 + b[7] * x2;
                s2 += b[2] * x0 + b[5] * x1 + b[8] * x2;
            }
        }
        s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
        if (lane == 0) {
            dAu[3 * (int64_t)row] = s0, dAu[3 * (int64_t)row + 1] = s1, dAu[3 * (int64_t)row + 2] = s2;
            part = (double)(s0 * du[3 * (int64_t)row]) + (double)(s1 * du[3 * (int64_t)row + 1]) + (double)(s2 * du[3 * (int64_t)row + 2]);
        }
    }
    if (lane == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const double t = threadIdx.x == 0 ? red[0] + red[1] + red[2] + red[3] : 0.0;
    __syncthreads();
    grid_sum_store(t, 0.0, 1, gr, s + 1, nullptr, red);
}
// u += w du ; r -= w A du ; z = Dinv r ; s[4] += z'r      (w = s[0] / s[1])
template <class T>
__global__ __launch_bounds__(256) void k_cg_update(const T* __restrict__ Dinv, const T* __restrict__ du, const T* __restrict__ dAu, T* __restrict__ u, T* __restrict__ r, T* __restrict__ z, int n, double* s, GridRed gr)
{
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (!cg_active(s[0], s[8])) return; // s[0] is the current z'r since k_cg_spmv_dot, and nothing writes it here
    const double omega = s[0] / s[1];
    const T wp = (T)omega, wm = (T)(-omega);
    double part = 0;
    if (i < n) {
        T rr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            u[3 * (int64_t)i + c] += wp * du[3 * (int64_t)i + c];
            rr[c] = r[3 * (int64_t)i + c] + wm * dAu[3 * (int64_t)i + c];
            r[3 * (int64_t)i + c] = rr[c];
        }
        const T* d = Dinv + 9 * (int64_t)i;
        const T z0 = d[0] * rr[0] + d[3] * rr[1] + d[6] * rr[2], z1 = d[1] * rr[0] + d[4] * rr[1] + d[7] * rr[2], z2 = d[2] * rr[0] + d[5] * rr[1] + d[8] * rr[2];
        z[3 * (int64_t)i] = z0, z[3 * (int64_t)i + 1] = z1, z[3 * (int64_t)i + 2] = z2;
        part = (double)(z0 * rr[0]) + (double)(z1 * rr[1]) + (double)(z2 * rr[2]);
    }
    const double t = block_sum_256<double>(part, red);
    grid_sum_store(t, 0.0, 1, gr, s + 4, nullptr, red);
}
// du = z + b du      (b = s[4] / s[0])
template <class T>
__global__ void k_cg_direction(size_t n3, const T* __restrict__ z, T* __restrict__ du, double* s, double* hm, double* ticket, double ticket_val)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = cg_active(s[0], s[8]);
    if (act && i < n3) du[i] = z[i] + (T)(s[4] / s[0]) * du[i];
    if (i == 0) { // nobody reads s[6] / s[9] in this launch
        if (act) {
            s[6] = 1.0;
            const double cnt = s[9] + 1.0;
            s[9] = cnt;
            hm[0] = s[4], hm[1] = cnt;
        }
        if (ticket) host_ticket_store(ticket, ticket_val); // last launch of a group: the host is waiting for hm (Ctx::wait_ticket)
    }
}
// ---- cg_smooth on a SMALL top level as ONE persistent launch (C2 level 2: 5.4 k rows, a 49 MB matrix, ~6 iterations per V-cycle): the
// three launches per iteration above cost ~20 us each there, an iteration's arithmetic 10 us.  One workgroup per compute unit at most, rows
// dealt to wavefronts once and for all (a wavefront touches only its own rows of u, r, z, dAu: plain loads / stores); du is the one vector
// read across workgroups (the SpMV's gathers): published with write-through stores and read with agent-scope loads, like k_gs_sweep's
// unknowns.  Three grid barriers per iteration (arrival counter + spin, all workgroups are resident); a dot product = every workgroup
// deposits its partial sum before the barrier and adds ALL deposits in index order after it, so every workgroup holds the same bits and
// takes the same exit decision.  Same recurrences as k_cg_spmv_dot / k_cg_update / k_cg_direction; only the association of the dot
// products differs.  count / exitc are zero between launches (the last workgroup to leave resets them).
// RPW > 0 (round 6): a wavefront has at most RPW rows, and everything of them LIVES IN REGISTERS for the whole solve — the matrix row (lane k holds entries
// k and k + 64: 18 scalars + 2 column ids a row), and in lane 0 D^-1, u, r, z, du, dAu.  An iteration then goes to memory for the gathers of du in the product
// and the publication of the new du, nothing else (the streaming version walks six dependent round trips an iteration besides its three barriers: 176 us a
// solve at C2's level 2, 5.4 k rows, of which the barriers are the smaller part).  Same arithmetic, bit-identical results.
template <class T, int RPW = 0>
__global__ __launch_bounds__(1024) void k_cg_persist(const int32_t* __restrict__ col, const T* __restrict__ val, const T* __restrict__ Dinv, const T* __restrict__ init, T* __restrict__ u,
    T* __restrict__ r, T* __restrict__ z, T* du, T* __restrict__ dAu, int n, int max_iters, unsigned phase0 /*barriers this context's persistent solves have passed so far, mod 4*/, double* dep /*[4][gridDim.x][SS]*/, int SS /*doubles between two workgroups' slots*/, double* hm, double* ticket, double ticket_val,
    int* err /*pinned host word (k_gs_sweep's): a barrier that does not complete — the workgroups are not all resident because something else holds the chip — sets it; the host redoes the solve with launches*/)
{
    __shared__ double red[32], sres[2];
    __shared__ int s_bail;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int G = gridDim.x, wg = blockIdx.x;
    unsigned phase = phase0;
    if (tid == 0) s_bail = 0;
    auto ldu = [&](int64_t j, int c) { return __hip_atomic_load(du + 3 * j + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto sdu = [&](int64_t j, int c, T v) { __hip_atomic_store(du + 3 * j + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // Grid barrier + two dot products in one: workgroup-wide fixed-order sum of the wavefronts' lane-0 values, deposit, wait until every workgroup's
    // deposit of this phase is there, ordered sum of all deposits -> (o0, o1) everywhere.  The deposits are their own arrival flags (round 6; rounds 4 - 5:
    // deposit, wait for it to be performed, two-level arrival counters, poll, then fetch the deposits: five dependent trips to the memory side, ~5 us a
    // barrier, three barriers an iteration): four rotating sets of slots, a slot holds a signalling-NaN pattern no sum can produce until its owner writes
    // the phase's sums (write-through stores, agent-scope loads: k_gs_sweep's hand-off); with the deposit of phase p a workgroup resets its slots of phase
    // p + 2 — everybody has left phase p - 2, the previous tenant of that set, before anybody deposits for p - 1.  Wavefront 0 polls and sums the first
    // value, wavefront 1 the second: a lane adds slots lane, lane + 64, .. in ascending order, then the fixed DPP tree: the same bits everywhere.
    auto all_sum = [&](double v0, double v1, double& o0, double& o1) -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every thread's write-through stores of du have been performed before its workgroup deposits
        if (lane == 0) red[w] = v0, red[16 + w] = v1;
        __syncthreads();
        // (a workgroup's two sums share a line; the workgroups' lines are SS doubles apart, so that the 256 pollers' uncached loads spread over the memory
        // channels instead of queueing on the one or two that hold a packed 4 KB array)
        double* d = dep + (size_t)(phase & 3u) * G * SS;
        if (tid == 0) {
            double t0 = 0, t1 = 0;
            for (int k = 0; k < 16; ++k) t0 += red[k], t1 += red[16 + k];
            double* dn = dep + (size_t)((phase + 2u) & 3u) * G * SS + (size_t)wg * SS;
            __hip_atomic_store((unsigned long long*)dn, GsUnset<double>::bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((unsigned long long*)(dn + 1), GsUnset<double>::bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d + (size_t)wg * SS, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d + (size_t)wg * SS + 1, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (w < 2) {
            const double* dv = d + w;
            double a = 0;
            int spins = 0, bail = 0;
            for (;;) {
                bool all = true;
                a = 0;
                for (int k = lane; k < G; k += 64) {
                    const double x = __hip_atomic_load(dv + (size_t)k * SS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    all = all && !GsUnset<double>::is(x);
                    a += x;
                }
                if (__all(all)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 21) || ((spins & 255) == 0 && *(volatile int*)err)) { // (wave-uniform)
                    *(volatile int*)err = 1;
                    bail = 1;
                    break;
                }
            }
            a = wave_sum(a);
            if (lane == 0) {
                sres[w] = a;
                if (bail) s_bail = 1;
            }
        }
        __syncthreads();
        if (s_bail) return false; // workgroup-uniform
        o0 = sres[0], o1 = sres[1];
        ++phase;
        return true;
    };
    auto scale = [&](int64_t i, const T (&v)[3], T (&o)[3]) {
        const T* d = Dinv + 9 * i;
        o[0] = d[0] * v[0] + d[3] * v[1] + d[6] * v[2], o[1] = d[1] * v[0] + d[4] * v[1] + d[7] * v[2], o[2] = d[2] * v[0] + d[5] * v[1] + d[8] * v[2];
    };
    const int stride = 16 * G;
    if constexpr (RPW > 0) {
        // ---- the register-resident version
        int rowq[RPW];
        int32_t mc[RPW][2];
        T mv[RPW][2][9], di[RPW][9], uu[RPW][3], rr[RPW][3], zz[RPW][3], dd[RPW][3], ad[RPW][3];
        double p0 = 0, p1 = 0;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int row = wg * 16 + w + q * stride;
            rowq[q] = row < n ? row : -1;
            const int64_t rc = row < n ? row : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = min(lane + 64 * h, 124);
                mc[q][h] = col[rc * 125 + k];
#pragma unroll
                for (int e = 0; e < 9; ++e) mv[q][h][e] = (lane + 64 * h < 125) ? val[rc * 1125 + k * 9 + e] : (T)0;
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) di[q][e] = Dinv[9 * rc + e];
            T a[3], za[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) a[c] = init[3 * rc + c], rr[q][c] = r[3 * rc + c], uu[q][c] = u[3 * rc + c], ad[q][c] = (T)0;
#pragma unroll
            for (int c = 0; c < 3; ++c) za[c] = di[q][c] * a[0] + di[q][3 + c] * a[1] + di[q][6 + c] * a[2], zz[q][c] = di[q][c] * rr[q][0] + di[q][3 + c] * rr[q][1] + di[q][6 + c] * rr[q][2];
            if (lane == 0 && rowq[q] >= 0) {
                p0 += (double)(za[0] * a[0]) + (double)(za[1] * a[1]) + (double)(za[2] * a[2]);
                p1 += (double)(zz[q][0] * rr[q][0]) + (double)(zz[q][1] * rr[q][1]) + (double)(zz[q][2] * rr[q][2]);
                for (int c = 0; c < 3; ++c) sdu(row, c, zz[q][c]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) dd[q][c] = zz[q][c];
        }
        double zTr0, zTr;
        if (!all_sum(p0, p1, zTr0, zTr)) return;
        const double tol = (double)(T)(zTr0 * 0.25); // cgratio 0.5, squared (MultigridPreconditioner.h:203-209)
        // The product gathers du from LDS: behind the barrier that publishes it every workgroup copies the WHOLE vector (3 n scalars: 140 KB at C2's level 2)
        // with 16-byte uncached loads, coalesced — 33 MB an iteration over the chip.  The streaming version gathers du entry by entry with agent-scope loads,
        // which pass the L2 one 8-byte request at a time: 2 M of them an iteration (5.4 k rows x 125 entries x 3), ~50 us — that, not the barriers, was an
        // iteration's cost (measured: neither a cheaper barrier nor one barrier fewer moved it; cached gathers behind an agent-scope acquire fence, which
        // invalidates the XCD's L2 from every wavefront, cost 2.6 x more).
        extern __shared__ double cg_lds[]; // [3 n] du
        T* ldsdu = (T*)cg_lds;
        const int n3 = 3 * n;
        auto pull_du = [&]() {
            const int n16 = n3 >> 1; // 16-byte pieces
            for (int e0 = 0; e0 < n16; e0 += 4 * 1024) {
                double2 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = min(e0 + k * 1024 + tid, n16 - 1);
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[k]) : "v"((const double2*)du + e) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = e0 + k * 1024 + tid;
                    if (e < n16) ((double2*)ldsdu)[e] = v[k];
                }
            }
            if ((n3 & 1) && tid == 0) ldsdu[n3 - 1] = ldu(n - 1, 2);
            __syncthreads();
        };
        int cnt = 0;
#ifdef HOT_AB_KERNELS
        unsigned long long tk[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t0_ = wall_clock64(); // A/B build, HOT_CG_DBG: 100 MHz clock of workgroup 0 between the phases, summed over the iterations
#define CG_TK(i) \
    do { \
        const unsigned long long t_ = wall_clock64(); \
        tk[i] += t_ - t0_, t0_ = t_; \
    } while (0)
#else
#define CG_TK(i)
#endif
        for (; cnt < max_iters && cg_active(zTr, tol); ++cnt) {
            pull_du();
            CG_TK(0);
            double pd = 0;
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (lane + 64 * h < 125) {
                        const int j = mc[q][h];
                        const T* b = mv[q][h];
                        const T x0 = ldsdu[3 * j], x1 = ldsdu[3 * j + 1], x2 = ldsdu[3 * j + 2];
                        s0 += b[0] * x0 + b[3] * x1 + b[6] * x2;
                        s1 += b[1] * x0 + b[4] * x1 + b[7] * x2;
                        s2 += b[2] * x0 + b[5] * x1 + b[8] * x2;
                    }
                }
                s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
                ad[q][0] = s0, ad[q][1] = s1, ad[q][2] = s2;
                if (lane == 0 && rowq[q] >= 0) pd += (double)(s0 * dd[q][0]) + (double)(s1 * dd[q][1]) + (double)(s2 * dd[q][2]);
            }
            CG_TK(1);
            double dAd, unused;
            if (!all_sum(pd, 0.0, dAd, unused)) return;
            CG_TK(2);
            const double omega = zTr / dAd;
            const T wp = (T)omega, wm = (T)(-omega);
            double pz = 0;
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
#pragma unroll
                for (int c = 0; c < 3; ++c) uu[q][c] += wp * dd[q][c], rr[q][c] = rr[q][c] + wm * ad[q][c];
#pragma unroll
                for (int c = 0; c < 3; ++c) zz[q][c] = di[q][c] * rr[q][0] + di[q][3 + c] * rr[q][1] + di[q][6 + c] * rr[q][2];
                if (lane == 0 && rowq[q] >= 0) pz += (double)(zz[q][0] * rr[q][0]) + (double)(zz[q][1] * rr[q][1]) + (double)(zz[q][2] * rr[q][2]);
            }
            CG_TK(3);
            double zTrNew;
            if (!all_sum(pz, 0.0, zTrNew, unused)) return;
            CG_TK(4);
            const T beta = (T)(zTrNew / zTr);
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
#pragma unroll
                for (int c = 0; c < 3; ++c) dd[q][c] = zz[q][c] + beta * dd[q][c];
                if (lane == 0 && rowq[q] >= 0)
                    for (int c = 0; c < 3; ++c) sdu(rowq[q], c, dd[q][c]);
            }
            zTr = zTrNew;
            CG_TK(5);
            if (!all_sum(0.0, 0.0, unused, unused)) return;
            CG_TK(6);
        }
#ifdef HOT_AB_KERNELS
        if (tid == 0 && wg == 0)
            for (int i = 0; i < 7; ++i) hm[8 + i] = (double)tk[i];
#endif
        // what the streaming version leaves in memory: u, r, z, dAu of the last iteration (du has been published)
#pragma unroll
        for (int q = 0; q < RPW; ++q)
            if (lane == 0 && rowq[q] >= 0)
                for (int c = 0; c < 3; ++c) {
                    const int64_t e = 3 * (int64_t)rowq[q] + c;
                    u[e] = uu[q][c], r[e] = rr[q][c], z[e] = zz[q][c], dAu[e] = ad[q][c];
                }
        if (tid == 0 && wg == 0) {
            hm[0] = zTr, hm[1] = (double)cnt, hm[2] = tol;
            if (ticket) host_ticket_store(ticket, ticket_val);
        }
        return;
    }
    // ---- set-up: z'r of the reference residual (the tolerance) and of r; du = z = Dinv r
    double p0 = 0, p1 = 0;
    for (int row = wg * 16 + w; row < n; row += stride) {
        if (lane == 0) {
            T a[3] = { init[3 * (int64_t)row], init[3 * (int64_t)row + 1], init[3 * (int64_t)row + 2] }, za[3];
            scale(row, a, za);
            p0 += (double)(za[0] * a[0]) + (double)(za[1] * a[1]) + (double)(za[2] * a[2]);
            T b[3] = { r[3 * (int64_t)row], r[3 * (int64_t)row + 1], r[3 * (int64_t)row + 2] }, zb[3];
            scale(row, b, zb);
            for (int c = 0; c < 3; ++c) z[3 * (int64_t)row + c] = zb[c], sdu(row, c, zb[c]);
            p1 += (double)(zb[0] * b[0]) + (double)(zb[1] * b[1]) + (double)(zb[2] * b[2]);
        }
    }
    double zTr0, zTr;
    if (!all_sum(p0, p1, zTr0, zTr)) return;
    const double tol = (double)(T)(zTr0 * 0.25); // cgratio 0.5, squared (MultigridPreconditioner.h:203-209)
    int cnt = 0;
    for (; cnt < max_iters && cg_active(zTr, tol); ++cnt) {
        // dAu = A du on this workgroup's rows, du'dAu
        double pd = 0;
        for (int row = wg * 16 + w; row < n; row += stride) {
            const int32_t* c = col + (int64_t)row * 125;
            const T* v = val + (int64_t)row * 1125;
            T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = lane + 64 * q;
                if (k < 125) {
                    const int64_t j = c[k];
                    const T* b = v + k * 9;
                    const T x0 = ldu(j, 0), x1 = ldu(j, 1), x2 = ldu(j, 2);
                    s0 += b[0] * x0 + b[3] * x1 + b[6] * x2;
                    s1 += b[1] * x0 + b[4] * x1 + b[7] * x2;
                    s2 += b[2] * x0 + b[5] * x1 + b[8] * x2;
                }
            }
            s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
            if (lane == 0) {
                dAu[3 * (int64_t)row] = s0, dAu[3 * (int64_t)row + 1] = s1, dAu[3 * (int64_t)row + 2] = s2;
                pd += (double)(s0 * ldu(row, 0)) + (double)(s1 * ldu(row, 1)) + (double)(s2 * ldu(row, 2));
            }
        }
        double dAd, unused;
        if (!all_sum(pd, 0.0, dAd, unused)) return;
        // u += w du ; r -= w A du ; z = Dinv r ; z'r
        const double omega = zTr / dAd;
        const T wp = (T)omega, wm = (T)(-omega);
        double pz = 0;
        for (int row = wg * 16 + w; row < n; row += stride) {
            if (lane == 0) {
                T rr[3], zz[3];
                for (int c = 0; c < 3; ++c) {
                    u[3 * (int64_t)row + c] += wp * ldu(row, c);
                    rr[c] = r[3 * (int64_t)row + c] + wm * dAu[3 * (int64_t)row + c];
                    r[3 * (int64_t)row + c] = rr[c];
                }
                scale(row, rr, zz);
                for (int c = 0; c < 3; ++c) z[3 * (int64_t)row + c] = zz[c];
                pz += (double)(zz[0] * rr[0]) + (double)(zz[1] * rr[1]) + (double)(zz[2] * rr[2]);
            }
        }
        double zTrNew;
        if (!all_sum(pz, 0.0, zTrNew, unused)) return;
        // du = z + b du, published before the next SpMV gathers it
        const T beta = (T)(zTrNew / zTr);
        for (int row = wg * 16 + w; row < n; row += stride)
            if (lane == 0)
                for (int c = 0; c < 3; ++c) sdu(row, c, z[3 * (int64_t)row + c] + beta * ldu(row, c));
        zTr = zTrNew;
        if (!all_sum(0.0, 0.0, unused, unused)) return;
    }
    if (tid == 0) {
        if (wg == 0) {
            hm[0] = zTr, hm[1] = (double)cnt, hm[2] = tol;
            if (ticket) host_ticket_store(ticket, ticket_val);
        }
    }
}

template <class T>
__global__ void k_scal_v(size_t n, T a, T* x)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= a;
}
template <class T>
void Ctx<T>::scal(size_t n, T a, T* x)
{
    HOT_LAUNCH(this, "scal", k_scal_v<T>, div_up(n, 256), 256, 0, n, a, x);
}
template <class T>
void Ctx<T>::spmv_dev(Level<T>& L, const T* x, T* y)
{
    if (L.part && halo_mode()) halo_gather(L, const_cast<T*>(x)); // the entries of x the owned rows couple to
    HOT_LAUNCH(this, lname("spmv", L.id).c_str(), k_spmv<T>, xcd_grid(div_up(L.n, 4)), 256, 0, L.col.p, L.val.p, x, y, L.n, L.mask());
    if (!halo_mode()) exchange(L, y, -1); // first-generation sharding: every rank computed the rows it owns; all of y is needed by the replicated vector algebra
}

// ------------------------------------------------------------------------------------------------ transfers
template <class T>
__global__ void k_restrict(const int32_t* __restrict__ child, const T* __restrict__ fine, T* coarse, int nc, const uint8_t* __restrict__ coarse_own, const uint8_t* __restrict__ fine_own,
    T* unset /*not null: the coarse level's GS forward target, marked "not written yet" here (see smooth_dev)*/, T* zero_out = nullptr /*not null: a coarse-level vector cleared on the way (the V-cycle's coarse iterate, instead of a fill launch)*/)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * nc) return;
    if (unset) gs_store_unset(unset + e);
    if (zero_out) zero_out[e] = (T)0;
    int I = e / 3, d = e - 3 * I;
    if (coarse_own && !coarse_own[I]) return; // sharded, both levels partitioned: the coarse rows this rank owns (their children are in the fine halo)
    // all 27 child ids first, then all 27 values (clamped index, dropped by the select): two rounds of independent loads instead of
    // 27 dependent pairs; the sum keeps the child order
    int ci[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) ci[q] = child[I * 27 + q];
    T fv[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) fv[q] = fine[3 * (int64_t)(ci[q] < 0 ? 0 : ci[q]) + d];
    T s = 0;
#pragma unroll
    for (int q = 0; q < 27; ++q) {
        const T w = ((q / 9 != 1) ? (T)0.5 : (T)1) * (((q / 3) % 3 != 1) ? (T)0.5 : (T)1) * ((q % 3 != 1) ? (T)0.5 : (T)1);
        s = (ci[q] < 0 || (fine_own && !fine_own[ci[q]])) ? s : s + w * fv[q]; // fine_own: partial sum over this rank's fine rows (replicated coarse level, all-reduced)
    }
    coarse[e] = s;
}
template <class T>
__global__ void k_prolong(const int32_t* __restrict__ pcol, const T* __restrict__ pw, const T* __restrict__ coarse, T* fine, int n, const uint8_t* __restrict__ own)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * n) return;
    int i = e / 3, d = e - 3 * i;
    if (own && !own[i]) return;
    T s = 0;
    for (int l = 0; l < 8; ++l) s += pw[8 * (int64_t)i + l] * coarse[3 * (int64_t)pcol[8 * (int64_t)i + l] + d];
    fine[e] = s;
}
// Same decision as smooth_dev / vcycle_dev take (kind 5, chained launch, unknowns as their own flags).
template <class T>
bool Ctx<T>::gs_marks_wanted(int level) const
{
    const Level<T>& L = *levels[level];
    const bool baseline = cfg.useBaselineMultigrid != 0;
    const int splitLevel = cfg.topDownMGS ? 1 : cfg.levelCnt - 1;
    const int kind = level < splitLevel ? (baseline ? 5 : cfg.smoother) : (baseline ? 2 : cfg.coarseSolver);
    if (kind != 5 || gs_no_chain || !L.split || L.part || !L.tmp.p || cfg.gs_chain == 1) return false;
    if (ab_flag("HOT_SIMPLE_GS") || ab_flag("HOT_GS_PASS_COUNTERS") || ab_flag("HOT_GS_BLOCK_FLAGS")) return false;
    int max_nb = 0;
    for (int c = 0; c < 8; ++c) max_nb = std::max(max_nb, L.color_block_begin[c + 1] - L.color_block_begin[c]);
    return cfg.gs_chain == 2 || max_nb <= 256;
}
template <class T>
void Ctx<T>::restrict_dev(int level, const T* fine, T* coarse, T* zero_coarse)
{
    Level<T>& C = *levels[level + 1];
    Level<T>& F = *levels[level];
    if (F.part && halo_mode()) {
        if (zero_coarse) zero(3 * (size_t)C.n, zero_coarse);
        if (C.part) { // owner of a coarse row sums its 27 children: those owned elsewhere come with the fine level's halo
            halo_gather(F, const_cast<T*>(fine));
            HOT_LAUNCH(this, "restrict", k_restrict<T>, div_up(3 * (size_t)C.n, 256), 256, 0, C.child.p, fine, coarse, C.n, C.own.p, (const uint8_t*)nullptr, (T*)nullptr);
        }
        else { // replicated coarse level: every rank sums the children it owns, one all-reduce of the (small) coarse vector completes the rows
            HOT_LAUNCH(this, "restrict", k_restrict<T>, div_up(3 * (size_t)C.n, 256), 256, 0, C.child.p, fine, coarse, C.n, (const uint8_t*)nullptr, F.own.p, (T*)nullptr);
            CommTag tag(this, "coarse_vector_allreduce");
            c_allreduce(coarse, 3 * (int64_t)C.n, REAL, HOT_COMM_SUM, true);
        }
        return;
    }
    // (every smoother on the coarse level is preceded by a restriction into it: its GS forward target gets its marks here)
    T* mark = gs_marks_wanted(level + 1) ? C.tmp.p : (T*)nullptr;
    HOT_LAUNCH(this, "restrict", k_restrict<T>, div_up(3 * (size_t)C.n, 256), 256, 0, C.child.p, fine, coarse, C.n, (const uint8_t*)nullptr, (const uint8_t*)nullptr, mark, zero_coarse);
    unset_level = mark ? level + 1 : -1;
}
template <class T>
void Ctx<T>::prolong_dev(int level, const T* coarse, T* fine)
{
    Level<T>& F = *levels[level];
    Level<T>& C = *levels[level + 1];
    const bool hm = F.part && halo_mode();
    if (hm && C.part) halo_gather(C, const_cast<T*>(coarse)); // the parents / window columns of the fine rows this rank owns (also read by the k_apmv_sub that follows)
    HOT_LAUNCH(this, "prolong", k_prolong<T>, div_up(3 * (size_t)F.n, 256), 256, 0, F.pcol.p, F.pw.p, coarse, fine, F.n, hm ? F.own.p : (const uint8_t*)nullptr);
}

// ------------------------------------------------------------------------------------------------ smoothers
// mr_i = Dinv_i r_i (scale_diagonal_{entry,block}_inverse, MultigridPreconditioner.h:143-154)
template <class T>
__global__ void k_scale(const T* __restrict__ D, const T* __restrict__ r, T* mr, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T* d = D + 9 * (int64_t)i;
    T a = r[3 * (int64_t)i], b = r[3 * (int64_t)i + 1], c = r[3 * (int64_t)i + 2];
    mr[3 * (int64_t)i] = d[0] * a + d[3] * b + d[6] * c;
    mr[3 * (int64_t)i + 1] = d[1] * a + d[4] * b + d[7] * c;
    mr[3 * (int64_t)i + 2] = d[2] * a + d[5] * b + d[8] * c;
}

// SquareMatrix::estimate2norm (reference Projects/multigrid/SquareMatrix.h:375-475, active #else branch): power iteration on
// A*A from a +-1 start vector.  The reference seeds the signs with srand(time(NULL)); here they are a fixed hash of the
// entry index, the converged value does not depend on it within the 1e-6 stopping tolerance.
template <class T>
__global__ void k_cheb_start(T* v, size_t n3)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n3) v[e] = ((((unsigned)e * 2654435761u) >> 16) & 1u) ? (T)1 : (T)-1;
}
template <class T>
__global__ void k_abs(T* v, size_t n3)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n3) v[e] = v[e] < 0 ? -v[e] : v[e];
}
template <class T>
__global__ void k_div1(T* v, T c, size_t n3)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n3) v[e] = v[e] / c;
}
template <class T>
void Ctx<T>::estimate_2norm(Level<T>& L, double tol)
{
    MaskScope mscope(this, halo_mode() ? L.mask() : nullptr); // this level's vectors (halo mode: the rows the rank owns)
    constexpr int MaxIters = 512;
    size_t n3 = 3 * (size_t)L.n;
    T *v = L.du.p, *x = L.dAu.p; // work vectors of the level, free while the hierarchy is being built
    HOT_LAUNCH(this, "cheb_start", k_cheb_start<T>, div_up(n3, 256), 256, 0, v, n3);
    spmv_dev(L, v, x);
    HOT_LAUNCH(this, "cheb_abs", k_abs<T>, div_up(n3, 256), 256, 0, x, n3);
    T e = (T)std::sqrt(dot_host(n3, x, x));
    if (e == 0) {
        L.lMin = L.lMax = 0;
        return;
    }
    HOT_LAUNCH(this, "cheb_div", k_div1<T>, div_up(n3, 256), 256, 0, x, e, n3);
    T e0 = 0;
    for (int iter = 0; iter < MaxIters && std::abs(e - e0) > (T)tol * e; ++iter) {
        e0 = e;
        spmv_dev(L, x, v);
        spmv_dev(L, v, x);
        T normx = (T)std::sqrt(dot_host(n3, x, x));
        e = normx / (T)std::sqrt(dot_host(n3, v, v));
        HOT_LAUNCH(this, "cheb_div", k_div1<T>, div_up(n3, 256), 256, 0, x, normx, n3);
    }
    L.lMax = e;
    L.lMin = L.lMax / 30; // "experience" (:473)
}

// out_i = D_i in_i for an arbitrary array of 3x3 blocks (matrix-free block-diagonal preconditioner)
template <class T>
void Ctx<T>::block_apply_dev(const T* D, const T* in, T* out, int n)
{
    HOT_LAUNCH(this, "diag_scale", k_scale<T>, div_up(n, 256), 256, 0, D, in, out, n);
}
template <class T>
void Ctx<T>::scale_dev(Level<T>& L, const T* in, T* out)
{
    HOT_LAUNCH(this, "diag_scale", k_scale<T>, div_up(L.n, 256), 256, 0, L.diagInv.p, in, out, L.n);
}

#ifdef HOT_AB_KERNELS
#include "ab_src/mg_solve_ab1.hip"
#endif

// Two-phase block GS (the production path; k_gs_color above is the simple reference kernel kept for A/B checks).
// The reference sweeps the nodes of one 4^3 colour block sequentially (MultigridPreconditioner.h:266-318).  Here a
// colour block is cut into 64/SB consecutive sub-blocks of SB nodes; one launch per colour, one workgroup per block, which
// walks the block's sub-blocks in sweep order (a launch per (colour, sub-block) behind HOT_GS_SPLIT_LAUNCHES).  Nodes of the
// same block that belong to an earlier sub-block are final in global memory by then (stored before a workgroup barrier) and
// are treated like any other preceding node, so the sequence of updates each node sees is the reference's.  SB = 32 keeps the LDS footprint at 36 KB (fp64), several workgroups
// per CU overlap their phases, and one launch fits the chip in a single round.
//   phase A (all waves, bandwidth-bound): every wave streams the preceding half of whole matrix rows (rows are
//           regrouped by k_gs_split_rows, lane = slot).  Couplings to nodes outside the sub-block are folded into
//           s_i = rhs_i - sum A_ij x_j ; couplings inside it are copied into an LDS triangular array laid out by
//           (column, row) so that phase B reads it conflict-free.
//   phase B (1 wave, latency-bound but LDS/register only): right-looking block substitution, lane = row:
//           step c: lane c finalises h_c = Dinv_c s_c, broadcasts it, every later row subtracts L[row][c] h_c.
// Only the association order of the row sums differs from k_gs_color.
template <class T, int SB>
struct GsLds {
    static constexpr int TRI = SB * (SB - 1) / 2 + 1; // ordered pairs + one always-zero entry (last) for masked lanes
    static constexpr size_t bytes = (size_t)9 * TRI * sizeof(T) + SB * 3 * sizeof(T) + 5 * SB * sizeof(int32_t);
};
// inverse image of a colour block and direction: nine planes of TRI scalars, padded to a multiple of 16 bytes (the LDS-DMA pieces)
template <class T>
struct GsWinv {
    static constexpr int img_elems = (9 * GsLds<T, 64>::TRI + 15) / 16 * 16;
};
template <int SB>
__device__ __forceinline__ int gs_tri_fwd(int row, int colm) { return (SB - 1) * colm - (colm * (colm - 1)) / 2 + (row - colm - 1); } // row > colm
__device__ __forceinline__ int gs_tri_bwd(int row, int colm) { return (colm * (colm - 1)) / 2 + row; } // row < colm
// entry (row, column) of a block's inverse image (k_gs_winv -> k_gs_sweep<.., WINV>): packed row by row, so that the lanes of a row (lane = column)
// read consecutive scalars of each of the nine planes; forward: columns before the row, backward (mirrored): columns after it
template <bool FWD>
__device__ __forceinline__ int gs_winv_idx(int row, int colm) { return FWD ? (row * (row - 1)) / 2 + colm : ((63 - row) * (62 - row)) / 2 + (63 - colm); }

// The LDS triangle holds -(Dinv_i A_ij) and the right-hand sides Dinv_i s_i, so that the substitution phase is a pure
// multiply-add chain: h_i = Dinv_i s_i + sum_j (-(Dinv_i A_ij)) h_j  (same value as Dinv_i (s_i - sum_j A_ij h_j) up to
// the association of the 3x3 products).
template <class T>
__device__ __forceinline__ void gs_store_tri(T* tri, int TRI, int idx, const T* __restrict__ di, const T (&b)[9])
{
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) tri[(r + 3 * c) * TRI + idx] = -(di[r] * b[3 * c] + di[r + 3] * b[3 * c + 1] + di[r + 6] * b[3 * c + 2]);
}
template <class T>
__device__ __forceinline__ void gs_store_rhs(T* sv, int ii, const T* __restrict__ di, T r0, T r1, T r2)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) sv[ii * 3 + r] = di[r] * r0 + di[r + 3] * r1 + di[r + 6] * r2;
}

// ---- inverse images of the in-block triangles (chained levels).  With N the strictly lower (forward) / upper (backward) in-block couplings of a
// colour block, premultiplied as -(D_r^-1 A_rc), the block's half-sweep solve is h = a + N h, i.e. h = (I - N)^-1 a =: a + W a.  W is dense
// (every node of a 4^3 block reaches every later one through the chain), 64 x 63 / 2 blocks of 3 x 3 = 145 KB in fp64 — what the substitution
// reads of N is 2/3 of that, so on a level bound by the 64-step dependency chain and not by bytes the product with W is the better trade.
// One workgroup per (block, direction): N into the LDS triangle exactly as the sweep kernels file it, then column c0 of W is the substitution
// applied to the three unit vectors of position c0 (lane = row, the three right-hand sides together: 9 LDS reads, 27 multiply-adds per step),
// four columns per wavefront; the result goes out row-packed (gs_winv_idx).  The backward image follows from the forward one (A symmetric).
// Built once per hierarchy build.
template <class T>
__global__ __launch_bounds__(1024) void k_gs_winv(const int32_t* __restrict__ col, const T* __restrict__ val, const uint32_t* __restrict__ ckey, const int32_t* __restrict__ gs_order,
    const int32_t* __restrict__ block_start, const int32_t* __restrict__ rowcnt, const T* __restrict__ diagBlockInv, const T* __restrict__ diagVal, T* __restrict__ gs_w)
{
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    constexpr int TRI = GsLds<T, 64>::TRI;
    T* tri = (T*)gs_smem; // [9][TRI]
    T* sDi = tri + 9 * TRI; // [64][9] D^-1 of the block's rows
    T* sDv = sDi + 64 * 9; // [64][9] D
    int32_t* nodes = (int32_t*)(sDv + 64 * 9); // [64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.x;
    const int start = block_start[b], cnt = min(64, block_start[b + 1] - start);
    for (int e = tid; e < 9 * TRI; e += 1024) tri[e] = (T)0;
    if (tid < 64) nodes[tid] = tid < cnt ? gs_order[start + tid] : -1;
    __syncthreads();
    for (int e = tid; e < 9 * cnt; e += 1024) {
        const int64_t i = nodes[e / 9];
        sDi[e] = diagBlockInv[9 * i + e % 9], sDv[e] = diagVal[9 * i + e % 9];
    }
    for (int e = tid; e < 64 * cnt; e += 1024) { // (row position, in-block slot of the row's preceding half)
        const int ii = e >> 6, ks = e & 63;
        const int64_t i = nodes[ii];
        const int po = rowcnt[4 * i], pi = rowcnt[4 * i + 1];
        if (ks >= pi) continue;
        const int k = po + ks, j = col[i * 125 + k];
        const int l = (int)(ckey[j] & 127u) - 1;
        T bb[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) bb[q] = val[(i * 125 + k) * 9 + q];
        gs_store_tri<T>(tri, TRI, gs_tri_fwd<64>(ii, l), diagBlockInv + 9 * i, bb);
    }
    __syncthreads();
    T* outf = gs_w + ((size_t)b * 2) * GsWinv<T>::img_elems;
    T* outb = outf + GsWinv<T>::img_elems;
    if (tid < 9) outf[tid * TRI + TRI - 1] = (T)0, outb[tid * TRI + TRI - 1] = (T)0; // the entry masked lanes read
    // ---- forward image: column c0 of W = the substitution applied to the three unit vectors of position c0
    for (int k4 = 0; k4 < 4; ++k4) {
        const int c0 = w + 16 * k4; // wave-uniform
        if (c0 >= cnt) break;
        T a[3][3]; // [right-hand side s][component]: column c0 of W, row = lane, as it builds up
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[s_][q] = (lane == c0 && s_ == q) ? (T)1 : (T)0;
        for (int c = c0; c < cnt - 1; ++c) { // every column but the last has later rows to update
            const bool act = lane > c && lane < cnt;
            const int idx = act ? gs_tri_fwd<64>(lane, c) : TRI - 1;
            T Lc[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) Lc[e] = tri[e * TRI + idx];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const T b0 = lane_bcast(a[s_][0], c), b1 = lane_bcast(a[s_][1], c), b2 = lane_bcast(a[s_][2], c);
                a[s_][0] = fma(Lc[0], b0, a[s_][0]), a[s_][1] = fma(Lc[1], b0, a[s_][1]), a[s_][2] = fma(Lc[2], b0, a[s_][2]);
                a[s_][0] = fma(Lc[3], b1, a[s_][0]), a[s_][1] = fma(Lc[4], b1, a[s_][1]), a[s_][2] = fma(Lc[5], b1, a[s_][2]);
                a[s_][0] = fma(Lc[6], b2, a[s_][0]), a[s_][1] = fma(Lc[7], b2, a[s_][1]), a[s_][2] = fma(Lc[8], b2, a[s_][2]);
            }
        }
        if (lane > c0 && lane < cnt) { // W(lane, c0), strictly below the diagonal
            const int idx = gs_winv_idx<true>(lane, c0);
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int q = 0; q < 3; ++q) outf[(q + 3 * s_) * TRI + idx] = a[s_][q];
        }
    }
    // ---- backward image from the forward one: with U = L^T (A symmetric), I + W_b = (D + U)^-1 D = ((D + L)^-1)^T D = D^-1 (I + W_f)^T D, i.e.
    // W_b(r, c) = D_r^-1 W_f(c, r)^T D_c for c > r: two 3 x 3 products per entry instead of a second substitution
    __threadfence();
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 1024) {
        const int r = e >> 6, c = e & 63;
        if (c <= r || c >= cnt) continue;
        const int fi = gs_winv_idx<true>(c, r), bi = gs_winv_idx<false>(r, c);
        T M[9], T1[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = __builtin_nontemporal_load(outf + q * TRI + fi);
        const T* Dc = sDv + 9 * c;
        const T* Ir = sDi + 9 * r;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) T1[i + 3 * j] = M[3 * i] * Dc[3 * j] + M[1 + 3 * i] * Dc[1 + 3 * j] + M[2 + 3 * i] * Dc[2 + 3 * j]; // (M^T D_c)(i, j) = sum_k M(k, i) D_c(k, j)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) outb[(i + 3 * j) * TRI + bi] = Ir[i] * T1[3 * j] + Ir[i + 3] * T1[1 + 3 * j] + Ir[i + 6] * T1[2 + 3 * j];
    }
}
template <class T>
void Ctx<T>::build_gs_winv(Level<T>& L)
{
    constexpr size_t per = 2 * (size_t)GsWinv<T>::img_elems;
    L.gs_w.reserve(per * (size_t)L.nblocks + 256);
    const size_t lds = ((size_t)9 * GsLds<T, 64>::TRI + 2 * 64 * 9) * sizeof(T) + 64 * sizeof(int32_t);
    if (!attr_winv_set) {
        HOT_HIP(hipFuncSetAttribute((const void*)k_gs_winv<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_winv_set = true;
    }
    HOT_LAUNCH(this, lname("gs_winv", L.id).c_str(), k_gs_winv<T>, L.nblocks, 1024, lds, L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, L.rowcnt.p, L.diagBlockInv.p, L.diagVal.p, L.gs_w.p);
    L.gs_w_ready = true;
}

template <class T, bool FWD, int SB, bool WT = false>
__device__ __forceinline__ void gs_phase_b(const T* tri, const T* sv, const int32_t* nodes, int cnt, int lane, const T* __restrict__ diagVal, const T* __restrict__ diagBlockInv, T* x, T* hD,
    const T* ldsD = nullptr, T* ldsX = nullptr);

// six waves per SIMD (three 512-thread workgroups per CU: a colour of the finest level is resident in one round) = at most 80 VGPRs
#ifdef HOT_AB_KERNELS
// A/B build only, TIMING experiments with wrong results (tools/gs_where.py): bit 0 skip the substitution phase, bit 1 no x gathers,
// bit 2 no matrix value loads, bit 3 no phase A at all
__device__ int gs_dbg_flags = 0;
#define GS_DBG(bit) (gs_dbg_flags & (bit))
#else
#define GS_DBG(bit) 0
#endif
template <class T, bool FWD, int SB>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6))) void k_gs_block(const int32_t* __restrict__ col, const T* __restrict__ val, const uint32_t* __restrict__ ckey, const int32_t* __restrict__ gs_order,
    const int32_t* __restrict__ block_start, const T* __restrict__ diagVal, const T* __restrict__ diagBlockInv, const T* __restrict__ rhs, T* x, T* hD, int block0, int sub,
    const int32_t* __restrict__ rowcnt, const int32_t* __restrict__ gs_pad)
{
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    constexpr int TRI = GsLds<T, SB>::TRI;
    T* tri = (T*)gs_smem; // [9][TRI]
    T* sv = tri + 9 * TRI; // [SB][3]
    int32_t* nodes = (int32_t*)(sv + 3 * SB);
    int32_t* rcl = nodes + SB; // [SB][4] row class counts (from gs_pad)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = block0 + blockIdx.x;
    // sub-blocks [sub & 255, +nmerge) of the colour block are processed back to back by this workgroup, in sweep order: the
    // launch boundary between them (gap, dispatch ramp, header round trip: ~7 us of a ~30 us pass) is replaced by a barrier;
    // what the later sub-block reads of the earlier one was stored before the barrier by the same workgroup
    const int nmerge = max(sub >> 16, 1), sub_first = sub & 255;
    const int nthreads = blockDim.x, nwaves = blockDim.x >> 6;
    for (int m = 0; m < nmerge; ++m) {
    sub = FWD ? sub_first + m : sub_first + nmerge - 1 - m;
    const int lo = sub * SB; // first local index of this sub-block
    const int start = block_start[b] + lo, cnt = min(SB, block_start[b + 1] - start);
    if (cnt <= 0) continue; // workgroup-uniform
    if (m > 0) __syncthreads(); // the previous sub-block's substitution wave is done with the triangle / rhs / node tables
    for (int e = tid; e < 9 * TRI; e += nthreads) tri[e] = (T)0;
    if (tid < SB) {
        // one 32-byte record per position: node id + its row class counts (no block_start -> gs_order -> rowcnt chain)
        const int4 rec0 = *(const int4*)(gs_pad + 8 * ((int64_t)b * 64 + lo + tid));
        const int rec1 = gs_pad[8 * ((int64_t)b * 64 + lo + tid) + 4];
        nodes[tid] = rec0.x;
        rcl[4 * tid] = rec0.y, rcl[4 * tid + 1] = rec0.z, rcl[4 * tid + 2] = rec0.w, rcl[4 * tid + 3] = rec1;
    }
    __syncthreads();
    // ---------------- phase A: RQ rows of this wave are in flight at once (lane = slot of the needed half row)
    constexpr int RQ = 2;
    for (int t0 = 0; w + nwaves * t0 < cnt && !GS_DBG(8); t0 += RQ) {
        T bv[RQ][9];
        int jj[RQ], rowi[RQ], kb[RQ], ke[RQ], ib[RQ], ie[RQ];
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int ii = w + nwaves * (t0 + q);
            rowi[q] = -1, jj[q] = -1, kb[q] = ke[q] = ib[q] = ie[q] = 0;
#pragma unroll
            for (int e = 0; e < 9; ++e) bv[q][e] = (T)0;
            if (ii < cnt) { // wave-uniform
                const int i = __builtin_amdgcn_readfirstlane(nodes[ii]); // row id in an SGPR: its metadata loads are scalar
                rowi[q] = i;
                const int po = rcl[4 * ii], pi = rcl[4 * ii + 1], fi = rcl[4 * ii + 2], fo = rcl[4 * ii + 3];
                const int kbeg = FWD ? 0 : po + pi + 1, kend = FWD ? po + pi : po + pi + 1 + fi + fo;
                kb[q] = kbeg, ke[q] = kend, ib[q] = FWD ? po : kbeg, ie[q] = FWD ? po + pi : kbeg + fi;
                // unconditional loads from a clamped slot: predicated loads made the compiler serialise the value loads
                // behind s_waitcnt vmcnt(0); lanes past the range re-read its last slot and are dropped via jj < 0
                const int k = kbeg + lane, kc = max(min(k, kend - 1), 0);
                const int jl = col[(int64_t)i * 125 + kc];
                const T* bb = val + ((int64_t)i * 125 + kc) * 9;
                if (!GS_DBG(4)) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) bv[q][e] = bb[e];
                }
                jj[q] = k < kend ? jl : -1;
            }
        }
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int i = rowi[q], ii = w + nwaves * (t0 + q);
            if (i < 0) continue; // wave-uniform
            T di[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) di[e] = diagBlockInv[9 * (int64_t)i + e];
            const T rh0 = rhs[3 * (int64_t)i], rh1 = rhs[3 * (int64_t)i + 1], rh2 = rhs[3 * (int64_t)i + 2]; // requested early: not a dependent load after the reduction
            T s0 = 0, s1 = 0, s2 = 0;
            // couplings inside the sub-block go to the LDS triangle (the in-block slots of a regrouped row hold only
            // non-zero blocks, so the padded alias slots of SquareMatrix.h:563-566 cannot clobber an entry), the rest
            // is folded into the row sum
            auto entry = [&](int k, int j, const T (&b9)[9]) {
                int lj = -1;
                if (k >= ib[q] && k < ie[q]) {
                    const int l = (int)(ckey[j] & 127u) - 1 - lo;
                    if (l >= 0 && l < SB) lj = l;
                }
                if (lj >= 0)
                    gs_store_tri<T>(tri, TRI, FWD ? gs_tri_fwd<SB>(ii, lj) : gs_tri_bwd(ii, lj), di, b9);
                else {
                    const T x0 = GS_DBG(2) ? (T)1 : x[3 * (int64_t)j], x1 = GS_DBG(2) ? (T)1 : x[3 * (int64_t)j + 1], x2 = GS_DBG(2) ? (T)1 : x[3 * (int64_t)j + 2];
                    s0 += b9[0] * x0 + b9[3] * x1 + b9[6] * x2;
                    s1 += b9[1] * x0 + b9[4] * x1 + b9[7] * x2;
                    s2 += b9[2] * x0 + b9[5] * x1 + b9[8] * x2;
                }
            };
            if (jj[q] >= 0) entry(kb[q] + lane, jj[q], bv[q]);
            for (int k = kb[q] + 64 + lane; k < ke[q]; k += 64) { // half rows longer than one wave: boundary-free interior rows never are
                T bt[9];
#pragma unroll
                for (int e = 0; e < 9; ++e) bt[e] = val[((int64_t)i * 125 + k) * 9 + e];
                entry(k, col[(int64_t)i * 125 + k], bt);
            }
            s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
            if (lane == 0) gs_store_rhs<T>(sv, ii, di, rh0 - s0, rh1 - s1, rh2 - s2);
        }
    }
    __syncthreads();
    if (w == 0 && !GS_DBG(1)) gs_phase_b<T, FWD, SB>(tri, sv, nodes, cnt, lane, diagVal, diagBlockInv, x, hD);
    }
}

// ---------------- phase B of the block GS kernels: lane = row, executed by one wavefront.  WT: publish x with
// write-through (sc1) stores so that other workgroups of the same launch can read it with sc1 loads
template <class T, bool FWD, int SB, bool WT>
__device__ __forceinline__ void gs_phase_b(const T* tri, const T* sv, const int32_t* nodes, int cnt, int lane, const T* __restrict__ diagVal, const T* __restrict__ diagBlockInv, T* x, T* hD,
    const T* ldsD, T* ldsX)
{
    constexpr int TRI = GsLds<T, SB>::TRI;
    const int me = lane;
    const int i = me < cnt ? nodes[me] : -1;
    T a0 = me < cnt ? sv[me * 3] : (T)0, a1 = me < cnt ? sv[me * 3 + 1] : (T)0, a2 = me < cnt ? sv[me * 3 + 2] : (T)0;
    // column `cidx` of the triangle for this lane's row (zero where the row does not follow the column); the next
    // column is fetched from LDS while the current step's dependent arithmetic runs
    auto load_col = [&](int cidx, T (&L)[9]) {
        bool act = FWD ? (me > cidx && me < cnt) : (me < cidx);
        int idx = act ? (FWD ? gs_tri_fwd<SB>(me, cidx) : gs_tri_bwd(me, cidx)) : TRI - 1; // masked lanes read the zero entry
#pragma unroll
        for (int e = 0; e < 9; ++e) L[e] = tri[e * TRI + idx];
    };
    // step: row cidx is final (every earlier column has been applied); broadcast it and apply its column
    auto step = [&](int cidx, const T (&L)[9]) {
        T b0 = lane_bcast(a0, cidx), b1 = lane_bcast(a1, cidx), b2 = lane_bcast(a2, cidx); // v_readlane: cidx is wave-uniform
        a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
        a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
        a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
    };
    auto colof = [&](int s) { return FWD ? s : cnt - 1 - s; };
    T LA[9], LB[9];
    if (cnt > 0) load_col(colof(0), LA);
    int s = 0;
    for (; s + 1 < cnt; s += 2) { // two steps per trip: the column buffers alternate without register copies
        load_col(colof(s + 1), LB);
        step(colof(s), LA);
        load_col(colof(min(s + 2, cnt - 1)), LA); // unconditional (clamped): a conditional load makes the compiler copy the buffers
        step(colof(s + 1), LB);
    }
    if (s < cnt) step(colof(s), LA);
    const T h0 = a0, h1 = a1, h2 = a2;
    T dd[9]; // D_i for hD = D h (forward sweep only); loaded here, not before the loop, to keep the kernel under 80
             // VGPRs (three 512-thread workgroups per CU, i.e. one round per launch on the finest level)
    if (FWD) {
#pragma unroll
        for (int e = 0; e < 9; ++e) dd[e] = i >= 0 ? (ldsD ? ldsD[9 * me + e] : diagVal[9 * (int64_t)i + e]) : (T)0; // ldsD: staged by the caller
    }
    if (ldsX && i >= 0) ldsX[3 * me] = h0, ldsX[3 * me + 1] = h1, ldsX[3 * me + 2] = h2; // k_gs_block2: the block's other sub-block reads these instead of global memory
    if (i >= 0) {
        if (WT) {
            __hip_atomic_store(x + 3 * (int64_t)i, h0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(x + 3 * (int64_t)i + 1, h1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(x + 3 * (int64_t)i + 2, h2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else
            x[3 * (int64_t)i] = h0, x[3 * (int64_t)i + 1] = h1, x[3 * (int64_t)i + 2] = h2;
        if (FWD) {
            hD[3 * (int64_t)i] = dd[0] * h0 + dd[3] * h1 + dd[6] * h2;
            hD[3 * (int64_t)i + 1] = dd[1] * h0 + dd[4] * h1 + dd[7] * h2;
            hD[3 * (int64_t)i + 2] = dd[2] * h0 + dd[5] * h1 + dd[8] * h2;
        }
        else if (hD) { // backward: hD is the iterate u, which takes the correction here (u += du of gs_smooth) instead of in an axpy launch
            hD[3 * (int64_t)i] += h0, hD[3 * (int64_t)i + 1] += h1, hD[3 * (int64_t)i + 2] += h2;
        }
    }
}

// ---------------- the finest-level colour pass as two kernels (levels prepared by k_gs_images, mg_build.hip)
// Measured on k_gs_block / k_gs_block2 (per-phase timestamps, C2): a colour launch lasts as long as its slowest workgroup, an interior
// block, which walks a ~45 us chain of dependent round trips (header -> columns -> values -> gathers -> D^-1 -> sums, twice per
// sub-block) even when it has nothing but its in-block couplings to read (first colour), and up to 45 us more where the off-block
// half rows are long (last colour) — while half of the workgroups (surface blocks) have long finished and HBM idles.  So:
//   k_gs_offblock  the off-block products of the colour's rows, summed per SLOT (a run of up to 16 stored entries of one row; four slots per
//                  wavefront step, below): no serial part, streams like k_gs_residual, balanced whatever the body; k_gs_subst subtracts a
//                  row's slot sums from its right-hand side;
//   k_gs_subst     one wavefront per colour block: h = D^-1 p1 + (strict in-block triangle of -(D^-1 A)) h by substitution (below).
template <class T>
__device__ __forceinline__ T row16_sum(T v) // sum over each 16-lane DPP row; valid in the row's lane 15
{
    v += dpp_move<0xb1, 0xf>(v);
    v += dpp_move<0x4e, 0xf>(v);
    v += dpp_move<0x114, 0xf>(v);
    v += dpp_move<0x118, 0xf>(v);
    return v;
}
template <class T>
struct GsOffItem { // what a 16-lane group has in flight for its slot between the value loads and the sums
    int j;
    bool valid;
    T bv[9];
};
// Four slots (runs of up to 16 stored off-block entries of one row, k_gs_slot_fill) per wavefront and step, one per 16-lane group, as a
// software pipeline: the slot descriptors of step n+2 (scalar cache), the column ids + values of step n+1 and the gathers of step n are
// in flight together — the descriptor -> values -> gathers chain of dependent round trips is paid once per wavefront, not per row.
// Why slots: with one wavefront per row (rows of 3..98 entries) more than half of the lanes of every load carry nothing, and the kernel
// is bound by the load instructions a compute unit can retire (measured: the same time with 4096 or 16384 wavefronts resident), not by HBM.
// Branch-free on purpose (see k_gs_subst): a lane past the slot's end reads the slot's first entry and its product is dropped by a
// select — a load under a branch, even a wave-uniform one, is a join at which the compiler waits for ALL loads in flight.
// Sums: fixed-order DPP tree over the group's 16 lanes (wave_sum's first four additions); the group's lane 15 stores the slot's three sums,
// k_gs_subst subtracts a row's slots from its right-hand side in slot order.
// The pipeline itself: steps w, w + W, ... below nstep of the slot range [s_begin, s_end); store(slot, s0, s1, s2) takes a slot's three sums.
template <class T, class Store>
__device__ __forceinline__ void gs_off_steps(const int2* __restrict__ slot, const int32_t* __restrict__ gcol, const T* __restrict__ val, const T* x, int s_begin, int s_end, int w, int W, int nstep, Store store)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, l16 = lane & 15;
    if (w >= nstep) return;
    auto descriptor = [&](int n) __attribute__((always_inline)) { // of this lane's group (past the colour's last slot: the last slot's again)
        return slot[min(s_begin + 4 * min(n, nstep - 1) + g, s_end - 1)];
    };
    auto values = [&](int n, const int2 d, GsOffItem<T>& R) __attribute__((always_inline)) {
        R.valid = n < nstep && s_begin + 4 * n + g < s_end && l16 < d.y;
        const int64_t e = (int64_t)d.x + (R.valid ? l16 : 0);
        R.j = nt_load(gcol + e);
        const T* bb = val + e * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) R.bv[t] = nt_load(bb + t);
    };
    auto finish = [&](int n, const GsOffItem<T>& R, int nn, const int2 dn, GsOffItem<T>& N, int nd, int2& dd) __attribute__((always_inline)) {
        const int64_t jj = R.j; // (a dropped lane: the column of the slot's first entry)
        const T x0 = x[3 * jj], x1 = x[3 * jj + 1], x2 = x[3 * jj + 2];
        asm volatile("" ::: "memory"); // gathers first, then the next step's values and the descriptor after that: the sums below wait for the former only
        values(nn, dn, N);
        dd = descriptor(nd);
        asm volatile("" ::: "memory"); // issued HERE, a step (two steps) ahead of their use
        const T(&b)[9] = R.bv;
        T s[3] = { b[0] * x0 + b[3] * x1 + b[6] * x2, b[1] * x0 + b[4] * x1 + b[7] * x2, b[2] * x0 + b[5] * x1 + b[8] * x2 };
#pragma unroll
        for (int d = 0; d < 3; ++d) s[d] = row16_sum(R.valid ? s[d] : (T)0);
        const int sl = s_begin + 4 * n + g;
        if (l16 == 15 && n < nstep && sl < s_end) store(sl, s[0], s[1], s[2]);
    };
    GsOffItem<T> A, B;
    int2 d1 = descriptor(w + W), d2;
    values(w, descriptor(w), A);
    d2 = descriptor(w + 2 * W);
    asm volatile("" ::: "memory");
    for (int n = w; n < nstep; n += 2 * W) {
        int2 d3, d4;
        finish(n, A, n + W, d1, B, n + 3 * W, d3);
        finish(n + W, B, n + 2 * W, d2, A, n + 4 * W, d4); // (an odd number of steps: one step past the end, computed from the last step's descriptor and not stored)
        d1 = d3, d2 = d4;
    }
}
// The streaming role: workgroup `bid` of `nwg` (256 threads each) over the slots [s_begin, s_end), sums to part[3 slot ..].
template <class T>
__device__ __forceinline__ void gs_off_stream(const int2* __restrict__ slot, const int32_t* __restrict__ gcol, const T* __restrict__ val, const T* x, T* part, int s_begin, int s_end, int bid, int nwg)
{
    // Steps are dealt to the XCDs in eight contiguous runs (workgroup b runs on XCD b % 8 — observed placement, used for speed only; any
    // placement gives the same sums): slots follow the colour's blocks in first-touch (page) order, so an XCD's run gathers x from one
    // region of the grid and that part of x stays in ITS L2.  Dealt round robin, every XCD pulled all of x through its own L2 in every
    // launch: measured 155 MB of fabric reads per launch against 109 MB algorithmic (profiles/r04_pmc_summary.json), the difference being
    // eight copies of x (C2: 6.5 MB each).
    const int nstep_all = (s_end - s_begin + 3) >> 2;
    const bool by_xcd = (nwg & 7) == 0;
    const int chunk = by_xcd ? (nstep_all + 7) >> 3 : nstep_all, xcd = by_xcd ? (bid & 7) : 0;
    const int W = by_xcd ? (nwg >> 3) * 4 : nwg * 4;
    const int nstep = min(nstep_all, (xcd + 1) * chunk); // end of this XCD's run of steps
    const int w = __builtin_amdgcn_readfirstlane(xcd * chunk + (int)((by_xcd ? bid >> 3 : bid) * 4 + (threadIdx.x >> 6))); // wave-uniform, and known to be: descriptors come through the scalar cache
    gs_off_steps<T>(slot, gcol, val, x, s_begin, s_end, w, W, nstep, [&](int sl, T s0, T s1, T s2) __attribute__((always_inline)) {
        T* o = part + 3 * (int64_t)sl;
        o[0] = s0, o[1] = s1, o[2] = s2;
    });
}
template <class T>
__global__ __launch_bounds__(256) void k_gs_offblock(const int2* __restrict__ slot, const int32_t* __restrict__ gcol, const T* __restrict__ val, const int32_t* __restrict__ gs_pad,
    const T* __restrict__ x, T* __restrict__ part, int s_begin, int s_end /*the colour's slots of this sweep direction (Level::gs_slot_start: known to the host since the build)*/)
{
    gs_off_stream<T>(slot, gcol, val, x, part, s_begin, s_end, (int)blockIdx.x, (int)gridDim.x);
}

// The block's 64-row triangular solve, one wavefront per colour block, lane = row, NO LDS: column c of the premultiplied in-block image
// (GsImg: entries packed column by column in sweep order, rows ascending inside a column) is fetched straight into registers D steps
// before the substitution reaches it — the loads depend on nothing but the block id and the masks, so the only chain left is the
// substitution's own (broadcast of the finished row, nine multiply-adds) instead of an LDS round trip per step plus a dependent-load
// prologue.  One wavefront runs alone on its SIMD, so what a step costs is its instruction count (every dependent instruction pays the
// full pipeline latency): the loop is kept branch-free and lean — the column masks sit in a register pair (lane c = column c), a lane's
// entry index is column offset + v_mbcnt of the mask, a lane without an entry in the column reads the image's all-zero entry 0 instead of
// being masked out of the multiply-adds.  (Measured, C2, per launch: 28 us with the entries stored as nine coalesced planes — more address arithmetic —, 21 us as below; steps
// 32..64 of a block take 150 ns each whether 8 or 16 columns are in flight: at 4.4 TB/s over the 729 blocks of a colour the kernel is
// bound by HBM, not by its chain any more.)
template <class T, bool FWD, int D>
__global__ __launch_bounds__(64) void k_gs_subst(const T* __restrict__ img, const uint16_t* __restrict__ imgi, const int32_t* __restrict__ gs_pad, const T* __restrict__ part, T* x,
    T* hD, int block0, const T* __restrict__ rhs, T* hsub /*backward, or null: the forward sweep's h, which becomes h - du row by row for k_gs_residual<T, true>*/)
{
    using I = GsImg<T>;
    const int lane = threadIdx.x;
    const int b = block0 + blockIdx.x;
    const T* hdr = img + (size_t)b * I::per_block;
    const T* ent = hdr + I::hdr_elems + (FWD ? 0 : I::per_dir);
    const int64_t pos = (int64_t)b * 64 + lane;
    const int32_t* rec = gs_pad + 8 * pos;
    const int node = rec[0], nslot = ((FWD ? rec[1] : rec[4]) + 15) >> 4, slot0 = rec[FWD ? 5 : 6]; // the row's off-block slots of this direction (none in the first colour of a half sweep)
    // this row's entry index at each of the 64 steps, 16 bits each: 128 bytes per lane, one round trip before the first entry load
    uint32_t iw[32];
    {
        const uint4* ip = (const uint4*)(imgi + ((size_t)b * 2 + (FWD ? 0 : 1)) * I::idx_per_dir + lane * 64);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 v = ip[q];
            iw[4 * q] = v.x, iw[4 * q + 1] = v.y, iw[4 * q + 2] = v.z, iw[4 * q + 3] = v.w;
        }
    }
    T ring[D][9];
    // The 64 steps are unrolled, so a step's index is a fixed half of a fixed register and its loads are a multiply and five loads off one
    // base (24 instructions a step; 50 with the column masks + v_mbcnt ranks of the first version).  A lane without an entry in the column
    // reads the all-zero entry 0 instead of being masked out (a load under a divergent branch makes the compiler wait for EVERY outstanding
    // load at the join).
#define HOT_GS_ISSUE(s, L)                                                                   \
    do {                                                                                      \
        const uint32_t idx_ = (iw[(s) >> 1] >> (16 * ((s)&1))) & 0xffffu;                     \
        const T* p_ = ent + (size_t)idx_ * 9;                                                 \
        _Pragma("unroll") for (int e_ = 0; e_ < 9; ++e_) L[e_] = p_[e_];                      \
        asm volatile("" ::: "memory"); /* the loads stay HERE, D steps ahead of their use */ \
    } while (0)
#pragma unroll
    for (int k = 0; k < D; ++k) HOT_GS_ISSUE(k, ring[k]);
    // a = D^-1 p1, p1 = rhs - the row's off-block products: the sums of its slots (k_gs_offblock) in slot order; D^-1 is stored by position
    T a0, a1, a2;
    {
        const T* src = rhs + 3 * (int64_t)max(node, 0);
        T q0 = src[0], q1 = src[1], q2 = src[2];
        T ps[8][3]; // (a row has at most 124 off-block columns: eight slots; branch-free: past the row's last slot its first one — or the padding — is read and dropped)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const T* pp = part + 3 * (int64_t)(slot0 + (q < nslot ? q : 0));
            ps[q][0] = pp[0], ps[q][1] = pp[1], ps[q][2] = pp[2];
        }
        T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s0 += q < nslot ? ps[q][0] : (T)0, s1 += q < nslot ? ps[q][1] : (T)0, s2 += q < nslot ? ps[q][2] : (T)0;
        q0 -= s0, q1 -= s1, q2 -= s2;
        const T* di = hdr + 576 + 9 * lane;
        a0 = di[0] * q0 + di[3] * q1 + di[6] * q2, a1 = di[1] * q0 + di[4] * q1 + di[7] * q2, a2 = di[2] * q0 + di[5] * q1 + di[8] * q2; // gs_store_rhs's product
    }
    T dd[9];
    if (FWD) {
#pragma unroll
        for (int e = 0; e < 9; ++e) dd[e] = hdr[9 * lane + e];
    }
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const int c = FWD ? s : 63 - s;
        const T b0 = lane_bcast(a0, c), b1 = lane_bcast(a1, c), b2 = lane_bcast(a2, c);
        T(&L)[9] = ring[s % D];
        a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
        a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
        a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
        if (s + D < 64) HOT_GS_ISSUE(s + D, L);
    }
#undef HOT_GS_ISSUE
    if (node < 0) return;
    x[3 * (int64_t)node] = a0, x[3 * (int64_t)node + 1] = a1, x[3 * (int64_t)node + 2] = a2;
    if (FWD) {
        hD[3 * (int64_t)node] = dd[0] * a0 + dd[3] * a1 + dd[6] * a2;
        hD[3 * (int64_t)node + 1] = dd[1] * a0 + dd[4] * a1 + dd[7] * a2;
        hD[3 * (int64_t)node + 2] = dd[2] * a0 + dd[5] * a1 + dd[8] * a2;
    }
    else {
        if (hD) hD[3 * (int64_t)node] += a0, hD[3 * (int64_t)node + 1] += a1, hD[3 * (int64_t)node + 2] += a2; // backward: hD is the iterate u, which takes the correction here (u += du of gs_smooth)
        if (hsub) hsub[3 * (int64_t)node] -= a0, hsub[3 * (int64_t)node + 1] -= a1, hsub[3 * (int64_t)node + 2] -= a2; // (nothing in the backward sweep reads h)
    }
}

// ---------------- one rank, finest levels: the colour pass as ONE launch with two roles (levels prepared with the four slot lists of k_gs_slot_fill2)
// A row's off-block columns of a half sweep are of two ages: those of the colour swept just before the row's own, which are final when that
// colour's launch ends, and older ones, final one launch earlier.  So the launch of colour c runs, side by side and independent of one another,
//   (a) workgroups [0, nb): one per block of colour c.  Wavefronts 1 .. 3 sum the slots of the block's rows that read the PREVIOUS colour
//       (gs_off_steps, sums to LDS) while wavefront 0 walks the dependent loads at the head of the substitution (index table, first image
//       columns, record, right-hand side, the older slots' sums from the launch before); one barrier; wavefront 0 then substitutes as
//       k_gs_subst does.  The previous-colour sums never see memory and cost the substitution nothing: they land before its own prologue does;
//   (b) workgroups [nb_pad, grid): the OLDER slots of the NEXT colour, streamed as k_gs_offblock does (they read nothing this launch writes).
// The 729 substitution wavefronts of a C2 colour (< 1 per SIMD, a 64-step dependent chain each) no longer own the chip alone, and a half
// sweep is 8 launches instead of 15.  Row sums: previous-colour slots first, then the older ones, each in slot order (the pair path cuts the
// concatenated run into slots instead: equal to rounding).
// TURN (forward only, the LAST colour of the forward sweep): the backward sweep starts with the same colour, whose rows have no following off-block
// column — a block's backward substitution needs nothing but its own forward result (right-hand side D h of its own rows).  The wavefront runs it right
// behind the forward one (backward index table fetched with the forward one, backward image columns requested into the ring slots the forward walk frees),
// and the backward sweep's first launch — 729 wavefronts walking a dependent chain with the chip otherwise empty — does not happen.  Same arithmetic as the
// two launches: bit-identical results.
// Development aid (-DHOT_GSC_CLOCKS, tools/gs_colour_phases.sh): the 100 MHz clock of every colour block's substitution wavefront at its phase boundaries
// (prologue round trips 1 + 2, image requests, wait for the previous colour's sums, the 64 steps, the stores), of its first summing wavefront at the barrier,
// and the start / end of every block and streaming workgroup — plain stores per workgroup (atomics on shared words cost more than the kernel), summed per
// pass of the symmetric sweep by k_gsc_pass behind each launch; smooth_dev prints the table every ten sweeps (profiles/r06_gs_colour_clocks.txt).
#ifdef HOT_GSC_CLOCKS
__device__ unsigned long long gsc_clk[16][12]; // [pass of the symmetric sweep: forward q | 8 + backward q][0 blocks, 1 - 5 phases, 6 block workgroups' span, 7 summing wavefront, 8 streaming workgroups' span, 9 launches]
__device__ unsigned long long gsc_blk[4096][8]; // of the launch in flight, per block workgroup: five phases, summing wavefront done, start, end (plain stores: atomics on one word per block cost more than the kernel)
__device__ unsigned long long gsc_str[4096][2]; // per streaming workgroup: start, end
__global__ void k_gsc_pass(int p, int nb, int ns) // behind a launch: its workgroups' clocks to the sums of its pass
{
    __shared__ unsigned long long red[8][256], lo[2][256], hi[2][256];
    const int t = threadIdx.x;
    unsigned long long a[8] = {}, l0 = ~0ull, h0 = 0, l1 = ~0ull, h1 = 0;
    for (int i = t; i < nb && i < 4096; i += 256) {
        for (int k = 0; k < 6; ++k) a[k] += gsc_blk[i][k];
        l0 = min(l0, gsc_blk[i][6]), h0 = max(h0, gsc_blk[i][7]);
    }
    for (int i = t; i < ns && i < 4096; i += 256) l1 = min(l1, gsc_str[i][0]), h1 = max(h1, gsc_str[i][1]);
    for (int k = 0; k < 6; ++k) red[k][t] = a[k];
    lo[0][t] = l0, hi[0][t] = h0, lo[1][t] = l1, hi[1][t] = h1;
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < 256; ++i) {
            for (int k = 0; k < 6; ++k) red[k][0] += red[k][i];
            lo[0][0] = min(lo[0][0], lo[0][i]), hi[0][0] = max(hi[0][0], hi[0][i]), lo[1][0] = min(lo[1][0], lo[1][i]), hi[1][0] = max(hi[1][0], hi[1][i]);
        }
        unsigned long long* c = gsc_clk[p];
        c[0] += (unsigned long long)nb;
        for (int k = 0; k < 5; ++k) c[1 + k] += red[k][0];
        c[7] += red[5][0];
        if (nb > 0 && hi[0][0] > lo[0][0]) c[6] += hi[0][0] - lo[0][0];
        if (ns > 0 && hi[1][0] > lo[1][0]) c[8] += hi[1][0] - lo[1][0];
        c[9] += 1;
    }
}
#define HOT_GS_CLK(i)                                  \
    do {                                               \
        asm volatile("" ::: "memory");                 \
        const unsigned long long t_ = wall_clock64();  \
        clk_[i] = t_ - tl_, tl_ = t_;                  \
        asm volatile("" ::: "memory");                 \
    } while (0)
#else
#define HOT_GS_CLK(i)
#endif
template <class T, bool FWD, int D, bool TURN = false>
__global__ __launch_bounds__(256) void k_gs_colour(const T* __restrict__ img, const uint16_t* __restrict__ imgi, const int32_t* __restrict__ gs_pad, const int4* __restrict__ srec, T* x, T* hD, int block0,
    int nb, int nb_pad /*nb rounded up to a multiple of 8: the streaming workgroups keep their XCD (workgroup id % 8)*/, const T* __restrict__ rhs, T* hsub, const int2* __restrict__ slot,
    const int32_t* __restrict__ gcol, const T* __restrict__ val, T* part, int s_begin, int s_end /*streaming role: the next colour's older slots*/,
    T* xb /*TURN: the backward sweep's target*/, T* ub /*TURN: the iterate, which takes the correction (or null)*/)
{
    static_assert(!TURN || FWD, "the turn is the end of the forward sweep");
#ifdef HOT_GSC_CLOCKS
    const unsigned long long t00_ = wall_clock64();
    unsigned long long tl_ = t00_, clk_[6] = {};
#endif
    if ((int)blockIdx.x >= nb_pad) {
        gs_off_stream<T>(slot, gcol, val, x, part, s_begin, s_end, (int)blockIdx.x - nb_pad, (int)gridDim.x - nb_pad);
#ifdef HOT_GSC_CLOCKS
        const int sid_ = (int)blockIdx.x - nb_pad;
        if (threadIdx.x == 0 && sid_ < 4096) gsc_str[sid_][0] = t00_, gsc_str[sid_][1] = wall_clock64();
#endif
        return;
    }
    if ((int)blockIdx.x >= nb) return;
    using I = GsImg<T>;
    __shared__ T lprev[3 * 512]; // sums of the block's previous-colour slots (a row has at most 124 off-block columns: eight slots)
    __shared__ T ldv[9 * 64]; // D^-1 by position, [entry][position]
    __shared__ uint32_t lidx[(TURN ? 2 : 1) * 64 * 33]; // wavefront 0: its lanes' rows of the index table (33 words a row: lane l reads bank (33 l + s / 2) % 64)
    const int b = block0 + blockIdx.x;
    const int p0 = FWD ? srec[(int64_t)b * 64].y : srec[(int64_t)b * 64].w, p1 = FWD ? srec[(int64_t)b * 64 + 64].y : srec[(int64_t)b * 64 + 64].w; // the block's previous-colour slots
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T* hdr = img + (size_t)b * I::per_block;
    if (wave != 0) { // the previous colour's share of the block's row sums, to LDS
        gs_off_steps<T>(slot, gcol, val, x, p0, p1, wave - 1, 3, (p1 - p0 + 3) >> 2, [&](int sl, T s0, T s1, T s2) __attribute__((always_inline)) {
            T* o = lprev + 3 * (sl - p0);
            o[0] = s0, o[1] = s1, o[2] = s2;
        });
#ifdef HOT_GSC_CLOCKS
        if (threadIdx.x == 64 && blockIdx.x < 4096) gsc_blk[blockIdx.x][5] = wall_clock64() - t00_;
#endif
        __syncthreads();
        return;
    }
    // wavefront 0: the substitution.  A lane's row of the index table goes through LDS (its own 132 bytes: no barrier), not through 32 registers.
    // While the other wavefronts sum the previous colour's slots it walks its own two dependent round trips: (1) index table, position record, slot
    // starts; (2) right-hand side, the older slots' sums (the launch before), D^-1 -> rhs - older sums in six registers, D^-1 parked in LDS; then
    // the first D image columns are requested and land under the wait for the barrier.
    const T* ent = hdr + I::hdr_elems + (FWD ? 0 : I::per_dir);
    const int64_t pos = (int64_t)b * 64 + lane;
    const int4 r0 = *(const int4*)(gs_pad + 8 * pos), r1 = *(const int4*)(gs_pad + 8 * pos + 4), sr = srec[pos];
    {
        const uint4* ip = (const uint4*)(imgi + ((size_t)b * 2 + (FWD ? 0 : 1)) * I::idx_per_dir + lane * 64);
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ip[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) lidx[33 * lane + 4 * q] = v[q].x, lidx[33 * lane + 4 * q + 1] = v[q].y, lidx[33 * lane + 4 * q + 2] = v[q].z, lidx[33 * lane + 4 * q + 3] = v[q].w;
        if (TURN) {
            const uint4* ipb = (const uint4*)(imgi + ((size_t)b * 2 + 1) * I::idx_per_dir + lane * 64);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ipb[q];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                lidx[64 * 33 + 33 * lane + 4 * q] = v[q].x, lidx[64 * 33 + 33 * lane + 4 * q + 1] = v[q].y, lidx[64 * 33 + 33 * lane + 4 * q + 2] = v[q].z, lidx[64 * 33 + 33 * lane + 4 * q + 3] = v[q].w;
        }
    }
    const int node = r0.x, nall = FWD ? r0.y : r1.x, nprev_e = FWD ? (r1.w & 0xffff) : ((r1.w >> 16) & 0xffff);
    const int nold = (nall - nprev_e + 15) >> 4, nprev = (nprev_e + 15) >> 4, so = FWD ? sr.x : sr.z, sp = (FWD ? sr.y : sr.w) - p0;
    T q0, q1, q2;
    {
        const T* src = rhs + 3 * (int64_t)max(node, 0);
        q0 = src[0], q1 = src[1], q2 = src[2];
        T ps[8][3]; // branch-free: past the row's last slot its first one — or the padding — is read and dropped
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const T* pp = part + 3 * (int64_t)(so + (q < nold ? q : 0));
            ps[q][0] = pp[0], ps[q][1] = pp[1], ps[q][2] = pp[2];
        }
        const T* di = hdr + 576 + 9 * lane;
#pragma unroll
        for (int e = 0; e < 9; ++e) ldv[64 * e + lane] = di[e];
        T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s0 += q < nold ? ps[q][0] : (T)0, s1 += q < nold ? ps[q][1] : (T)0, s2 += q < nold ? ps[q][2] : (T)0;
        q0 -= s0, q1 -= s1, q2 -= s2;
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2)::"memory"); // the image columns are requested BEHIND these sums (the asm consumes them): 72 registers of row data and 144 of columns never live together
    }
    HOT_GS_CLK(0);
    const uint16_t* lrow = (const uint16_t*)(lidx + 33 * lane);
    T ring[D][9];
#define HOT_GS_ISSUE(E, LR, s, L)                                                            \
    do {                                                                                      \
        const uint32_t idx_ = (LR)[s];                                                        \
        const T* p_ = (E) + (size_t)idx_ * 9;                                                 \
        _Pragma("unroll") for (int e_ = 0; e_ < 9; ++e_) L[e_] = nt_load(p_ + e_);            \
        asm volatile("" ::: "memory"); /* the loads stay HERE, D steps ahead of their use */ \
    } while (0)
#pragma unroll
    for (int k = 0; k < D; ++k) HOT_GS_ISSUE(ent, lrow, k, ring[k]);
    HOT_GS_CLK(1);
    __syncthreads(); // the previous colour's share of the row sums is in LDS
    HOT_GS_CLK(2);
    T a0, a1, a2;
    {
        T s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const T* pp = lprev + 3 * (q < nprev ? sp + q : 0);
            const T v0 = pp[0], v1 = pp[1], v2 = pp[2];
            s0 += q < nprev ? v0 : (T)0, s1 += q < nprev ? v1 : (T)0, s2 += q < nprev ? v2 : (T)0;
            if (q & 1) asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2)); // two slots' reads in flight, not eight: their 48 registers would come on top of the 144 of the image columns
        }
        q0 -= s0, q1 -= s1, q2 -= s2; // rhs - older slots - previous-colour slots, each run in slot order
        // gs_store_rhs's product D^-1 q, one row of D^-1 at a time (six registers of it beside the image columns, not eighteen)
        a0 = ldv[lane] * q0 + ldv[64 * 3 + lane] * q1 + ldv[64 * 6 + lane] * q2;
        asm volatile("" : "+v"(a0));
        a1 = ldv[64 + lane] * q0 + ldv[64 * 4 + lane] * q1 + ldv[64 * 7 + lane] * q2;
        asm volatile("" : "+v"(a1));
        a2 = ldv[64 * 2 + lane] * q0 + ldv[64 * 5 + lane] * q1 + ldv[64 * 8 + lane] * q2;
    }
    T dd[9];
    const T* entb = hdr + I::hdr_elems + I::per_dir;
    const uint16_t* lrowb = (const uint16_t*)(lidx + 64 * 33 + 33 * lane);
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const int c = FWD ? s : 63 - s;
        const T b0 = lane_bcast(a0, c), b1 = lane_bcast(a1, c), b2 = lane_bcast(a2, c);
        T(&L)[9] = ring[s % D];
        a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
        a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
        a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
        if (s + D < 64) HOT_GS_ISSUE(ent, lrow, s + D, L);
        else if (TURN) HOT_GS_ISSUE(entb, lrowb, s + D - 64, L); // the backward walk's first columns, into the slots the forward walk no longer needs
        if (FWD && s + D == 64) { // D by position (for hD = D h)
#pragma unroll
            for (int e = 0; e < 9; ++e) dd[e] = hdr[9 * lane + e];
            asm volatile("" ::: "memory");
        }
    }
#ifdef HOT_GSC_CLOCKS
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2)::"memory");
    HOT_GS_CLK(3);
    auto clk_out = [&]() __attribute__((always_inline)) {
        HOT_GS_CLK(4);
        if (lane == 0) {
            if (blockIdx.x < 4096) {
                for (int i = 0; i < 5; ++i) gsc_blk[blockIdx.x][i] = clk_[i];
                gsc_blk[blockIdx.x][6] = t00_, gsc_blk[blockIdx.x][7] = tl_;
            }
        }
    };
#endif
    if (!TURN && node < 0) return;
    T h0 = 0, h1 = 0, h2 = 0;
    if (FWD) {
        h0 = dd[0] * a0 + dd[3] * a1 + dd[6] * a2, h1 = dd[1] * a0 + dd[4] * a1 + dd[7] * a2, h2 = dd[2] * a0 + dd[5] * a1 + dd[8] * a2;
        if (node >= 0) {
            if (!TURN) x[3 * (int64_t)node] = a0, x[3 * (int64_t)node + 1] = a1, x[3 * (int64_t)node + 2] = a2; // (TURN: written below as h - du, or as h)
            hD[3 * (int64_t)node] = h0, hD[3 * (int64_t)node + 1] = h1, hD[3 * (int64_t)node + 2] = h2;
        }
    }
    else {
        x[3 * (int64_t)node] = a0, x[3 * (int64_t)node + 1] = a1, x[3 * (int64_t)node + 2] = a2;
        if (hD) hD[3 * (int64_t)node] += a0, hD[3 * (int64_t)node + 1] += a1, hD[3 * (int64_t)node + 2] += a2; // backward: hD is the iterate u, which takes the correction here (u += du of gs_smooth)
        if (hsub) hsub[3 * (int64_t)node] -= a0, hsub[3 * (int64_t)node + 1] -= a1, hsub[3 * (int64_t)node + 2] -= a2; // (nothing in the backward sweep reads h)
    }
    if constexpr (TURN) {
        // the block's backward substitution: right-hand side D h of its own rows (no following off-block column exists), du = D^-1 (D h) + the strictly upper
        // in-block triangle, exactly what the backward sweep's first launch would compute from the stored D h
        const T f0 = a0, f1 = a1, f2 = a2; // h
        a0 = ldv[lane] * h0 + ldv[64 * 3 + lane] * h1 + ldv[64 * 6 + lane] * h2;
        asm volatile("" : "+v"(a0));
        a1 = ldv[64 + lane] * h0 + ldv[64 * 4 + lane] * h1 + ldv[64 * 7 + lane] * h2;
        asm volatile("" : "+v"(a1));
        a2 = ldv[64 * 2 + lane] * h0 + ldv[64 * 5 + lane] * h1 + ldv[64 * 8 + lane] * h2;
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const int c = 63 - s;
            const T b0 = lane_bcast(a0, c), b1 = lane_bcast(a1, c), b2 = lane_bcast(a2, c);
            T(&L)[9] = ring[(64 + s) % D]; // the ring keeps turning: the forward walk's step 64 - D + k requested the backward walk's column k into slot (64 - D + k) % D
            a0 = fma(L[0], b0, a0), a1 = fma(L[1], b0, a1), a2 = fma(L[2], b0, a2);
            a0 = fma(L[3], b1, a0), a1 = fma(L[4], b1, a1), a2 = fma(L[5], b1, a2);
            a0 = fma(L[6], b2, a0), a1 = fma(L[7], b2, a1), a2 = fma(L[8], b2, a2);
            if (s + D < 64) HOT_GS_ISSUE(entb, lrowb, s + D, L);
        }
        if (node < 0) return;
        xb[3 * (int64_t)node] = a0, xb[3 * (int64_t)node + 1] = a1, xb[3 * (int64_t)node + 2] = a2;
        if (ub) ub[3 * (int64_t)node] += a0, ub[3 * (int64_t)node + 1] += a1, ub[3 * (int64_t)node + 2] += a2;
        // the forward target: h - du where the residual wants it (hsub), h otherwise
        if (hsub)
            x[3 * (int64_t)node] = f0 - a0, x[3 * (int64_t)node + 1] = f1 - a1, x[3 * (int64_t)node + 2] = f2 - a2;
        else
            x[3 * (int64_t)node] = f0, x[3 * (int64_t)node + 1] = f1, x[3 * (int64_t)node + 2] = f2;
    }
#ifdef HOT_GSC_CLOCKS
    if (!TURN) clk_out(); // (the turn's second walk is not clocked: its lanes without a row have left)
#endif
#undef HOT_GS_ISSUE
}

// A whole half sweep (all colours, all sub-blocks) in ONE launch.  Workgroups are ordered by pass = (colour, sub-block)
// in sweep order; a workgroup of pass p
//   1. streams the needed half of its rows into registers and files the in-sub-block couplings into the LDS triangle —
//      none of this depends on the unknowns, so it overlaps with the substitution phase of earlier passes;
//   2. a) makes sure everything older than pass p-1 that it reads is published and folds those columns into the staged
//         right-hand side;  b) waits for what pass p-1 publishes (its adjacent blocks of that colour, or its own block's
//         previous sub-block): point-to-point through gs_flag stamps, or pass counters (HOT_GS_PASS_COUNTERS);
//   3. gathers the remaining columns, reduces the row sums, runs phase B, publishes (write-through stores, then the stamp).
// Progress: workgroups are dispatched in index order (per XCD), so every workgroup a resident one waits for has been
// dispatched before it and waits on nothing itself that is not; the spin is bounded anyway and reports through `err`.
// "not written yet in this half sweep": signalling-NaN payloads that no arithmetic result carries (a computed NaN is the canonical quiet one)
template <class T>
__global__ void k_gs_fill_unset(size_t n, T* x)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (sizeof(T) == 8)
        ((unsigned long long*)x)[i] = GsUnset<double>::bits;
    else
        ((unsigned*)x)[i] = GsUnset<float>::bits;
}
struct GsPasses {
    int npass;
    int wg_begin[34]; // first workgroup of pass p ; wg_begin[npass] = grid size
    int block0[33]; // first colour block of the pass
    int sub[33]; // sub-block index of the pass
    int color[33]; // colour of the pass
};

// Development aid (-DHOT_GS_CLOCKS, tools/gs_phases.sh): shader clocks of wavefront 0 of every workgroup of k_gs_sweep between its phase boundaries,
// summed per pass: [pass][0 header + image copy issued, 1 rows streamed (first barrier), 2 early gathers, 3 wait + late gathers, 4 in-block solve + stores]
#ifdef HOT_GS_CLOCKS
__device__ unsigned long long gs_clk[34 * 8];
#define GS_CLK(i) \
    do { \
        const unsigned long long t_ = clock64(); \
        gclk_[i] += t_ - gt0_, gt0_ = t_; \
    } while (0)
#else
#define GS_CLK(i)
#endif
// WINV (SB = 64): the block's in-block triangular solve is ONE dense product with the precomputed inverse (gs_w: (I - N)^-1 - I of the block and
// direction, row-packed planes, k_gs_winv in mg_build.hip): h = a + W a, a = D^-1 (rhs - off-block products).  The image is copied into the LDS
// area the triangle of in-block couplings occupied (before the wait for the previous pass), every wavefront forms four rows of the product
// (lane = column, fixed-order DPP sums) — 64 dependent broadcast-FMA steps of 150 - 190 ns become one round of ~1 us.
template <class T, bool FWD, int SB, bool WINV = false>
__global__ __launch_bounds__(SB * 16) void k_gs_sweep(const int32_t* __restrict__ col, const T* __restrict__ val, const uint32_t* __restrict__ ckey, const int32_t* __restrict__ gs_order,
    const int32_t* __restrict__ block_start, const T* __restrict__ diagVal, const T* __restrict__ diagBlockInv, const T* __restrict__ rhs, T* x, T* hD, GsPasses P,
    const int32_t* __restrict__ rowcnt, int* done, int* err, const int32_t* __restrict__ nbr, int* flag, int epoch, int dataflag,
    T* unset_next /*not null: the target of the NEXT half sweep (nobody reads it during this one): every workgroup marks its rows' unknowns there "not written yet", instead of a fill launch between the sweeps*/,
    const T* __restrict__ gs_w /*WINV: [block][direction][9][TRI]*/)
{
    extern __shared__ __attribute__((aligned(16))) char gs_smem[];
    constexpr int TRI = GsLds<T, SB>::TRI;
    static_assert(!WINV || SB == 64, "the inverse images are whole-block images");
#ifdef HOT_GS_CLOCKS
    unsigned long long gclk_[5] = { 0, 0, 0, 0, 0 }, gt0_ = clock64();
#endif
    constexpr int RQ = 4, NW = SB / RQ; // rows per wave, waves per workgroup (blockDim.x == 64 * NW)
    // dataflag: the unknowns are their own flags.  The host fills x with a bit pattern no computation produces (GsUnset) before the
    // sweep; a reader of another block's unknown re-loads it until it is something else.  No flag array, no "data, wait for the
    // acknowledgement, flag" on the producer's side and no "flag, then data" round trip on the consumer's: a value is used the
    // moment it lands.  Every node is written exactly once per half sweep, so no stale value can be mistaken for a new one.
    auto ld3 = [&](int64_t j, T& x0, T& x1, T& x2) {
        x0 = __hip_atomic_load(x + 3 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), x1 = __hip_atomic_load(x + 3 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
        x2 = __hip_atomic_load(x + 3 * j + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!dataflag) return;
        int spins = 0;
        while (GsUnset<T>::is(x0) || GsUnset<T>::is(x1) || GsUnset<T>::is(x2)) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 21) || ((spins & 1023) == 0 && *(volatile int*)err)) {
                *(volatile int*)err = 1;
                break;
            }
            x0 = __hip_atomic_load(x + 3 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), x1 = __hip_atomic_load(x + 3 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
            x2 = __hip_atomic_load(x + 3 * j + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    T* tri = (T*)gs_smem; // [9][TRI]
    T* sv = tri + (WINV ? GsWinv<T>::img_elems : 9 * TRI); // [SB][3] (WINV: behind the image's 16-byte padding, which the DMA writes too)
    int32_t* nodes = (int32_t*)(sv + 3 * SB);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int p = 0;
    while (p + 1 < P.npass && (int)blockIdx.x >= P.wg_begin[p + 1]) ++p;
    const int b = P.block0[p] + ((int)blockIdx.x - P.wg_begin[p]);
    const int lo = P.sub[p] * SB;
    const int start = block_start[b] + lo, cnt = max(0, min(SB, block_start[b + 1] - start));
    T* sDinv = (T*)(nodes + 5 * SB); // [SB][9] D_i^-1 and (forward) [SB][9] D_i of the rows: fetched before the wait, so that
    T* sD = sDinv + 9 * SB; // nothing after it has to go to global memory for them
    T* srhs = sD + 9 * SB; // [SB][3] right-hand sides of the rows, likewise
    if (!WINV)
        for (int e = tid; e < 9 * TRI; e += 64 * NW) tri[e] = (T)0;
    if (tid < SB) nodes[tid] = tid < cnt ? gs_order[start + tid] : -1;
    __syncthreads();
    GS_CLK(0);
    if (WINV) {
        // the block's inverse image -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wavefront instruction, lane i lands at base + 16 i, no
        // registers; issued behind the first barrier, which would drain it, so that it travels beside the rows' loads): 142 pieces dealt to the 16 wavefronts
        constexpr int IMG_BYTES = GsWinv<T>::img_elems * (int)sizeof(T);
        const char* wg = (const char*)(gs_w + ((size_t)b * 2 + (FWD ? 0 : 1)) * GsWinv<T>::img_elems);
        for (int c = w; c * 1024 < IMG_BYTES; c += NW) {
            const int off = c * 1024 + lane * 16;
            if (off < IMG_BYTES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + off), (__attribute__((address_space(3))) void*)((char*)tri + c * 1024), 16, 0, 0);
        }
    }
    for (int e = tid; e < 9 * cnt; e += 64 * NW) {
        const int64_t i = nodes[e / 9];
        sDinv[e] = diagBlockInv[9 * i + e % 9];
        if (FWD) sD[e] = diagVal[9 * i + e % 9];
    }
    for (int e = tid; e < 3 * cnt; e += 64 * NW) srhs[e] = rhs[3 * (int64_t)nodes[e / 3] + e % 3];
    if (unset_next)
        for (int e = tid; e < 3 * cnt; e += 64 * NW) gs_store_unset(unset_next + 3 * (int64_t)nodes[e / 3] + e % 3);
    // ---- 1. stream the half rows (lane = slot), keep what couples to nodes outside the sub-block
    T bv[RQ][9];
    int jj[RQ], node[RQ], kb2[RQ], ke[RQ];
    // which 64 slots of a half row stay in registers: the end of the half where the columns of pass p-1 sit.  First sub-block of
    // its colour in sweep order: the previous colour's columns (sorted to the outer end of the half by k_gs_split_rows); a later
    // sub-block: the own block's previous sub-block, i.e. the in-block slots at the inner end.
    const bool first_sub = p == 0 || P.color[p - 1] != P.color[p];
    const bool head = FWD ? first_sub : !first_sub;
    bool late[RQ]; // the column is published by pass p-1: its x is gathered after the wait, every other one before
    const uint32_t prevkey = p > 0 ? ((uint32_t)P.color[p - 1] << 8) | (uint32_t)P.sub[p - 1] : 0xffffffffu;
    if constexpr (WINV) {
        // only the OFF-block half of every row is read (the in-block couplings are inside the image), and the four rows of a wavefront go
        // through the dependent loads together — class counts, then column ids + values, then the columns' colour keys: three round trips per
        // wavefront instead of twelve (measured: the workgroups of the look-ahead passes needed 19 us to stream their 245 KB, longer than the
        // passes in front of them took to finish)
        int4 rc[RQ];
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int ii = w + NW * q;
            jj[q] = -1, kb2[q] = 0, ke[q] = 0, late[q] = false;
            node[q] = ii < cnt ? nodes[ii] : -1;
#pragma unroll
            for (int e = 0; e < 9; ++e) bv[q][e] = (T)0;
        }
#pragma unroll
        for (int q = 0; q < RQ; ++q) rc[q] = node[q] >= 0 ? *(const int4*)(rowcnt + 4 * (int64_t)node[q]) : make_int4(0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            if (node[q] < 0) continue; // wave-uniform
            const int64_t i = node[q];
            const int kbeg = FWD ? 0 : rc[q].x + rc[q].y + 1 + rc[q].z, kend = FWD ? rc[q].x : kbeg + rc[q].w;
            kb2[q] = head ? kbeg + 64 : kbeg, ke[q] = head ? kend : kend - 64;
            const int k = head ? kbeg + lane : kend - 64 + lane;
            if (k >= kbeg && k < kend) {
                jj[q] = col[i * 125 + k];
                const T* bb = val + (i * 125 + k) * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) bv[q][e] = bb[e];
            }
        }
#pragma unroll
        for (int q = 0; q < RQ; ++q)
            if (jj[q] >= 0) {
                const uint32_t keyj = ckey[jj[q]];
                late[q] = (((keyj >> 28) << 8) | (((keyj & 127u) - 1u) / (uint32_t)SB)) == prevkey;
            }
    }
    else
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
        const int ii = w + NW * q;
        jj[q] = -1, node[q] = -1, kb2[q] = 0, ke[q] = 0, late[q] = false;
#pragma unroll
        for (int e = 0; e < 9; ++e) bv[q][e] = (T)0;
        if (ii < cnt) {
            const int i = nodes[ii];
            node[q] = i;
            const int po = rowcnt[4 * i], pi = rowcnt[4 * i + 1], fi = rowcnt[4 * i + 2], fo = rowcnt[4 * i + 3];
            const int kbeg = FWD ? 0 : po + pi + 1, kend = FWD ? po + pi : po + pi + 1 + fi + fo;
            const int ibeg = FWD ? po : kbeg, iend = FWD ? po + pi : kbeg + fi;
            // the 64 slots kept in registers (see `head`); the rest of a longer half row is the "tail"
            kb2[q] = head ? kbeg + 64 : kbeg, ke[q] = head ? kend : kend - 64;
            const int k = head ? kbeg + lane : kend - 64 + lane;
            if (k >= kbeg && k < kend) {
                const int j = col[(int64_t)i * 125 + k];
                const T* bb = val + ((int64_t)i * 125 + k) * 9;
#pragma unroll
                for (int e = 0; e < 9; ++e) bv[q][e] = bb[e];
                jj[q] = j;
                const uint32_t keyj = ckey[j];
                late[q] = (((keyj >> 28) << 8) | (((keyj & 127u) - 1u) / (uint32_t)SB)) == prevkey;
                if (k >= ibeg && k < iend) {
                    const int l = (int)(keyj & 127u) - 1 - lo;
                    if (l >= 0 && l < SB) {
                        if (!WINV) { // (WINV: the in-block couplings are inside the inverse image)
                            const int idx = FWD ? gs_tri_fwd<SB>(ii, l) : gs_tri_bwd(ii, l);
                            gs_store_tri<T>(tri, TRI, idx, diagBlockInv + 9 * (int64_t)i, bv[q]);
                        }
                        jj[q] = -1;
                    }
                }
            }
        }
    }
    // ---- 2a. every column except those of pass p-1 was published two or more passes ago: make sure pass p-2 is complete (it
    //          nearly always is) and fold those columns into the staged right-hand side now, off the critical path
    // point-to-point mode (nbr != null): a sub-block only waits for the adjacent blocks whose colours run
    // earlier in the sweep, each of which stamps flag[block] with the sweep number when its nodes are published — no pass-wide
    // counter, so a slow block holds up its neighbours only
    // flag index = 4 * block + sub-block.  A sub-block waits for the sub-block before it in its own block, which has waited
    // for the one before that, so a block's last sub-block in sweep order vouches for the whole block.
    int early_idx = -1, late_idx = -1; // lanes 0..27 of wavefront 0: what to see stamped before the early / the late gather
    if (nbr && tid < 26) {
        const int nb = nbr[(int64_t)b * 26 + tid];
        if (nb >= 0) {
            const int cn = nb >> 28, gid = nb & 0x0fffffff;
            int qlast = -1;
            for (int q = 0; q < P.npass; ++q)
                if (P.color[q] == cn) qlast = q;
            if (qlast >= 0 && qlast < p) { // that colour runs before this one
                if (qlast < p - 1)
                    early_idx = 4 * gid + P.sub[qlast];
                else {
                    late_idx = 4 * gid + P.sub[qlast];
                    if (qlast > 0 && P.color[qlast - 1] == cn) early_idx = 4 * gid + P.sub[qlast - 1];
                }
            }
        }
    }
    if (nbr && tid == 26 && p > 0 && P.color[p - 1] == P.color[p]) late_idx = 4 * b + P.sub[p - 1];
    if (nbr && tid == 27 && p > 1 && P.color[p - 2] == P.color[p]) early_idx = 4 * b + P.sub[p - 2];
    auto wait_blocks = [&](int idx) { // spin until that sub-block carries the current sweep number
        if (idx < 0) return;
        int spins = 0;
        while (__hip_atomic_load(flag + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 22) || ((spins & 1023) == 0 && *(volatile int*)err)) {
                *(volatile int*)err = 1;
                break;
            }
        }
    };
    if (dataflag) {
    }
    else if (nbr)
        wait_blocks(early_idx);
    else if (p > 1 && tid == 0) {
        const int need2 = P.wg_begin[p - 1] - P.wg_begin[p - 2];
        int spins = 0;
        while (__hip_atomic_load(done + p - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need2) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1 << 22) || ((spins & 1023) == 0 && *(volatile int*)err)) {
                *(volatile int*)err = 1;
                break;
            }
        }
    }
    __syncthreads(); // also orders the staging of D^-1 / D / rhs (and the zeroed triangle) before their users
    GS_CLK(1);
    auto is_late = [&](uint32_t keyj) { return (((keyj >> 28) << 8) | (((keyj & 127u) - 1u) / (uint32_t)SB)) == prevkey; };
    bool tail_late[RQ]; // the tail of the half row (slots past the first 64) holds columns of pass p-1 (rows are sorted to avoid it)
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
        const int ii = w + NW * q;
        tail_late[q] = false;
        if (ii >= cnt) continue; // wave-uniform
        const int i = node[q];
        T e0 = 0, e1 = 0, e2 = 0;
        if (jj[q] >= 0 && !late[q]) {
            const int64_t j = jj[q];
            T x0, x1, x2;
            ld3(j, x0, x1, x2);
            e0 = bv[q][0] * x0 + bv[q][3] * x1 + bv[q][6] * x2;
            e1 = bv[q][1] * x0 + bv[q][4] * x1 + bv[q][7] * x2;
            e2 = bv[q][2] * x0 + bv[q][5] * x1 + bv[q][8] * x2;
        }
        // half rows longer than one wave: plain strided tail; the in-block slots come first (FWD: last) in the range, so
        // the tail may still hold sub-block couplings
        bool tl = false;
        for (int k = kb2[q] + lane; k < ke[q]; k += 64) {
            const int j = col[(int64_t)i * 125 + k];
            const T* bb = val + ((int64_t)i * 125 + k) * 9;
            const uint32_t keyj = ckey[j], keyi = ckey[i];
            const int l = (int)(keyj & 127u) - 1 - lo;
            if ((keyj >> 7) == (keyi >> 7) && l >= 0 && l < SB) {
                if (!WINV) {
                    const int idx = FWD ? gs_tri_fwd<SB>(ii, l) : gs_tri_bwd(ii, l);
                    T bt[9];
#pragma unroll
                    for (int e = 0; e < 9; ++e) bt[e] = bb[e];
                    gs_store_tri<T>(tri, TRI, idx, diagBlockInv + 9 * (int64_t)i, bt);
                }
            }
            else if (is_late(keyj))
                tl = true;
            else {
                T x0, x1, x2;
                ld3(j, x0, x1, x2);
                e0 += bb[0] * x0 + bb[3] * x1 + bb[6] * x2;
                e1 += bb[1] * x0 + bb[4] * x1 + bb[7] * x2;
                e2 += bb[2] * x0 + bb[5] * x1 + bb[8] * x2;
            }
        }
        tail_late[q] = __ballot(tl) != 0ull;
        e0 = wave_sum(e0), e1 = wave_sum(e1), e2 = wave_sum(e2);
        if (lane == 0) srhs[3 * ii] -= e0, srhs[3 * ii + 1] -= e1, srhs[3 * ii + 2] -= e2;
    }
    GS_CLK(2);
    // ---- 2b. wait for the previous pass
    if (!dataflag) {
        if (nbr)
            wait_blocks(late_idx);
        else if (p > 0 && tid == 0) {
            const int need = P.wg_begin[p] - P.wg_begin[p - 1];
            int spins = 0;
            while (__hip_atomic_load(done + p - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(4);
                ++spins;
                if ((spins & 1023) == 0 && *(volatile int*)err) break; // some workgroup already gave up: drain quickly
                if (spins > (1 << 22)) {
                    *(volatile int*)err = 1;
                    break;
                }
            }
        }
        __syncthreads();
    }
    // x of other workgroups was published with write-through stores and is read with sc1 loads below: no cache
    // maintenance (buffer_wbl2 / buffer_inv) on either side
    // ---- 3. the columns of pass p-1 against the now final unknowns
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
        const int ii = w + NW * q;
        if (ii >= cnt) continue; // wave-uniform
        const int i = node[q];
        T s0 = 0, s1 = 0, s2 = 0;
        if (jj[q] >= 0 && late[q]) {
            const int64_t j = jj[q];
            T x0, x1, x2;
            ld3(j, x0, x1, x2);
            s0 = bv[q][0] * x0 + bv[q][3] * x1 + bv[q][6] * x2;
            s1 = bv[q][1] * x0 + bv[q][4] * x1 + bv[q][7] * x2;
            s2 = bv[q][2] * x0 + bv[q][5] * x1 + bv[q][8] * x2;
        }
        if (tail_late[q]) { // wave-uniform, rare
            for (int k = kb2[q] + lane; k < ke[q]; k += 64) {
                const int j = col[(int64_t)i * 125 + k];
                const T* bb = val + ((int64_t)i * 125 + k) * 9;
                const uint32_t keyj = ckey[j], keyi = ckey[i];
                const int l = (int)(keyj & 127u) - 1 - lo;
                if (!((keyj >> 7) == (keyi >> 7) && l >= 0 && l < SB) && is_late(keyj)) {
                    T x0, x1, x2;
                    ld3(j, x0, x1, x2);
                    s0 += bb[0] * x0 + bb[3] * x1 + bb[6] * x2;
                    s1 += bb[1] * x0 + bb[4] * x1 + bb[7] * x2;
                    s2 += bb[2] * x0 + bb[5] * x1 + bb[8] * x2;
                }
            }
        }
        s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
        if (lane == 0) gs_store_rhs<T>(sv, ii, sDinv + 9 * ii, srhs[3 * ii] - s0, srhs[3 * ii + 1] - s1, srhs[3 * ii + 2] - s2);
    }
    __syncthreads();
    GS_CLK(3);
#ifdef HOT_GS_CLOCKS
#define GS_CLK_OUT() \
    do { \
        GS_CLK(4); \
        if (tid == 0) \
            for (int i = 0; i < 5; ++i) atomicAdd(&gs_clk[p * 8 + i], gclk_[i]); \
        if (tid == 0) atomicAdd(&gs_clk[p * 8 + 7], 1ull); \
    } while (0)
#else
#define GS_CLK_OUT()
#endif
    if (WINV) {
        // h_r = a_r + sum_c W_rc a_c over the columns before (forward) / after (backward) row r; wavefront w forms rows 4 w .. 4 w + 3, lane = column
        const bool cin = lane < cnt;
        const T ac0 = cin ? sv[3 * lane] : (T)0, ac1 = cin ? sv[3 * lane + 1] : (T)0, ac2 = cin ? sv[3 * lane + 2] : (T)0;
        T h0 = 0, h1 = 0, h2 = 0; // of row 4 w + lane, lanes 0..3
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rr = 4 * w + k;
            if (rr >= cnt) break; // wave-uniform
            const bool act = FWD ? lane < rr : (lane > rr && cin);
            const int idx = act ? gs_winv_idx<FWD>(rr, lane) : TRI - 1; // masked lanes read the all-zero entry
            T Lw[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) Lw[e] = tri[e * TRI + idx];
            T t0 = Lw[0] * ac0 + Lw[3] * ac1 + Lw[6] * ac2, t1 = Lw[1] * ac0 + Lw[4] * ac1 + Lw[7] * ac2, t2 = Lw[2] * ac0 + Lw[5] * ac1 + Lw[8] * ac2;
            t0 = wave_sum(t0), t1 = wave_sum(t1), t2 = wave_sum(t2);
            if (lane == k) h0 = sv[3 * rr] + t0, h1 = sv[3 * rr + 1] + t1, h2 = sv[3 * rr + 2] + t2;
        }
        const int me = 4 * w + lane;
        if (lane < 4 && me < cnt) {
            const int64_t i = nodes[me];
            __hip_atomic_store(x + 3 * i, h0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(x + 3 * i + 1, h1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(x + 3 * i + 2, h2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (FWD) {
                const T* dd = sD + 9 * me;
                hD[3 * i] = dd[0] * h0 + dd[3] * h1 + dd[6] * h2, hD[3 * i + 1] = dd[1] * h0 + dd[4] * h1 + dd[7] * h2, hD[3 * i + 2] = dd[2] * h0 + dd[5] * h1 + dd[8] * h2;
            }
            else if (hD)
                hD[3 * i] += h0, hD[3 * i + 1] += h1, hD[3 * i + 2] += h2;
        }
        GS_CLK_OUT();
        return; // (data-flag hand-off only: the write-through stores are the publication)
    }
    if (w != 0) return;
    if (cnt > 0) gs_phase_b<T, FWD, SB, true>(tri, sv, nodes, cnt, lane, diagVal, diagBlockInv, x, hD, sD);
    GS_CLK_OUT();
    // ---- publish: the write-through stores of every lane have left the CU before lane 0 bumps the pass counter
    if (dataflag) return; // the write-through stores of phase B are the publication
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        if (nbr)
            __hip_atomic_store(flag + 4 * b + P.sub[p], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            __hip_atomic_fetch_add(done + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// h -= du in place (every entry, owned or not: what k_gs_residual<T, true> gathers)
template <class T>
__global__ void k_gs_hdiff(size_t n3, T* __restrict__ h, const T* __restrict__ du)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) h[i] -= du[i];
}
// r_i = sum over the nl slots preceding row i of A_ik (h - du)_k   (rows regrouped by k_gs_split_rows)
// DIFF: h holds h - du already (k_gs_hdiff): three gathered loads per entry instead of six — the gathers, not the matrix stream, are what
// the six-load version waits for (one cache line per lane and instruction)
template <class T, bool DIFF = false>
__global__ __launch_bounds__(256) void k_gs_residual(const int32_t* __restrict__ col, const T* __restrict__ val, const int32_t* __restrict__ rowcnt, const T* __restrict__ h,
    const T* __restrict__ du, T* __restrict__ r, int n, const uint8_t* __restrict__ own, const uint8_t* __restrict__ owner /*rank-local GS (hot_config.shard_gs): owning rank of every row, else null*/,
    int me, const T* __restrict__ l1e /*shard_gs = 2: E = D' - D of the l1-scaled sweep, 3 per row (else null): r - A du = L (h - du) + E du*/)
{
    const int lane = threadIdx.x & 63;
    const int row = xcd_block() * 4 + (threadIdx.x >> 6);
    if (row >= n || (own && !own[row])) return;
    const int nl = rowcnt[4 * row] + rowcnt[4 * row + 1];
    const int32_t* c = col + (int64_t)row * 125;
    const T* v = val + (int64_t)row * 1125;
    T s0 = 0, s1 = 0, s2 = 0;
    auto add = [&](int k, int j) {
        const T* b = v + k * 9;
        T x0 = h[3 * (int64_t)j], x1 = h[3 * (int64_t)j + 1], x2 = h[3 * (int64_t)j + 2];
        if (!DIFF) x0 -= du[3 * (int64_t)j], x1 -= du[3 * (int64_t)j + 1], x2 -= du[3 * (int64_t)j + 2];
        s0 += b[0] * x0 + b[3] * x1 + b[6] * x2;
        s1 += b[1] * x0 + b[4] * x1 + b[7] * x2;
        s2 += b[2] * x0 + b[5] * x1 + b[8] * x2;
    };
    {
        // lane = (entry, column of its 3 x 3 block), 21 entries per wavefront step: a lane reads 24 contiguous bytes of the matrix (the wavefront 1.5 KB
        // contiguous) and ONE gathered scalar, three lanes to a node.  With lane = entry (nine loads 72 bytes apart from lane to lane, three gathers of a
        // cache line per lane) the kernel waited for its load instructions, not for HBM: C2 level 0 270 -> 238 us per launch (4.7 TB/s).
        const int q = lane / 3, cc = lane - 3 * q;
        for (int base = 0; base < nl; base += 63) // (wave-uniform trip count: one round for most rows, two for a row late in the sweep order — at most 124 preceding entries)
#pragma unroll
        for (int k0 = 0; k0 < 63; k0 += 21) {
            const int k = base + k0 + q;
            const bool ok = lane < 63 && k < nl;
            const int kk = ok ? k : 0; // (branch-free: a lane without an entry reads the row's first one and multiplies by zero)
            const int64_t j = nt_load(c + kk);
            const T* b = v + kk * 9 + 3 * cc;
            const T b0 = nt_load(b), b1 = nt_load(b + 1), b2 = nt_load(b + 2);
            T x = h[3 * j + cc];
            if (!DIFF) x -= du[3 * j + cc];
            x = ok ? x : (T)0;
            s0 += b0 * x, s1 += b1 * x, s2 += b2 * x;
        }
    }
    if (owner) {
        // rank-local sweeps: the identity r - A du = L (h - du) holds for the rank's own diagonal block of A.  What is left of A du are the
        // couplings to other ranks' rows: those preceding the row are in the loop above already (h is zero there: never computed here, never
        // exchanged), those following it are picked out of the following half here, with the same expression
        const int ub = nl + 1 + rowcnt[4 * row + 2], ue = ub + rowcnt[4 * row + 3];
        for (int k = ub + lane; k < ue; k += 64) {
            const int j = c[k];
            if (owner[j] != me) add(k, j);
        }
    }
    s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2);
    if (lane == 0) {
        if (l1e) s0 += l1e[3 * (int64_t)row] * du[3 * (int64_t)row], s1 += l1e[3 * (int64_t)row + 1] * du[3 * (int64_t)row + 1], s2 += l1e[3 * (int64_t)row + 2] * du[3 * (int64_t)row + 2];
        r[3 * (int64_t)row] = s0, r[3 * (int64_t)row + 1] = s1, r[3 * (int64_t)row + 2] = s2;
    }
}

template <class T>
__global__ void k_cg_scalars(double* s, int what)
{
    // s[0]=zTrk  s[1]=dAu.du  s[2]=omega  s[3]=-omega  s[4]=zTrk_new  s[5]=beta
    if (what == 0) {
        s[2] = s[0] / s[1];
        s[3] = -s[2];
    }
    else {
        s[5] = s[4] / s[0];
        s[0] = s[4];
    }
}

// final_residual = false: the caller never reads r after this call (the post-smoothing leg of the V-cycle: the
// reference updates the residual there too, MultigridPreconditioner.h:266-318, but nothing consumes it), so the last
// residual update of the stationary smoothers is skipped.  The iterates u are unaffected.
template <class T>
void Ctx<T>::smooth_dev(int level, int kind, int iterations, T tolerance, T* u, T* r, T* du, T* dAu, bool final_residual)
{
    Level<T>& L = *levels[level];
    size_t n3 = 3 * (size_t)L.n;
    MaskScope mscope(this, halo_mode() ? L.mask() : nullptr); // halo mode: this level's vectors live on the rows the rank owns (replicated level: everywhere)
    const bool hm = L.part && halo_mode();
    bool tmp_marked = unset_level == level; // L.tmp carries the chained GS sweep's "not written yet" marks (restrict_dev / vcycle_dev ran just before)
    unset_level = -1; // whatever this call does with L.tmp, the marks are spent
    auto Aproject = [&](T* v) {
        if (level == 0 && !cfg.systemBCProject) project_dev(v);
    };
    auto scaler = [&](const T* in, T* out) { scale_dev(L, in, out); };
    if (kind == 0) {
        for (; iterations--;) {
            scaler(r, du);
            scal(n3, (T)cfg.topomega, du);
            axpy(n3, (T)1, du, u);
            if (!final_residual && iterations == 0) break;
            spmv_dev(L, du, dAu);
            Aproject(dAu);
            axpy(n3, (T)-1, dAu, r);
        }
    }
    else if (kind == 1) {
        for (; iterations--;) {
            double rr = dot_host(n3, r, r);
            if (std::sqrt(rr) < (double)tolerance) break;
            scaler(r, du);
            spmv_dev(L, du, dAu);
            Aproject(dAu);
            double a = dot_host(n3, du, r), b = dot_host(n3, du, dAu);
            T omega = (T)(a / b);
            axpy(n3, omega, du, u);
            axpy(n3, -omega, dAu, r);
        }
    }
    else if (kind == 2) {
        T* z = L.tmp.p;
        double* s = dscal.p + 40;
        const bool cg_unfused = ab_flag("HOT_CG_UNFUSED"); // A/B build only: one launch per vector operation
        const bool fused = !cg_unfused && !(level == 0 && !cfg.systemBCProject) && !L.part; // (partitioned level: the generic path below, whose SpMV exchanges)
        int cnt = 0;
        double zTrk = 0, tol = 0;
        // (not when ranks may share a device, nor after a spinning kernel has timed out on this context; fp64 only: in float the association of
        // the dot products decides which of several line-search halvings is taken two iterations later — cond ~ 1e8 —, and the whole-step fp32
        // parity test was validated against the launch-per-operation sums)
        if (fused && sizeof(T) == 8 && L.n <= 65536 && !gs_no_chain && !sharded() && !ab_flag("HOT_CG_LAUNCHES")) { // A/B build: HOT_CG_LAUNCHES = three launches per iteration on small levels too
            // the whole solve in one persistent launch (k_cg_persist), one host round trip for the iteration count
            const int G = std::min(std::min(ab_int("HOT_CG_WGS", 256), device_cus()), div_up(L.n, 16)); // 1024-thread workgroups that must all be resident: one per compute unit at most
            const int cg_ss = ab_int("HOT_CG_SLOT_STRIDE", 32); // doubles between the workgroups' deposit slots
            if (!cg_dep.p || cg_bar_dirty || cg_G != G) { // the deposit slots start "not written" (k_cg_persist's barrier); again after a launch that gave up — a barrier timed
                // out, which switches this path off until rearm_chain() switches it on 32 clean steps later — and when the grid changes (the slots are laid out by it)
                cg_dep.reserve((size_t)4 * 256 * cg_ss);
                HOT_LAUNCH(this, "gs_fill_unset", k_gs_fill_unset<double>, div_up((size_t)4 * 256 * cg_ss, 256), 256, 0, (size_t)4 * 256 * cg_ss, cg_dep.p);
                cg_bar_dirty = false, cg_G = G, cg_phase = 0;
            }
            // (levels of up to two rows per wavefront: everything of a row stays in registers between the barriers; A/B build: HOT_CG_STREAM = the streaming version)
            const bool cg_resident = L.n <= 2 * 16 * G && 3 * (size_t)L.n * sizeof(double) <= 150 * 1024 && !ab_flag("HOT_CG_STREAM"); // (du of the whole level in LDS)
            if (cg_resident && !attr_cg_set) {
                HOT_HIP(hipFuncSetAttribute((const void*)k_cg_persist<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                attr_cg_set = true;
            }
            if (cg_resident)
                HOT_LAUNCH(this, lname("cg_persistent", L.id).c_str(), (k_cg_persist<T, 2>), G, 1024, 3 * (size_t)L.n * sizeof(double), L.col.p, L.val.p, L.diagInv.p, L.initialResidual.p, u, r, z, du, dAu, L.n, iterations, cg_phase, cg_dep.p, cg_ss,
                    hscal + 40, hscal + 251, new_ticket(), (int*)(hscal + 250));
            else
                HOT_LAUNCH(this, lname("cg_persistent", L.id).c_str(), k_cg_persist<T>, G, 1024, 0, L.col.p, L.val.p, L.diagInv.p, L.initialResidual.p, u, r, z, du, dAu, L.n, iterations, cg_phase, cg_dep.p, cg_ss,
                    hscal + 40, hscal + 251, new_ticket(), (int*)(hscal + 250));
            wait_ticket(); // a timed-out barrier shows in hscal[250]: sync() inside throws ERR_RETRY and the caller redoes the operation with launches
            cnt = (int)hscal[41];
            if (ab_flag("HOT_CG_DBG")) // A/B build: 10 ns ticks of workgroup 0 by phase
                fprintf(stderr, "cg_persistent level %d: %d rows, %d workgroups, %d iterations; ticks pull %.0f product %.0f barrier1 %.0f update %.0f barrier2 %.0f direction %.0f barrier3 %.0f\n", L.id, L.n, G, cnt,
                    hscal[48], hscal[49], hscal[50], hscal[51], hscal[52], hscal[53], hscal[54]);
            cg_phase = (cg_phase + 1u + 3u * (unsigned)cnt) & 3u; // one barrier at the set-up, three an iteration
            iterations = 0;
        }
        else if (fused) {
            // no host round trip before the first iteration and one per group of iterations afterwards (see k_cg_setup)
            scaler(L.initialResidual.p, z);
            dot_to(n3, z, L.initialResidual.p, s + 7);
            scaler(r, z);
            copy(n3, z, du);
            dot_to(n3, z, r, s);
            HOT_LAUNCH(this, "cg_setup", k_cg_setup<T>, 1, 1, 0, s, hscal + 40);
            int group = std::max(1, std::min(cg_group, 16)); // as many iterations as the previous solve on this level needed
            bool active = true;
            while (iterations > 0 && active) {
                const int g = std::min(group, iterations);
                for (int k = 0; k < g; ++k) {
                    HOT_LAUNCH(this, lname("spmv", L.id).c_str(), k_cg_spmv_dot<T>, div_up(L.n, 4), 256, 0, L.col.p, L.val.p, du, dAu, L.n, s, gred(div_up(L.n, 4)));
                    HOT_LAUNCH(this, "cg_update", k_cg_update<T>, div_up(L.n, 256), 256, 0, L.diagInv.p, du, dAu, u, r, z, L.n, s, gred(div_up(L.n, 256)));
                    const bool last = k + 1 == g;
                    HOT_LAUNCH(this, "cg_direction", k_cg_direction<T>, div_up(n3, 256), 256, 0, n3, z, du, s, hscal + 40, last ? hscal + 251 : (double*)nullptr, last ? new_ticket() : 0.0);
                }
                iterations -= g;
                wait_ticket();
                active = !(hscal[40] < hscal[42]);
                group = 2;
            }
            cnt = (int)hscal[41];
            cg_group = cnt;
            iterations = 0;
        }
        else {
            scaler(L.initialResidual.p, z);
            double zTrk0 = dot_host(n3, z, L.initialResidual.p);
            scaler(r, z);
            copy(n3, z, du);
            zTrk = dot_host(n3, z, r);
            tol = (double)(T)(zTrk0 * 0.25); // cgratio = 0.5 hard-wired (:203-209)
            HOT_HIP(hipMemcpyAsync(s, &zTrk, sizeof(double), hipMemcpyHostToDevice, stream));
        }
        for (; iterations-- > 0;) {
            if (zTrk < tol) break;
            spmv_dev(L, du, dAu);
            Aproject(dAu);
            dot_to(n3, dAu, du, s + 1);
            HOT_LAUNCH(this, "cg_scalars", k_cg_scalars<T>, 1, 1, 0, s, 0);
            axpy_dev(n3, s + 2, 1.0, du, u);
            axpy_dev(n3, s + 3, 1.0, dAu, r);
            scaler(r, z);
            dot_to(n3, z, r, s + 4);
            HOT_LAUNCH(this, "cg_scalars", k_cg_scalars<T>, 1, 1, 0, s, 1);
            HOT_LAUNCH(this, "xpay", k_xpay_dev<T>, div_up(n3, 256), 256, 0, n3, s + 5, z, du);
            HOT_HIP(hipMemcpyAsync(hscal + 40, s, sizeof(double), hipMemcpyDeviceToHost, stream));
            sync();
            zTrk = hscal[40];
            ++cnt;
        }
        stats.linear_iterations += cnt;
    }
    else if (kind == 6) {
        // chebyshev_smooth (MultigridPreconditioner.h:227-264); the tolerance argument is unused there
        T* p = L.tmp.p;
        T d = (T)((L.lMax + L.lMin) / 2), c = (T)((L.lMax - L.lMin) / 2);
        int cnt = 1;
        iterations--;
        scaler(r, p);
        T alpha = 1 / d, beta;
        copy(n3, p, du);
        spmv_dev(L, du, dAu);
        Aproject(dAu);
        axpy(n3, alpha, du, u);
        axpy(n3, -alpha, dAu, r);
        for (; iterations-- > 0; ++cnt) {
            scaler(r, p);
            beta = (T)0.5 * c * c * alpha * alpha;
            if (cnt > 1) beta *= (T)0.5;
            alpha = 1 / (d - beta / alpha);
            scal(n3, beta, du); // du = p + beta du
            axpy(n3, (T)1, p, du);
            spmv_dev(L, du, dAu);
            Aproject(dAu);
            axpy(n3, alpha, du, u);
            if (!final_residual && iterations <= 0) break;
            axpy(n3, -alpha, dAu, r);
        }
    }
    else if (kind == 7) {
        // IC_smooth (MultigridPreconditioner.h:320-323): u = (L L^T)^-1 r, once; r is left alone.  The two triangular solves are block-GS
        // sweeps over the factor (mg_ic.hip): forward with D := L_ii writes y, backward with D := L_ii^T writes u
        HOT_CHECK(L.ic_ready && L.split, HOT_ERR_INVALID, "coarseSolver 7: the level has no incomplete-Cholesky factor (hot_build_mg)");
        if (!attr_gs_set) {
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_block<T, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GsLds<T, 64>::bytes));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_block<T, false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GsLds<T, 64>::bytes));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_sweep<T, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GsLds<T, 64>::bytes + 21 * 64 * sizeof(T) + 128)));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_sweep<T, false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GsLds<T, 64>::bytes + 21 * 64 * sizeof(T) + 128)));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_sweep<T, true, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GsLds<T, 64>::bytes + 21 * 64 * sizeof(T) + 128)));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_sweep<T, false, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GsLds<T, 64>::bytes + 21 * 64 * sizeof(T) + 128)));
            attr_gs_set = true;
        }
        T* y = L.tmp.p;
        for (int c = 0; c < 8; ++c) {
            const int b0 = L.color_block_begin[c], nb = L.color_block_begin[c + 1] - b0;
            if (nb > 0)
                HOT_LAUNCH(this, lname("ic_forward", L.id).c_str(), (k_gs_block<T, true, 64>), nb, 1024, (GsLds<T, 64>::bytes), L.ic_col.p, L.ic_val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p,
                    L.ic_d.p, L.ic_dinv.p, r, y, dAu, b0, 0 | (1 << 16), L.ic_rowcnt.p, L.ic_pad.p);
        }
        for (int c = 7; c >= 0; --c) {
            const int b0 = L.color_block_begin[c], nb = L.color_block_begin[c + 1] - b0;
            if (nb > 0)
                HOT_LAUNCH(this, lname("ic_backward", L.id).c_str(), (k_gs_block<T, false, 64>), nb, 1024, (GsLds<T, 64>::bytes), L.ic_col.p, L.ic_val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p,
                    L.ic_d.p, L.ic_dinvT.p, y, u, (T*)nullptr, b0, 0 | (1 << 16), L.ic_rowcnt.p, L.ic_pad.p);
        }
    }
    else if (kind == 5) {
        HOT_CHECK(L.nblocks > 0, HOT_ERR_INVALID, "GS smoother requested but the level was built without colouring");
        T* hdu = L.tmp.p;
        const bool simple_gs = ab_flag("HOT_SIMPLE_GS"); // A/B build only: one-wave-per-block reference kernel
        // hot_config.shard_gs = 1 on a row-partitioned level: a rank sweeps its own rows against its own rows only (processor-block GS: the
        // symmetric GS of the rank's diagonal block of A); one exchange per symmetric sweep instead of one per colour and direction
        const bool rank_local = L.part && cfg.shard_gs != 0;
        const int env_sb = cfg.gs_sub_block; // tuning override: sub-block size 16 / 32 / 64 (0 = by level size)
        const bool no_lres = ab_flag("HOT_GS_FULL_RESIDUAL"); // A/B build only: r -= A du by a full SpMV
        if (!attr_gs_set) {
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_block<T, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GsLds<T, 64>::bytes));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_block<T, false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GsLds<T, 64>::bytes));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_sweep<T, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GsLds<T, 64>::bytes + 21 * 64 * sizeof(T) + 128)));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_sweep<T, false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GsLds<T, 64>::bytes + 21 * 64 * sizeof(T) + 128)));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_sweep<T, true, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GsLds<T, 64>::bytes + 21 * 64 * sizeof(T) + 128)));
            HOT_HIP(hipFuncSetAttribute((const void*)k_gs_sweep<T, false, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GsLds<T, 64>::bytes + 21 * 64 * sizeof(T) + 128)));
            attr_gs_set = true;
#ifdef HOT_AB_KERNELS
            if (const char* e = getenv("HOT_GS_DBG")) {
                const int f = atoi(e);
                HOT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gs_dbg_flags), &f, sizeof(int)));
            }
#endif
        }
        // sub-block size: levels whose colours hold more blocks than the chip has CUs run half blocks (36 KB LDS, 4
        // workgroups per CU, one round per launch); small levels are latency-bound per launch and keep whole blocks
        int max_nb = 0;
        for (int c = 0; c < 8; ++c) max_nb = std::max(max_nb, L.color_block_begin[c + 1] - L.color_block_begin[c]);
        // one launch per half sweep (k_gs_sweep, passes chained by device-scope counters) unless the A/B switches ask
        // for one launch per pass
        const bool multilaunch = cfg.gs_chain == 1, force_dataflow = cfg.gs_chain == 2; // tuning overrides (0 = by level size)
        // measured (C2, fp64): the chained launch wins on levels whose colours fit the chip in one round (latency-bound
        // passes, no launch gaps); on the finest level the waiting workgroups cost more than the kernel boundaries
        const bool dataflow = !gs_no_chain && !multilaunch && !simple_gs && L.split && !L.part && (force_dataflow || max_nb <= 256); // a chained launch cannot stop for the exchange
        // (chained levels of more than 32 blocks a colour run half blocks too: twice the workgroups stream a colour's off-block rows — C2
        // level 1, 91 blocks a colour: 159 against 178 us per half sweep, 10.9 against 12.2 ms per step)
        // (chained levels with inverse images, k_gs_winv: whole blocks — the 64-row pass costs one dense product, not 64 dependent steps)
        const int sb = env_sb ? env_sb : ((max_nb > 256 || (dataflow && max_nb > 32 && !L.gs_w_ready)) ? 32 : 64);
        const int gs_threads = sb == 64 ? 1024 : 512;
        const int nsub = 64 / sb;
        HOT_CHECK(L.split || simple_gs, HOT_ERR_INVALID, "block GS kernels need the regrouped rows (k_gs_split_rows)");
        const int32_t* rc = L.rowcnt.p;
        // one launch per colour: its sub-blocks are walked inside the kernel (A/B switch: one launch per sub-block)
        const bool split_launches = ab_flag("HOT_GS_SPLIT_LAUNCHES"); // A/B build only, read per call: the tests flip it on one matrix
        const int nmerge = (split_launches || simple_gs) ? 1 : nsub;
        auto pass = [&](bool fwd, int c, int h) {
            int b0 = L.color_block_begin[c], nb = L.color_block_begin[c + 1] - b0;
            if (nb <= 0) return;
            if (nmerge > 1 && h != 0) return; // sub-blocks 1.. ride along with sub-block 0's launch
            const char* nm = fwd ? "gs_forward" : "gs_backward";
            const T* rhs = fwd ? r : dAu;
            T* xx = fwd ? hdu : du;
            // sharded: this rank substitutes the colour blocks it owns (a contiguous run of the colour's list), then every rank
            // receives the colour's new values before the next colour starts — the reference's update order, across ranks
            struct AfterPass {
                Ctx<T>* ctx;
                Level<T>& L;
                T* x;
                int c;
                bool last;
                ~AfterPass()
                {
                    if (L.part && last) ctx->exchange(L, x, c);
                }
            } after{ this, L, xx, c, !rank_local && (nmerge > 1 || (fwd ? h == nsub - 1 : h == 0)) };
            if (L.part) {
                const int R1 = comm.size + 1;
                b0 += L.csplit[c * R1 + comm.rank], nb = L.csplit[c * R1 + comm.rank + 1] - L.csplit[c * R1 + comm.rank];
                if (nb <= 0) return;
            }
            T* hD = fwd ? dAu : ((simple_gs || L.part) ? (T*)nullptr : u); // backward block kernels add du to u themselves (partitioned level: only the owner's rows would get it, see below)
#ifdef HOT_AB_KERNELS
            if (simple_gs) {
                if (h != 0) return;
                if (fwd)
                    HOT_LAUNCH(this, lname(nm, L.id).c_str(), (k_gs_color<T, true>), nb, 64, 0, L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, L.gs_d(), L.gs_dinv(), rhs, xx, hD, b0, nb);
                else
                    HOT_LAUNCH(this, lname(nm, L.id).c_str(), (k_gs_color<T, false>), nb, 64, 0, L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, L.gs_d(), L.gs_dinv(), rhs, xx, hD, b0, nb);
                return;
            }
#endif
#define HOT_GS_CASE(F, S)                                                                                                                                      \
    HOT_LAUNCH(this, lname(nm, L.id).c_str(), (k_gs_block<T, F, S>), nb, gs_threads, (GsLds<T, S>::bytes), L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, \
        L.gs_d(), L.gs_dinv(), rhs, xx, hD, b0, h | (nmerge << 16), rc, L.gs_pad.p)
            if (fwd) {
                if (sb == 64) HOT_GS_CASE(true, 64);
                else if (sb == 32) HOT_GS_CASE(true, 32);
                else HOT_GS_CASE(true, 16);
            }
            else {
                if (sb == 64) HOT_GS_CASE(false, 64);
                else if (sb == 32) HOT_GS_CASE(false, 32);
                else HOT_GS_CASE(false, 16);
            }
#undef HOT_GS_CASE
        };
        // ---- the finest-level colour passes as kernel pairs: k_gs_offblock (row sums over the off-block columns), then k_gs_subst (the blocks'
        // substitutions).  (Tried: the part of the next colour's off-block sums that reads only colours finished two passes ago on a second,
        // low-priority stream beside the substitution — event waits between the streams cost more than the overlap gains: C2 95 vs 84 ms a step.)
        const bool pair_path = sb == 32 && L.gs_img_ready && nmerge > 1 && !simple_gs && !ab_flag("HOT_GS_V1"); // A/B build: HOT_GS_V1 = one k_gs_block launch per colour
        auto pair_sweep = [&](bool fwd) {
            const T* rhs = fwd ? r : dAu;
            T* xx = fwd ? hdu : du;
            T* hD = fwd ? dAu : u;
            const char* nmT = fwd ? "gs_forward" : "gs_backward";
            const char* nmO = fwd ? "gs_forward_off" : "gs_backward_off";
            T* hsub = (!fwd && !L.part) ? hdu : (T*)nullptr; // h - du for the residual, row by row (partitioned level: a rank substitutes its own blocks only — k_gs_hdiff afterwards)
            if (L.part) hD = fwd ? dAu : (T*)nullptr; // partitioned level: u takes the correction in one axpy after the colour exchanges (only the owner's rows would get it here)
            for (int q = 0; q < 8; ++q) {
                const int c = fwd ? q : 7 - q;
                int b0 = L.color_block_begin[c], nb = L.color_block_begin[c + 1] - b0;
                if (nb <= 0) continue;
                // sharded: this rank sums and substitutes the colour blocks it owns (a contiguous run of the colour's list), then every rank receives
                // the colour's new values before the next colour starts (rank-local sweeps: no hand-off inside the sweep) — as the k_gs_block path does
                struct AfterColour {
                    Ctx<T>* ctx;
                    Level<T>& L;
                    T* x;
                    int c;
                    bool on;
                    ~AfterColour()
                    {
                        if (on) ctx->exchange(L, x, c);
                    }
                } after{ this, L, xx, c, L.part && !rank_local };
                if (L.part) {
                    const int R1 = comm.size + 1;
                    b0 += L.csplit[c * R1 + comm.rank], nb = L.csplit[c * R1 + comm.rank + 1] - L.csplit[c * R1 + comm.rank];
                    if (nb <= 0) continue;
                }
                // (the first colour walked has no off-block columns before it — an empty slot range, like a rank without rows of the colour: no launch)
                const int s0 = L.gs_slot_rng[fwd ? 0 : 1][0][c], s1 = L.gs_slot_rng[fwd ? 0 : 1][1][c];
                const T* img_c = L.gs_img.p + L.gs_img_shift[c] * (long long)GsImg<T>::per_block; // (images exist for the owned blocks only: the colour's base, shifted)
                const uint16_t* imgi_c = L.gs_imgi.p + L.gs_img_shift[c] * 2 * (long long)GsImg<T>::idx_per_dir;
                const int grid = std::max(1, ab_int("HOT_GS_OFF_WAVES", 4096) / 4);
                if (s1 > s0) HOT_LAUNCH(this, lname(nmO, L.id).c_str(), k_gs_offblock<T>, grid, 256, 0, L.gs_slot.p, L.gs_col.p, L.val.p, L.gs_pad.p, xx, L.gs_p1.p, s0, s1);
                // (eight columns in flight per block: 4 .. 16 change nothing, §6 of DESIGN.md)
#ifdef HOT_AB_KERNELS
                const int depth = ab_int("HOT_GS_SUBST_D", 8); // A/B build: image columns in flight per block (4 / 6 / 10 / 12 / 16 instead of 8)
#define HOT_SUBST_D(DD)                                                                                                                                                       \
    if (depth == DD) {                                                                                                                                                        \
        if (fwd)                                                                                                                                                              \
            HOT_LAUNCH(this, lname(nmT, L.id).c_str(), (k_gs_subst<T, true, DD>), nb, 64, 0, img_c, imgi_c, L.gs_pad.p, L.gs_p1.p, xx, hD, b0, rhs, hsub);                 \
        else                                                                                                                                                                  \
            HOT_LAUNCH(this, lname(nmT, L.id).c_str(), (k_gs_subst<T, false, DD>), nb, 64, 0, img_c, imgi_c, L.gs_pad.p, L.gs_p1.p, xx, hD, b0, rhs, hsub);                \
        continue;                                                                                                                                                             \
    }
                HOT_SUBST_D(4)
                HOT_SUBST_D(6)
                HOT_SUBST_D(10)
                HOT_SUBST_D(12)
                HOT_SUBST_D(16)
#undef HOT_SUBST_D
#endif
                if (fwd)
                    HOT_LAUNCH(this, lname(nmT, L.id).c_str(), (k_gs_subst<T, true, 8>), nb, 64, 0, img_c, imgi_c, L.gs_pad.p, L.gs_p1.p, xx, hD, b0, rhs, hsub);
                else
                    HOT_LAUNCH(this, lname(nmT, L.id).c_str(), (k_gs_subst<T, false, 8>), nb, 64, 0, img_c, imgi_c, L.gs_pad.p, L.gs_p1.p, xx, hD, b0, rhs, hsub);
            }
        };
        // ---- one rank: the colour pass as ONE launch, the next colour's older off-block sums beside this colour's substitutions (k_gs_colour)
        const bool fused_path = pair_path && L.gs_fused_ready && !L.part;
        // the forward sweep's last colour also runs its blocks' backward substitutions (k_gs_colour<.., TURN>); A/B build: HOT_GS_NO_TURN = two launches
        const bool turn = !ab_flag("HOT_GS_NO_TURN");
        auto colour_sweep = [&](bool fwd) {
            const T* rhs = fwd ? r : dAu;
            T* xx = fwd ? hdu : du;
            T* hD = fwd ? dAu : u;
#ifndef HOT_AB_KERNELS
            const char* nm = fwd ? "gs_forward_fused" : "gs_backward_fused";
#endif
            T* hsub = !fwd ? hdu : (T*)nullptr; // h - du for the residual, row by row
            const int nstream = std::max(8, ab_int("HOT_GS_OFF_WAVES", 4096) / 4 / 8 * 8);
            int last = -1; // the last colour of the forward sweep that has blocks = the first of the backward sweep
            for (int c = 7; c >= 0 && last < 0; --c)
                if (L.color_block_begin[c + 1] > L.color_block_begin[c]) last = c;
            for (int q = 0; q < 8; ++q) {
                const int c = fwd ? q : 7 - q, cn = fwd ? c + 1 : c - 1;
                const int b0 = L.color_block_begin[c], nb = L.color_block_begin[c + 1] - b0, nb_pad = (nb + 7) & ~7;
                const int s0 = (cn >= 0 && cn < 8) ? L.gs_slot_rng2[fwd ? 0 : 2][0][cn] : 0, s1 = (cn >= 0 && cn < 8) ? L.gs_slot_rng2[fwd ? 0 : 2][1][cn] : 0;
                const bool do_turn = turn && c == last && nb > 0; // (forward: substitute both ways; backward: the forward launch has done this colour)
                const int grid = (!fwd && do_turn ? 0 : nb_pad) + (s1 > s0 ? nstream : 0);
                if (grid == 0) continue;
                const int nbk = (!fwd && do_turn) ? 0 : nb, nbk_pad = (!fwd && do_turn) ? 0 : nb_pad;
#ifdef HOT_AB_KERNELS
                const std::string nmq = ab_flag("HOT_GS_PROF_COLOURS") ? std::string(fwd ? "gs_forward_fused_q" : "gs_backward_fused_q") + char('0' + q) : std::string(fwd ? "gs_forward_fused" : "gs_backward_fused"); // A/B build: one profile record per pass of the half sweep
                const char* nm = nmq.c_str();
#endif
#define HOT_COLOUR_D(DD)                                                                                                                                                       \
    do {                                                                                                                                                                       \
        if (fwd && do_turn)                                                                                                                                                    \
            HOT_LAUNCH(this, lname(nm, L.id).c_str(), (k_gs_colour<T, true, DD, true>), grid, 256, 0, L.gs_img.p, L.gs_imgi.p, L.gs_pad.p, L.gs_srec.p, xx, hD, b0, nbk, nbk_pad, rhs, \
                (!L.part ? hdu : (T*)nullptr), L.gs_slot.p, L.gs_col.p, L.val.p, L.gs_p1.p, s0, s1, du, u);                                                                       \
        else if (fwd)                                                                                                                                                          \
            HOT_LAUNCH(this, lname(nm, L.id).c_str(), (k_gs_colour<T, true, DD>), grid, 256, 0, L.gs_img.p, L.gs_imgi.p, L.gs_pad.p, L.gs_srec.p, xx, hD, b0, nbk, nbk_pad, rhs, hsub, \
                L.gs_slot.p, L.gs_col.p, L.val.p, L.gs_p1.p, s0, s1, (T*)nullptr, (T*)nullptr);                                                                                \
        else                                                                                                                                                                   \
            HOT_LAUNCH(this, lname(nm, L.id).c_str(), (k_gs_colour<T, false, DD>), grid, 256, 0, L.gs_img.p, L.gs_imgi.p, L.gs_pad.p, L.gs_srec.p, xx, hD, b0, nbk, nbk_pad, rhs, hsub, \
                L.gs_slot.p, L.gs_col.p, L.val.p, L.gs_p1.p, s0, s1, (T*)nullptr, (T*)nullptr);                                                                                \
    } while (0)
#ifdef HOT_AB_KERNELS
                const int depth = ab_int("HOT_GS_SUBST_D", 0); // A/B build: image columns in flight per block (0: the production choice)
                if (depth == 4) {
                    HOT_COLOUR_D(4);
                    continue;
                }
                if (depth == 6) {
                    HOT_COLOUR_D(6);
                    continue;
                }
                if (depth == 7) {
                    HOT_COLOUR_D(7);
                    continue;
                }
                if (depth == 8) {
                    HOT_COLOUR_D(8);
                    continue;
                }
#endif
                // image columns in flight per substitution wavefront: fp64 seven (126 registers of them: the kernel stays below 168, three wavefronts per SIMD for
                // the streaming role; with eight it needs 176 — two per SIMD —: 42.6 against 41.4 us per launch at C2), fp32 eight
                if constexpr (sizeof(T) == 8)
                    HOT_COLOUR_D(7);
                else
                    HOT_COLOUR_D(8);
#undef HOT_COLOUR_D
#ifdef HOT_GSC_CLOCKS
                if (L.id == 0) hipLaunchKernelGGL(k_gsc_pass, dim3(1), dim3(256), 0, stream, (fwd ? 0 : 8) + q, nbk, grid - nbk_pad);
#endif
            }
        };
        HOT_CHECK(sb == 16 || sb == 32 || sb == 64, HOT_ERR_INVALID, "hot_config.gs_sub_block must be 0 (auto), 16, 32 or 64");
        HOT_CHECK(cfg.gs_chain >= 0 && cfg.gs_chain <= 2, HOT_ERR_INVALID, "hot_config.gs_chain must be 0 (auto), 1 (one launch per colour) or 2 (one chained launch per half sweep)");
        if (tmp_marked && !(dataflow && !ab_flag("HOT_GS_PASS_COUNTERS") && !ab_flag("HOT_GS_BLOCK_FLAGS"))) zero(n3, hdu), tmp_marked = false; // (cannot happen: gs_marks_wanted takes the same decision)
        GsPasses PF{}, PB{};
        if (dataflow) {
            auto add = [&](GsPasses& P, int c, int h) {
                int b0 = L.color_block_begin[c], nb = L.color_block_begin[c + 1] - b0;
                if (nb <= 0) return;
                P.block0[P.npass] = b0, P.sub[P.npass] = h, P.color[P.npass] = c, P.wg_begin[P.npass + 1] = P.wg_begin[P.npass] + nb;
                ++P.npass;
            };
            for (int c = 0; c < 8; ++c)
                for (int h = 0; h < nsub; ++h) add(PF, c, h);
            for (int c = 7; c >= 0; --c)
                for (int h = nsub - 1; h >= 0; --h) add(PB, c, h);
            gs_done.reserve(64);
        }
        bool du_marked = false;
        auto sweep = [&](bool fwd) {
            const GsPasses& P = fwd ? PF : PB;
            if (P.npass == 0) return;
            const char* nm = fwd ? "gs_forward" : "gs_backward";
            const T* rhs = fwd ? r : dAu;
            T* xx = fwd ? hdu : du;
            T* hD = fwd ? dAu : (simple_gs ? (T*)nullptr : u); // backward block kernels add du to u themselves
            // hand-off between passes: point-to-point block flags when a block is one sub-block (A/B switch: pass counters)
            const bool pass_counters = ab_flag("HOT_GS_PASS_COUNTERS");
            const bool block_flags = ab_flag("HOT_GS_BLOCK_FLAGS"); // A/B build only: per-block sweep stamps instead of the unknowns being their own flags
            const bool p2p = !pass_counters;
            const int dataflag = (pass_counters || block_flags) ? 0 : 1;
            // "not written yet" marks of the sweep's target: the forward target (L.tmp) by the kernel that ran just before this smoother on the level
            // (restrict_dev / the k_apmv_sub of the way up: unset_level), the backward target by the forward sweep itself; a fill launch otherwise
            const bool marked = fwd ? tmp_marked : du_marked;
            if (fwd) tmp_marked = false;
            if (dataflag && !marked)
                HOT_LAUNCH(this, "gs_fill_unset", k_gs_fill_unset<T>, div_up(n3, 256), 256, 0, n3, xx);
            else if (dataflag)
                ;
            else if (p2p)
                ++gs_epoch;
            else
                HOT_HIP(hipMemsetAsync(gs_done.p, 0, 40 * sizeof(int), stream));
            const int grid = P.wg_begin[P.npass];
#define HOT_GS_CASE(F, S, ...)                                                                                                                                         \
    HOT_LAUNCH(this, lname(nm, L.id).c_str(), (k_gs_sweep<T, F, S, ##__VA_ARGS__>), grid, 16 * S, (GsLds<T, S>::bytes + 21 * S * sizeof(T) + 128), L.col.p, L.val.p, L.ckey.p, L.gs_order.p, L.gs_block_start.p, \
        L.diagVal.p, L.diagBlockInv.p, rhs, xx, hD, P, rc, gs_done.p, (int*)(hscal + 250), p2p ? L.gs_nbr.p : (const int32_t*)nullptr, L.gs_flag.p, gs_epoch, dataflag, \
        (fwd && dataflag) ? du : (T*)nullptr, L.gs_w.p)
            du_marked = fwd && dataflag;
            // whole-block passes on the precomputed inverses of the in-block triangles (A/B build: HOT_GS_NO_WINV = the 64-step substitution)
            const bool winv = sb == 64 && L.gs_w_ready && dataflag && !ab_flag("HOT_GS_NO_WINV");
            if (winv) {
                if (fwd)
                    HOT_GS_CASE(true, 64, true);
                else
                    HOT_GS_CASE(false, 64, true);
            }
            else if (fwd) {
                if (sb == 64) HOT_GS_CASE(true, 64);
                else if (sb == 32) HOT_GS_CASE(true, 32);
                else HOT_GS_CASE(true, 16);
            }
            else {
                if (sb == 64) HOT_GS_CASE(false, 64);
                else if (sb == 32) HOT_GS_CASE(false, 32);
                else HOT_GS_CASE(false, 16);
            }
#undef HOT_GS_CASE
        };
#ifdef HOT_GS_CLOCKS
        auto clk_report = [&](const char* what) {
            unsigned long long h[34 * 8];
            HOT_HIP(hipStreamSynchronize(stream));
            HOT_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(gs_clk), sizeof(h)));
            if (h[7]) {
                fprintf(stderr, "k_gs_sweep %s level %d, clocks per workgroup [header, rows, early gathers, wait + late gathers, solve]:", what, L.id);
                for (int q = 0; q < 33 && h[q * 8 + 7]; ++q) fprintf(stderr, " | p%d(%llu wg) %.0f %.0f %.0f %.0f %.0f", q, h[q * 8 + 7], (double)h[q * 8] / h[q * 8 + 7], (double)h[q * 8 + 1] / h[q * 8 + 7], (double)h[q * 8 + 2] / h[q * 8 + 7], (double)h[q * 8 + 3] / h[q * 8 + 7], (double)h[q * 8 + 4] / h[q * 8 + 7]);
                fprintf(stderr, "\n");
            }
            memset(h, 0, sizeof(h));
            HOT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gs_clk), h, sizeof(h)));
        };
#endif
        iterations = ((iterations + 1) >> 1);
        for (; iterations--;) {
            prof.count(lname("gs_symsweeps", L.id));
            // no memset of hdu / du: a sweep writes every node before any later node reads it (only preceding nodes are read) —
            // except with rank-local sweeps, where the other ranks' unknowns are read as the zeros the sweep starts from
            if (rank_local) zero(n3, hdu), zero(n3, du);
            if (dataflow) {
                sweep(true);
#ifdef HOT_GS_CLOCKS
                clk_report("forward");
#endif
            }
            else if (fused_path)
                colour_sweep(true);
            else if (pair_path)
                pair_sweep(true);
            else
                for (int c = 0; c < 8; ++c)
                    for (int h = 0; h < nsub; ++h) pass(true, c, h);
            // dAu now holds D h ; du = backward solve
            if (dataflow)
                sweep(false);
            else if (fused_path) {
                colour_sweep(false);
#ifdef HOT_GSC_CLOCKS
                if (L.id == 0) { // every 10 symmetric sweeps: the per-role clocks of the 15 passes, averaged
                    static int sweeps = 0;
                    if (++sweeps % 10 == 0) {
                        unsigned long long h[16][12];
                        HOT_HIP(hipStreamSynchronize(stream));
                        HOT_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(gsc_clk), sizeof(h)));
                        fprintf(stderr, "k_gs_colour, per-role clocks (100 MHz), us: per block of the substitution wavefront | per launch\n"
                                        "pass        blocks  trips1+2  requests  wait-prev  64-steps  stores | summing-done | block-wgs-span  stream-wgs-span\n");
                        for (int p = 0; p < 16; ++p) {
                            if (!h[p][9] || !h[p][0]) continue;
                            const double nb = (double)h[p][0], nl = (double)h[p][9];
                            fprintf(stderr, "%s q%d  %7.0f  %8.2f  %8.2f  %9.2f  %8.2f  %6.2f | %12.2f | %14.2f  %15.2f\n", p < 8 ? "forward " : "backward", p & 7, nb / nl, h[p][1] / (100 * nb), h[p][2] / (100 * nb),
                                h[p][3] / (100 * nb), h[p][4] / (100 * nb), h[p][5] / (100 * nb), h[p][7] / (100 * nb), h[p][6] / (100 * nl), h[p][8] / (100 * nl));
                        }
                        memset(h, 0, sizeof(h));
                        HOT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gsc_clk), h, sizeof(h)));
                    }
                }
#endif
            }
            else if (pair_path)
                pair_sweep(false);
            else
                for (int c = 7; c >= 0; --c)
                    for (int h = nsub - 1; h >= 0; --h) pass(false, c, h);
            if (rank_local) exchange(L, du, -1); // the one hand-off of a rank-local symmetric sweep: every rank's du (halo mode: the entries this rank reads)
            if (simple_gs || L.part) axpy(n3, (T)1, du, u); // partitioned level: du is complete on every rank after the colour exchanges, u stays replicated
            if (!final_residual && iterations == 0) break;
            if (L.split && !simple_gs && !(level == 0 && !cfg.systemBCProject) && !no_lres) {
                // r - A du = L (h - du): with (D+L) h = r and (D+U) du = D h the full product A du collapses to the
                // strictly-preceding half of the matrix applied to (h - du) (same value, half the bytes of an SpMV)
                if (!dataflow) { // (the chained sweeps keep marks in hdu)
                    if (!(pair_path && !L.part)) HOT_LAUNCH(this, "gs_hdiff", k_gs_hdiff<T>, div_up(n3, 256), 256, 0, n3, hdu, du); // (the pair path's backward substitutions have subtracted already)
                    HOT_LAUNCH(this, lname("gs_residual", L.id).c_str(), (k_gs_residual<T, true>), xcd_grid(div_up(L.n, 4)), 256, 0, L.col.p, L.val.p, L.rowcnt.p, hdu, du, r, L.n, L.mask(),
                        rank_local ? L.owner.p : (const uint8_t*)nullptr, comm.rank, (rank_local && L.l1) ? L.gsE.p : (const T*)nullptr);
                }
                else
                    HOT_LAUNCH(this, lname("gs_residual", L.id).c_str(), k_gs_residual<T>, xcd_grid(div_up(L.n, 4)), 256, 0, L.col.p, L.val.p, L.rowcnt.p, hdu, du, r, L.n, L.mask(),
                        rank_local ? L.owner.p : (const uint8_t*)nullptr, comm.rank, (rank_local && L.l1) ? L.gsE.p : (const T*)nullptr);
                if (!hm) exchange(L, r, -1); // first-generation sharding: the restriction / the next smoother read all of r (halo mode: r is needed on owned rows only; restrict_dev fetches what it reads)
            }
            else {
                spmv_dev(L, du, dAu);
                Aproject(dAu);
                axpy(n3, (T)-1, dAu, r);
            }
        }
    }
    else
        HOT_CHECK(false, HOT_ERR_INVALID, "unsupported smoother kind");
}

template <class T>
void Ctx<T>::vcycle_dev(const T* in, T* out)
{
    int levelCnt = (int)levels.size();
    int times = cfg.times, levelscale = cfg.levelscale;
    int splitLevel;
    auto downIter = [&](int level) { return times + level * levelscale; };
    auto upIter = [&](int level) { return cfg.topDownMGS ? 0 : times + level * levelscale; };
    const bool baseline = cfg.useBaselineMultigrid != 0; // GS smoother, PCG on top, 10000 top iterations (MultigridSimulation.inl:446-453)
    auto topIter = [&](int level) {
        if (cfg.topDownMGS || baseline) return 10000;
        if (cfg.levelCnt == 1) return times + level * levelscale;
        if (!(cfg.coarseSolver == 2 || cfg.coarseSolver == 6)) return (times + level * levelscale) * 3;
        return 10000;
    };
    splitLevel = cfg.topDownMGS ? 1 : cfg.levelCnt - 1;
    T tolTop = (T)(cfg.cneps * cfg.cneps);
    auto run = [&](bool regular, int level, T* sol, int its, bool final_residual = true) {
        Level<T>& L = *levels[level];
        smooth_dev(level, regular ? (baseline ? 5 : cfg.smoother) : (baseline ? 2 : cfg.coarseSolver), its, regular ? (T)0 : tolTop, sol, L.residual.p, L.du.p, L.dAu.p, final_residual);
    };
    stats.vcycles++;
    Level<T>& L0 = *levels[0];
    size_t n0 = 3 * (size_t)L0.n;
    // residual := in (dRhs == 0, ImplicitSolver.h:483-484,579: correctResidualProjection is the identity), out := 0 and, on a single level, the
    // top solver's initial residual: one launch instead of two copies and a fill
    HOT_LAUNCH(this, "vcycle_start", k_vcycle_start<T>, (int)std::min<size_t>(div_up(n0, 256), 2048), 256, 0, n0, in, L0.residual.p, levelCnt > 1 ? (T*)nullptr : L0.initialResidual.p, out);
    if (levelCnt > 1) restrict_dev(0, L0.residual.p, levels[1]->initialResidual.p);
    for (int l = 1; l < levelCnt - 1; ++l) restrict_dev(l, levels[l]->initialResidual.p, levels[l + 1]->initialResidual.p);
    int level;
    for (level = 0; level < levelCnt - 1; ++level) {
        T* sol = level == 0 ? out : levels[level]->sol.p;
        run(level < splitLevel, level, sol, level < splitLevel ? upIter(level) : topIter(level));
        restrict_dev(level, levels[level]->residual.p, levels[level + 1]->residual.p, levels[level + 1]->sol.p); // (clears the coarse iterate on the way)
    }
    run(false, level, level == 0 ? out : levels[level]->sol.p, topIter(level));
    for (--level; level >= 0; --level) {
        Level<T>& L = *levels[level];
        T* sol = level == 0 ? out : L.sol.p;
        size_t n3 = 3 * (size_t)L.n;
        prolong_dev(level, levels[level + 1]->sol.p, L.du.p);
        axpy(n3, (T)1, L.du.p, sol);
        const bool full_spmv = ab_flag("HOT_MG_FULL_SPMV"); // A/B build only: r -= A (P e) like the reference
        if (full_spmv) {
            spmv_dev(L, L.du.p, L.dAu.p);
            axpy(n3, (T)-1, L.dAu.p, L.residual.p);
        }
        else
        {
            T* mark = gs_marks_wanted(level) ? L.tmp.p : (T*)nullptr;
            HOT_LAUNCH(this, lname("apmv", L.id).c_str(), k_apmv_sub<T>, xcd_grid(div_up(L.n, 4)), 256, 0, L.apc.p, L.apv.p, levels[level + 1]->sol.p, L.residual.p, L.n, L.mask(), mark);
            unset_level = mark ? level : -1;
        }
        if (L.part && !halo_mode() && (level < splitLevel ? (baseline ? 5 : cfg.smoother) : (baseline ? 2 : cfg.coarseSolver)) != 5)
            exchange(L, L.residual.p, -1); // a GS post-smoother reads only the rows it owns; every other smoother runs replicated vector algebra on all of r
        run(level < splitLevel, level, sol, level < splitLevel ? downIter(level) : topIter(level), false);
    }
}

template <class T>
void Ctx<T>::precondition_dev(const T* in, T* out)
{
    HOT_CHECK(!levels.empty() && levels[0]->built, HOT_ERR_INVALID, "preconditioner used before hot_build_mg");
    vcycle_dev(in, out);
}

// ------------------------------------------------------------------------------------------------ C ABI wrappers
template <class T>
void Ctx<T>::spmv(int32_t level, const void* x, void* y)
{
    need(level >= 0 && level < (int)levels.size(), "level out of range");
    Level<T>& L = *levels[level];
    size_t n3 = 3 * (size_t)L.n;
    DBuf<T> a, b;
    a.reserve(n3), b.reserve(n3);
    HOT_HIP(hipMemcpyAsync(a.p, x, n3 * sizeof(T), hipMemcpyDefault, stream));
    spmv_dev(L, a.p, b.p);
    if (halo_mode()) gather_all(L, b.p); // the C ABI hands out complete vectors
    download(y, b.p, n3);
    sync();
}
template <class T>
void Ctx<T>::restrict_(int32_t level, const void* fine, void* coarse)
{
    need(level >= 0 && level + 1 < (int)levels.size(), "level out of range");
    size_t nf = 3 * (size_t)levels[level]->n, nc = 3 * (size_t)levels[level + 1]->n;
    DBuf<T> a, b;
    a.reserve(nf), b.reserve(nc);
    HOT_HIP(hipMemcpyAsync(a.p, fine, nf * sizeof(T), hipMemcpyDefault, stream));
    restrict_dev(level, a.p, b.p);
    if (halo_mode()) gather_all(*levels[level + 1], b.p);
    download(coarse, b.p, nc);
    sync();
}
template <class T>
void Ctx<T>::prolong(int32_t level, const void* coarse, void* fine)
{
    need(level >= 0 && level + 1 < (int)levels.size(), "level out of range");
    size_t nf = 3 * (size_t)levels[level]->n, nc = 3 * (size_t)levels[level + 1]->n;
    DBuf<T> a, b;
    a.reserve(nc), b.reserve(nf);
    HOT_HIP(hipMemcpyAsync(a.p, coarse, nc * sizeof(T), hipMemcpyDefault, stream));
    prolong_dev(level, a.p, b.p);
    if (halo_mode()) gather_all(*levels[level], b.p);
    download(fine, b.p, nf);
    sync();
}
template <class T>
void Ctx<T>::smooth(int32_t level, int32_t kind, int32_t iterations, double tol, void* u, void* r, const void* r0)
{
    need(level >= 0 && level < (int)levels.size() && levels[level]->built, "hot_smooth: level not built (hot_build_mg)");
    Level<T>& L = *levels[level];
    size_t n3 = 3 * (size_t)L.n;
    DBuf<T> du_, dr_, r0_;
    du_.reserve(n3), dr_.reserve(n3), r0_.reserve(n3);
    HOT_HIP(hipMemcpyAsync(r0_.p, r0 ? r0 : r, n3 * sizeof(T), hipMemcpyDefault, stream)); // r may be overwritten below
    with_gs_retry([&] {
        HOT_HIP(hipMemcpyAsync(du_.p, u, n3 * sizeof(T), hipMemcpyDefault, stream));
        HOT_HIP(hipMemcpyAsync(dr_.p, r, n3 * sizeof(T), hipMemcpyDefault, stream));
        HOT_HIP(hipMemcpyAsync(L.initialResidual.p, r0_.p, n3 * sizeof(T), hipMemcpyDeviceToDevice, stream));
        smooth_dev(level, kind, iterations, (T)tol, du_.p, dr_.p, L.du.p, L.dAu.p);
        if (halo_mode()) gather_all(L, du_.p), gather_all(L, dr_.p);
        sync();
    });
    download(u, du_.p, n3);
    download(r, dr_.p, n3);
    sync();
}
template <class T>
void Ctx<T>::vcycle(const void* in, void* out)
{
    need(!levels.empty() && levels[0]->built, "hot_vcycle before hot_build_mg");
    size_t n3 = 3 * (size_t)Nn;
    HOT_HIP(hipMemcpyAsync(work0.p, in, n3 * sizeof(T), hipMemcpyDefault, stream));
    with_gs_retry([&] {
        vcycle_dev(work0.p, work1.p);
        if (halo_mode()) gather_all(*levels[0], work1.p);
        sync();
    });
    download(out, work1.p, n3);
    sync();
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
