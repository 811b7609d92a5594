// ORACLE (test infrastructure only — never linked into the product library; only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may load liboracle.so).
//
// CPU restatement of the HOT per-timestep hot path, reproducing the reference's data structures and its
// TBB parallel decomposition with OpenMP (8-colour SPGrid-block passes for every scatter, block loops for
// gathers, row-parallel ELL SpMV, block-parallel / in-block-sequential symmetric GS; sections that are
// serial in the reference are serial here).  Every member cites the reference lines it follows.
//
// This file: particle storage, sortParticlesAndPolluteGrid, the sparse block grid, kernel iteration,
// P2G, DOF numbering, mass vector, BCs, G2P.
#pragma once
#include <functional>
#include <limits>
#include <stdexcept>
#include <memory>
#include "../include/hot_mi355x.h"
#include "spgrid_index.hpp"
#include "corotated.hpp"
#include <algorithm>
#include <array>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <omp.h>

namespace hot_oracle {

// CPU-baseline variants (bench.py cpu_baseline).  "faithful" (default, and what every parity test uses):
// the sections that are serial in the reference are serial here (group scan + Set_Page loop, getNumNodes, block memset,
// Eigen dot products / vector updates of the solvers, hierarchy build).  "fair" (HOT_ORACLE_FAIR=1 at hoto_create): the
// ones that parallelise trivially run as OpenMP loops / reductions — a stronger CPU baseline, used for timing only
// (summation orders change, so results differ from the faithful variant by round-off).
inline bool& fair_flag()
{
    static bool f = false;
    return f;
}
#define HOT_FAIR_FOR _Pragma("omp parallel for schedule(static) if (hot_oracle::fair_flag())")

// "wide sums" variant (HOT_ORACLE_WIDE=1 at hoto_create; a no-op for T = double): every node sum of a particle scatter (P2G mass /
// momentum, force, CN tolerance, Hessian blocks, matrix-free block diagonal) and every dot product / norm is accumulated in double
// and rounded to T once, instead of in T like the reference (MpmSimulationBase.cpp:611-656 adds floats into GridState<float>; the
// energy is double in the reference too, MpmForceBase.cpp:355-364).  The HIP library's fp32 build does the same by construction (its
// LDS node tiles and grid reductions are double, hot_common.h AccT / grid_sum_store), so against this variant the fp32 parity bound
// measures the kernels and not the summation type.  The default (narrow) variant stays the restatement of the reference.
inline bool& wide_flag()
{
    static bool f = false;
    return f;
}

template <class T>
struct EllMat { // reference Projects/multigrid/SquareMatrix.h:27-34
    int colsize = 0, nrows = 0;
    std::vector<int> entryCol;
    std::vector<M3<T>> entryVal;
    std::vector<M3<T>> diagonalVal, diagonalEntry, diagonalBlock;
    std::array<std::vector<std::vector<int>>, 8> coloredBlockDofs;
    std::vector<std::array<int, 3>> colorOrder;
    T lMin = (T)1e-8, lMax = (T)1e2;
    // coarseSolver 7: block incomplete Cholesky of the level (setup_ic): strictly lower blocks by stencil slot (125 per row), L_ii and
    // its inverse, the stencil neighbours' ids (-1 = absent) and the rows in factorisation order
    std::vector<M3<T>> icL, icD, icDinv;
    std::vector<int> icNbr, icOrder;
    T icShift = 0;
};

template <class T>
struct Sim {
    static constexpr int LOG2S = sizeof(T) == 4 ? 6 : 7;
    using Mask = SpMask<LOG2S>;
    static constexpr int EPB = Mask::elements_per_block;
    using TV = V3<T>;
    using TM = M3<T>;

    struct Node { // reference Lib/MPM/MpmGrid.h:15-34 (GridState: v, m, new_v, idx)
        TV v;
        T m;
        TV new_v;
        int64_t idx;
    };
    struct CollisionNode { // reference Lib/Ziran/Math/Geometry/CollisionObject.h:16-45
        int node_id;
        TM P, R, Rinv;
        bool shouldRotate;
        TV dv;
        bool has_dv;
    };

    hot_config cfg;
    std::string err;
    T dx = 0, dt = 0;
    TV gravity;

    // ---- sharded mode (hot_set_comm; include/hot_mi355x.h "one connected body over several ranks"): this rank holds a
    // contiguous range of the globally sorted particle groups.  The oracle keeps every grid-sized array replicated — node
    // scatters and the assembled matrix are summed over the ranks with all-reduces — and partitions only the coloured
    // Gauss-Seidel passes (colour-synchronous exchange), which is what the product's decomposition has to reproduce.
    hot_comm comm{};
    bool sharded() const { return comm.size > 1 && comm.allreduce; }
    static constexpr int REAL = sizeof(T) == 4 ? HOT_COMM_F32 : HOT_COMM_F64;
    void allreduce(void* buf, int64_t n, int dtype, int op = HOT_COMM_SUM)
    {
        if (sharded() && n > 0 && comm.allreduce(comm.user, buf, n, dtype, op, 0) != 0) throw std::runtime_error("hot_comm.allreduce failed");
    }
    std::vector<int32_t> particle_ids; // hot_set_particle_ids (kept for the caller; this restatement never moves particles between ranks)
    std::vector<int> block_first; // [size+1] first global block first touched by each rank's particle groups
    std::vector<std::vector<int>> level_nstart; // per level [size+1]: rank r's id prefix = nodes first touched by ranks < r
    bool partitioned(int level) const
    {
        const int minrows = comm.partition_min_rows > 0 ? comm.partition_min_rows : 4096;
        return sharded() && level < (int)level_nstart.size() && level < (int)sysmats.size() && sysmats[level].nrows >= minrows;
    }
    int owner_of(int level, int node) const // rank whose id prefix holds `node`
    {
        const auto& ns = level_nstart[level];
        int r = 0;
        while (r + 1 < comm.size && node >= ns[r + 1]) ++r;
        return r;
    }
    // Owner of a 4^3 colour block (nodes in ascending id order): the product's rule (hot_config.shard_owner; a design choice of the sharded
    // decomposition, not of the reference, which is one process), restated from hot_amd/csrc/mg_build.hip k_color_owner_keys.
    //   1  the rank that first touches the block's lowest node;
    //   0  (default) by the smoother: 2 under colour-synchronous sweeps, 1 under rank-local ones (shard_gs = 1);
    //   2  finest level: the rank whose particle range — a contiguous range of the SPGrid page order; page_split[r] = the lowest page
    //      rank r + 1 holds — contains the block's own page, if its particle tiles reach the block (or, fp64, its x-companion page), else the
    //      reaching rank nearest to it in rank order (the lower one on a tie); coarser levels: the owner of the first existing child (2 C + d,
    //      d in {-1, 0, 1}^3, x slowest) of the block's lowest node.
    std::vector<uint64_t> page_split; // [size - 1]
    std::vector<uint64_t> block_touch; // per SPGrid block of the merged list: bit r = rank r's particle tiles cover it or its x-companion
    std::vector<std::vector<uint8_t>> node_owner; // per partitioned level and node: owner of the node's colour block (compute_owners)
    int owner_of_block(int level, const std::vector<int>& blockNodes) const
    {
        if (level < (int)node_owner.size() && !node_owner[level].empty()) return node_owner[level][blockNodes.front()];
        return owner_of(level, blockNodes.front());
    }
    void compute_owners()
    {
        node_owner.clear();
        const bool page_owner = cfg.shard_owner == 2 || (cfg.shard_owner == 0 && cfg.shard_gs == 0); // 0: by the smoother (hot_mi355x.h hot_config.shard_owner)
        if (!sharded() || !page_owner || (int)page_split.size() != comm.size - 1) return;
        const int R = comm.size;
        for (int level = 0; level < (int)sysmats.size() && level < (int)level_nstart.size() && level < (int)level_coords.size(); ++level) {
            const auto& coords = level_coords[level];
            std::vector<uint8_t> own(coords.size(), 0);
            std::unordered_map<unsigned long long, int> fine; // coordinate -> id of the next finer level
            auto ckey = [](int x, int y, int z) { return ((unsigned long long)(unsigned)x << 42) | ((unsigned long long)(unsigned)y << 21) | (unsigned long long)(unsigned)z; };
            if (level > 0)
                for (int j = 0; j < (int)level_coords[level - 1].size(); ++j) fine[ckey(level_coords[level - 1][j][0], level_coords[level - 1][j][1], level_coords[level - 1][j][2])] = j;
            for (int c = 0; c < 8; ++c)
                for (const auto& blockNodes : sysmats[level].coloredBlockDofs[c]) {
                    const int i = blockNodes.front();
                    int r = owner_of(level, i); // the first-touching rank
                    const int x = coords[i][0], y = coords[i][1], z = coords[i][2];
                    if (level == 0) {
                        const uint64_t page = Mask::linear_offset(x & ~3, y & ~3, z & ~3) >> 12;
                        int hr = 0;
                        while (hr < R - 1 && page >= page_split[hr]) ++hr;
                        auto it = page2block.find(Mask::linear_offset(x & ~3, y & ~3, z & ~3));
                        if (it == page2block.end() && Mask::block_xbits == 1) it = page2block.find(Mask::linear_offset((x & ~3) + 2, y & ~3, z & ~3));
                        const uint64_t sh = it != page2block.end() && it->second < (int)block_touch.size() ? block_touch[it->second] : 0;
                        if (sh)
                            for (int d = 0; d < R; ++d) {
                                if (hr - d >= 0 && ((sh >> (hr - d)) & 1ULL)) {
                                    r = hr - d;
                                    break;
                                }
                                if (hr + d < R && ((sh >> (hr + d)) & 1ULL)) {
                                    r = hr + d;
                                    break;
                                }
                            }
                    }
                    else if (level - 1 < (int)node_owner.size()) {
                        bool found = false;
                        for (int q = 0; q < 27 && !found; ++q) {
                            const int cx = 2 * x + q / 9 - 1, cy = 2 * y + (q / 3) % 3 - 1, cz = 2 * z + q % 3 - 1;
                            if ((cx | cy | cz) < 0) continue;
                            auto f = fine.find(ckey(cx, cy, cz));
                            if (f != fine.end()) r = node_owner[level - 1][f->second], found = true;
                        }
                    }
                    for (int n : blockNodes) own[n] = (uint8_t)r;
                }
            node_owner.push_back(std::move(own));
        }
    }

    // ---- particles
    int64_t Np = 0;
    std::vector<TV> X, Vel;
    std::vector<T> mass, vol, mu, lambda, Jp;
    std::vector<TM> C, F, Fn;
    // ---- sort products (reference MpmSimulationBase.h members of the same names)
    std::vector<uint64_t> particle_sorter, particle_base_offset;
    std::vector<int> particle_order;
    std::vector<std::pair<int, int>> particle_group;
    std::vector<uint64_t> block_offset; // page ids of the particle groups
    std::vector<uint64_t> blocks; // touched pages (byte offsets) in Set_Page insertion order
    std::unordered_map<uint64_t, int> page2block;
    std::vector<std::array<int, 8>> group_nb; // block ids of the 2x2x2 pages each group scatters into
    // ---- grid
    std::vector<Node> nodes; // blocks.size()*EPB, block-major, memory order inside a block
    std::vector<double> wacc; // wide-sums variant: double shadow of the node quantity being scattered (slot-major)
    static constexpr bool IS_F32 = sizeof(T) == 4;
    bool wide() const { return IS_F32 && wide_flag(); }
    int num_nodes = 0;
    std::vector<int> dof_slot;
    std::vector<std::array<int, 3>> id2coord;
    std::vector<T> mass_matrix;
    std::vector<TV> dv, vn;
    // ---- BC
    std::vector<CollisionNode> collision_nodes;
    std::vector<int> bc_of_node;
    std::vector<double> hs_origin, hs_normal;
    std::vector<hot_collision_object> cobjs; // analytic collision objects, evaluated per node in begin_step
    // ---- particle scratch (reference MpmForceBase members scratch_gradV / scratch_vp / scratch_stress)
    std::vector<TM> scratch_gradV, scratch_stress;
    std::vector<TV> scratch_vp;
    // ---- objective state (ImplicitSolverObjective)
    double Ek = 0;
    bool updated = false;
    std::vector<TV> dv0, rhs, dRhs;
    std::vector<T> nodeCNTol;
    T max_cn_tolerance = 0; // computeCharacteristicNorm's max_tol_p (MultigridSimulation.h:128-165)
    // ---- matrices
    std::vector<EllMat<T>> sysmats, promats, resmats;
    std::vector<std::vector<std::array<int, 3>>> level_coords;
    std::vector<std::vector<TV>> mg_residuals, mg_initialResiduals, mg_sols, mg_dus, mg_dAus, mg_tmps;
    int mg_level = 0;
    std::vector<std::unique_ptr<Sim<T>>> gmg; // --baseline: the coarse MPM grids (MultigridSimulation.h `multigrids`)
    hot_stats stats;

    // =============================================================== particles
    void set_particles(int64_t n, const T* x, const T* v, const T* m, const T* c, const T* f, const T* vl, const T* mu_, const T* la, const T* jp)
    {
        Np = n;
        X.resize(n), Vel.resize(n), mass.resize(n), vol.resize(n), mu.resize(n), lambda.resize(n), Jp.resize(n), C.resize(n), F.resize(n);
        for (int64_t i = 0; i < n; ++i) {
            for (int d = 0; d < 3; ++d) X[i](d) = x[3 * i + d], Vel[i](d) = v[3 * i + d];
            mass[i] = m[i], vol[i] = vl[i], mu[i] = mu_[i], lambda[i] = la[i];
            Jp[i] = jp ? jp[i] : (T)1;
            if (c)
                std::memcpy(C[i].a, c + 9 * i, 9 * sizeof(T));
            else
                C[i] = TM::zero();
            if (f)
                std::memcpy(F[i].a, f + 9 * i, 9 * sizeof(T));
            else
                F[i] = TM::identity();
        }
    }

    // =============================================================== indexing helpers
    // B-spline base node: reference Lib/Ziran/Math/Splines/BSplines.h:16-29, MathTools.h:21-25
    static inline int int_floor(T x)
    {
        int i = (int)x;
        return i - (i > x);
    }
    static inline int base_node(T x_index_space) { return int_floor(x_index_space - (T)0.5); }

    // quadratic B-spline weights: reference BSplines.h:55-81 ; BSplineWeights MpmGrid.h:55-78
    struct Spline {
        T w[3][3], dw[3][3], one_over_dx;
        int base[3];
    };
    inline void compute_spline(const TV& Xp, Spline& s) const
    {
        s.one_over_dx = 1 / dx;
        for (int d = 0; d < 3; ++d) {
            // both roundings spelled out (see hot_common.h "The index-space coordinate"): what g++ -O3 -march=native makes of the
            // reference's `x = one_over_dx * X; floor(x - 0.5); x - base` anyway, but not left to the compiler here
            int bn = int_floor(std::fma(s.one_over_dx, Xp(d), -(T)0.5));
            s.base[d] = bn;
            T d0 = std::fma(s.one_over_dx, Xp(d), -(T)bn);
            T z = ((T)1.5 - d0);
            T z2 = z * z;
            s.w[d][0] = (T)0.5 * z2;
            T d1 = d0 - 1;
            s.w[d][1] = (T)0.75 - d1 * d1;
            T d2 = 1 - d1;
            T zz = (T)1.5 - d2;
            T zz2 = zz * zz;
            s.w[d][2] = (T)0.5 * zz2;
            s.dw[d][0] = -z;
            s.dw[d][1] = -(T)2 * d1;
            s.dw[d][2] = zz;
        }
    }

    // node slot of kernel node (i,j,k) for a particle with cell offset `base` in group g.  Equivalent to
    // Packed_Add(base, Linear_Offset(i,j,k)) + virtual-memory lookup in the reference (MpmGrid.h:286-288):
    // in-page element coordinates carry into at most the +1 page per axis.
    inline int node_slot(int g, uint64_t base, int i, int j, int k) const
    {
        constexpr int xb = Mask::block_xbits, yb = Mask::block_ybits, zb = Mask::block_zbits;
        int e = (int)((base & 0xfff) >> Mask::data_bits);
        int ez = e & ((1 << zb) - 1), ey = (e >> zb) & ((1 << yb) - 1), ex = (e >> (zb + yb)) & ((1 << xb) - 1);
        int nx = ex + i, ny = ey + j, nz = ez + k;
        int ox = nx >> xb, oy = ny >> yb, oz = nz >> zb;
        nx &= (1 << xb) - 1, ny &= (1 << yb) - 1, nz &= (1 << zb) - 1;
        int elem = (nx << (yb + zb)) | (ny << zb) | nz;
        return group_nb[g][ox * 4 + oy * 2 + oz] * EPB + elem;
    }

    // reference MpmGrid::iterateKernel (MpmGrid.h:245-296): i outermost; w_ijk, grad w (already / dx)
    template <class OP>
    inline void iterate_kernel(const Spline& s, int g, uint64_t base, const OP& op)
    {
        T one_over_dx = s.one_over_dx;
        int coord[3];
        for (int i = 0; i < 3; ++i) {
            T wi = s.w[0][i];
            T dwidxi = one_over_dx * s.dw[0][i];
            coord[0] = s.base[0] + i;
            for (int j = 0; j < 3; ++j) {
                T wj = s.w[1][j];
                T wij = wi * wj;
                T dwijdxi = dwidxi * wj;
                T dwijdxj = wi * one_over_dx * s.dw[1][j];
                coord[1] = s.base[1] + j;
                for (int k = 0; k < 3; ++k) {
                    coord[2] = s.base[2] + k;
                    T wk = s.w[2][k];
                    T wijk = wij * wk;
                    TV dw{ { dwijdxi * wk, dwijdxj * wk, wij * one_over_dx * s.dw[2][k] } };
                    op(coord, wijk, dw, nodes[node_slot(g, base, i, j, k)]);
                }
            }
        }
    }

    // coloured block-parallel particle loop: reference MpmSimulationBase.h:251-264 and every
    // "for color ... tbb::parallel_for over particle_group" site (e.g. MpmSimulationBase.cpp:621-655)
    template <class OP>
    void for_each_particle_colored(const OP& op)
    {
        for (uint64_t color = 0; color < 8; ++color) {
#pragma omp parallel for schedule(dynamic, 4)
            for (int g = 0; g < (int)particle_group.size(); ++g) {
                if ((block_offset[g] & 7) != color) continue;
                for (int idx = particle_group[g].first; idx <= particle_group[g].second; ++idx) op(g, particle_order[idx]);
            }
        }
    }
    template <class OP>
    void for_each_particle_by_group(const OP& op)
    {
#pragma omp parallel for schedule(dynamic, 4)
        for (int g = 0; g < (int)particle_group.size(); ++g)
            for (int idx = particle_group[g].first; idx <= particle_group[g].second; ++idx) op(g, particle_order[idx]);
    }
    // iterateGrid (MpmGrid.h:205-243): visits nodes with idx >= 0, block-parallel
    template <class OP>
    void iterate_grid(const OP& op)
    {
#pragma omp parallel for schedule(static)
        for (int b = 0; b < (int)blocks.size(); ++b) {
            auto bc = Mask::linear_to_coord(blocks[b]);
            for (int i = 0; i < (1 << Mask::block_xbits); ++i)
                for (int j = 0; j < (1 << Mask::block_ybits); ++j)
                    for (int k = 0; k < (1 << Mask::block_zbits); ++k) {
                        int e = (i << (Mask::block_ybits + Mask::block_zbits)) | (j << Mask::block_zbits) | k;
                        Node& g = nodes[(size_t)b * EPB + e];
                        if (g.idx >= 0) {
                            int node[3] = { bc[0] + i, bc[1] + j, bc[2] + k };
                            op(node, g);
                        }
                    }
        }
    }

    // =============================================================== sortParticlesAndPolluteGrid
    // reference Lib/MPM/MpmSimulationBase.cpp:1066-1137
    int sort_particles()
    {
        constexpr int index_bits = 32 - Mask::block_bits;
        if (Np >= (1LL << index_bits)) {
            err = "particle count exceeds 2^(32-block_bits)";
            return HOT_ERR_CAPACITY;
        }
        particle_base_offset.resize(Np), particle_sorter.resize(Np), particle_order.resize(Np);
        T one_over_dx = (T)1 / dx;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < Np; ++i) {
            int b[3];
            for (int d = 0; d < 3; ++d) b[d] = int_floor(std::fma(one_over_dx, X[i](d), -(T)0.5));
            uint64_t offset = Mask::linear_offset(b[0], b[1], b[2]);
            particle_sorter[i] = ((offset >> Mask::data_bits) << index_bits) + (uint64_t)i;
        }
        parallel_sort(particle_sorter);

        particle_group.clear();
        block_offset.clear();
        int last_index = 0;
        for (int64_t i = 0; i < Np; ++i)
            if (i == Np - 1 || (particle_sorter[i] >> 32) != (particle_sorter[i + 1] >> 32)) {
                particle_group.push_back(std::make_pair(last_index, (int)i));
                block_offset.push_back(particle_sorter[i] >> 32);
                last_index = (int)i + 1;
            }
        // page map (serial in the reference): Set_Page on the page and its 2x2x2 upper neighbours,
        // a page is appended to the block list the first time it is set (SPGrid_Page_Map.h:61-70)
        blocks.clear();
        page2block.clear();
        auto set_page = [&](uint64_t offset) {
            uint64_t page = (offset >> 12) << 12;
            if (page2block.find(page) == page2block.end()) {
                page2block[page] = (int)blocks.size();
                blocks.push_back(page);
            }
        };
        if (fair_flag()) {
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < Np; ++i) {
                particle_order[i] = (int)(particle_sorter[i] & ((1ll << index_bits) - 1));
                particle_base_offset[particle_order[i]] = (particle_sorter[i] >> index_bits) << Mask::data_bits;
            }
        }
        for (int64_t i = 0; i < Np; ++i) {
            if (fair_flag() && !(i == Np - 1 || (particle_sorter[i] >> 32) != (particle_sorter[i + 1] >> 32))) continue;
            particle_order[i] = (int)(particle_sorter[i] & ((1ll << index_bits) - 1));
            uint64_t offset = (particle_sorter[i] >> index_bits) << Mask::data_bits;
            particle_base_offset[particle_order[i]] = offset;
            if (i == Np - 1 || (particle_sorter[i] >> 32) != (particle_sorter[i + 1] >> 32)) {
                set_page(offset);
                int x = 1 << Mask::block_xbits, y = 1 << Mask::block_ybits, z = 1 << Mask::block_zbits;
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b)
                        for (int c = 0; c < 2; ++c) set_page(Mask::packed_add(offset, Mask::linear_offset(x * a, y * b, z * c)));
            }
        }
        if (sharded()) {
            // the global block list = the ranks' lists concatenated in rank order, first occurrence kept: with shards that are
            // contiguous ranges of the global group order this is exactly the serial Set_Page order of the whole body
            std::vector<int64_t> counts(comm.size, 0);
            int64_t mine = (int64_t)blocks.size();
            if (comm.allgather(comm.user, &mine, counts.data(), sizeof(int64_t), 0) != 0) throw std::runtime_error("hot_comm.allgather failed");
            int64_t maxn = 0;
            for (auto c : counts) maxn = std::max(maxn, c);
            std::vector<uint64_t> send(maxn, 0), recv((size_t)maxn * comm.size);
            std::copy(blocks.begin(), blocks.end(), send.begin());
            if (comm.allgather(comm.user, send.data(), recv.data(), maxn * (int64_t)sizeof(uint64_t), 0) != 0) throw std::runtime_error("hot_comm.allgather failed");
            blocks.clear();
            page2block.clear();
            block_first.assign(comm.size + 1, 0);
            for (int r = 0; r < comm.size; ++r) {
                block_first[r] = (int)blocks.size();
                for (int64_t k = 0; k < counts[r]; ++k) set_page(recv[(size_t)r * maxn + k]);
            }
            block_first[comm.size] = (int)blocks.size();
            // which ranks' particle tiles cover a block (the product's tile plan, shard.hip build_tile_plan): the pages of a rank's list and, fp64, the
            // x-companion of each inside its 4^3 colour block
            block_touch.assign(blocks.size(), 0);
            for (int r = 0; r < comm.size; ++r)
                for (int64_t k = 0; k < counts[r]; ++k) {
                    const uint64_t page = (recv[(size_t)r * maxn + k] >> 12) << 12;
                    block_touch[page2block[page]] |= 1ULL << r;
                    if (Mask::block_xbits == 1) {
                        const auto c = Mask::linear_to_coord(page);
                        auto it = page2block.find(Mask::linear_offset(c[0] ^ 2, c[1], c[2]));
                        if (it != page2block.end()) block_touch[it->second] |= 1ULL << r;
                    }
                }
        }
        // neighbour table (replaces the reference's virtual-memory addressing)
        group_nb.resize(particle_group.size());
        for (size_t g = 0; g < particle_group.size(); ++g) {
            uint64_t page = block_offset[g] << 12;
            int x = 1 << Mask::block_xbits, y = 1 << Mask::block_ybits, z = 1 << Mask::block_zbits;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int c = 0; c < 2; ++c) {
                        uint64_t p = Mask::packed_add(page, Mask::linear_offset(x * a, y * b, z * c));
                        group_nb[g][a * 4 + b * 2 + c] = page2block[(p >> 12) << 12];
                    }
        }
        // memset of every touched block, idx = -1 (serial in the reference, :1126-1136)
        if (fair_flag()) {
            nodes.resize(blocks.size() * (size_t)EPB);
#pragma omp parallel for schedule(static)
            for (size_t k = 0; k < nodes.size(); ++k) nodes[k] = Node{ TV::zero(), 0, TV::zero(), -1 };
        }
        else
            nodes.assign(blocks.size() * (size_t)EPB, Node{ TV::zero(), 0, TV::zero(), -1 });
        num_nodes = 0;
        collision_nodes.clear();
        scratch_gradV.assign(Np, TM::zero());
        scratch_stress.assign(Np, TM::zero());
        scratch_vp.assign(Np, TV::zero());
        return 0;
    }

    static void parallel_sort(std::vector<uint64_t>& a)
    {
        // stands in for tbb::parallel_sort (:1087): chunk sort + pairwise merges
        int nt = omp_get_max_threads();
        size_t n = a.size();
        if (n < 1 << 14 || nt == 1) {
            std::sort(a.begin(), a.end());
            return;
        }
        int chunks = 1;
        while (chunks < nt) chunks <<= 1;
        std::vector<size_t> bounds(chunks + 1);
        for (int c = 0; c <= chunks; ++c) bounds[c] = n * c / chunks;
#pragma omp parallel for schedule(dynamic, 1)
        for (int c = 0; c < chunks; ++c) std::sort(a.begin() + bounds[c], a.begin() + bounds[c + 1]);
        for (int width = 1; width < chunks; width <<= 1) {
#pragma omp parallel for schedule(dynamic, 1)
            for (int c = 0; c < chunks; c += 2 * width)
                std::inplace_merge(a.begin() + bounds[c], a.begin() + bounds[c + width], a.begin() + bounds[std::min(c + 2 * width, chunks)]);
        }
    }

    // =============================================================== particlesToGrid
    // reference MpmSimulationBase.cpp:611-656 (particlesToGridHelper<true,false>) + :521-532
    void particles_to_grid()
    {
        const bool wd = wide();
        if (wd) wacc.assign(nodes.size() * 4, 0.0);
        for_each_particle_colored([&](int g, int i) {
            const TV& Xp = X[i];
            T m = mass[i];
            TV momentum = Vel[i] * m;
            TM Cm = C[i] * m;
            Spline s;
            compute_spline(Xp, s);
            iterate_kernel(s, g, particle_base_offset[i], [&](const int* node, T w, const TV& dw, Node& gs) {
                TV d{ { node[0] * dx - Xp(0), node[1] * dx - Xp(1), node[2] * dx - Xp(2) } };
                // velocity_delta = [C m | m v ; 0 | m] * [xi - xp ; 1] * w
                TV dvel = (Cm * d + momentum) * w;
                if (wd) {
                    double* a = &wacc[(size_t)(&gs - nodes.data()) * 4];
                    a[0] += (double)(m * w), a[1] += (double)dvel(0), a[2] += (double)dvel(1), a[3] += (double)dvel(2);
                    return;
                }
                gs.m += m * w;
                gs.v += dvel;
            });
        });
        if (wd) {
#pragma omp parallel for schedule(static)
            for (size_t s = 0; s < nodes.size(); ++s) nodes[s].m = (T)wacc[4 * s], nodes[s].v = TV{ { (T)wacc[4 * s + 1], (T)wacc[4 * s + 2], (T)wacc[4 * s + 3] } };
        }
        if (sharded()) { // sum the shards' partial node masses / momenta (every rank then numbers the same nodes)
            std::vector<T> buf(nodes.size() * 4);
            for (size_t s = 0; s < nodes.size(); ++s) buf[4 * s] = nodes[s].m, buf[4 * s + 1] = nodes[s].v(0), buf[4 * s + 2] = nodes[s].v(1), buf[4 * s + 3] = nodes[s].v(2);
            allreduce(buf.data(), (int64_t)buf.size(), REAL);
            for (size_t s = 0; s < nodes.size(); ++s) nodes[s].m = buf[4 * s], nodes[s].v = TV{ { buf[4 * s + 1], buf[4 * s + 2], buf[4 * s + 3] } };
        }
        num_nodes = get_num_nodes();
        if (sharded()) { // the ranks' page ranges: every rank's lowest page
            std::vector<uint64_t> lo(comm.size, 0);
            uint64_t mine = ~0ull;
            for (int64_t p = 0; p < Np; ++p) mine = std::min<uint64_t>(mine, particle_base_offset[p] >> 12);
            if (comm.allgather(comm.user, &mine, lo.data(), sizeof(uint64_t), 0) != 0) throw std::runtime_error("hot_comm.allgather failed");
            page_split.assign(lo.begin() + 1, lo.end());
        }
        if (sharded()) { // id prefix of every rank: the nodes of the blocks first touched by lower ranks
            level_nstart.assign(1, std::vector<int>(comm.size + 1, num_nodes));
            for (int r = 0; r <= comm.size; ++r) {
                int first = num_nodes;
                for (size_t s = (size_t)block_first[r] * EPB; s < nodes.size(); ++s)
                    if (nodes[s].idx >= 0) {
                        first = (int)nodes[s].idx;
                        break;
                    }
                level_nstart[0][r] = first;
            }
        }
        iterate_grid([&](const int*, Node& g) {
            if (g.m != 0)
                for (int d = 0; d < 3; ++d) g.v.a[d] = g.v.a[d] / g.m; // g.v /= g.m (:526)
            else
                g.v = TV::zero();
        });
        // id2coord is filled by ImplicitSolverObjective::buildMatrix in the reference (ImplicitSolver.h:474-477)
        id2coord.resize(num_nodes);
        dof_slot.resize(num_nodes);
#pragma omp parallel for schedule(static)
        for (int b = 0; b < (int)blocks.size(); ++b) {
            auto bc = Mask::linear_to_coord(blocks[b]);
            for (int e = 0; e < EPB; ++e) {
                Node& g = nodes[(size_t)b * EPB + e];
                if (g.idx >= 0) {
                    int ez = e & ((1 << Mask::block_zbits) - 1), ey = (e >> Mask::block_zbits) & ((1 << Mask::block_ybits) - 1), ex = e >> (Mask::block_zbits + Mask::block_ybits);
                    id2coord[g.idx] = { bc[0] + ex, bc[1] + ey, bc[2] + ez };
                    dof_slot[g.idx] = b * EPB + e;
                }
            }
        }
        build_mass_matrix();
    }

    // reference MpmGrid::getNumNodes (MpmGrid.h:148-161) — serial, insertion-ordered blocks, memory order
    int get_num_nodes()
    {
        if (fair_flag()) { // same numbering: per-block counts in parallel, serial prefix over the blocks, ids in parallel
            std::vector<int> base(blocks.size() + 1, 0);
#pragma omp parallel for schedule(static)
            for (size_t b = 0; b < blocks.size(); ++b) {
                int c = 0;
                for (int e = 0; e < EPB; ++e) c += nodes[b * EPB + e].m != 0;
                base[b + 1] = c;
            }
            for (size_t b = 0; b < blocks.size(); ++b) base[b + 1] += base[b];
#pragma omp parallel for schedule(static)
            for (size_t b = 0; b < blocks.size(); ++b) {
                int id = base[b];
                for (int e = 0; e < EPB; ++e)
                    if (nodes[b * EPB + e].m != 0) nodes[b * EPB + e].idx = id++;
            }
            return base[blocks.size()];
        }
        int total = 0;
        for (size_t b = 0; b < blocks.size(); ++b)
            for (int e = 0; e < EPB; ++e) {
                Node& g = nodes[b * EPB + e];
                if (g.m != 0) g.idx = total++;
            }
        return total;
    }

    // reference MpmSimulationBase.cpp:817-826
    void build_mass_matrix()
    {
        mass_matrix.resize(num_nodes);
        iterate_grid([&](const int*, Node& g) { mass_matrix[g.idx] = g.m; });
    }

    // =============================================================== BCs
    int set_bc(int nc, const int32_t* node_id, const T* P, const T* R, const T* Rinv, const uint8_t* slip, const T* dvc)
    {
        collision_nodes.resize(nc);
        for (int c = 0; c < nc; ++c) {
            CollisionNode& z = collision_nodes[c];
            if (node_id[c] < 0 || node_id[c] >= num_nodes) {
                err = "collision node id out of range";
                return HOT_ERR_INVALID;
            }
            z.node_id = node_id[c];
            std::memcpy(z.P.a, P + 9 * c, 9 * sizeof(T));
            if (R)
                std::memcpy(z.R.a, R + 9 * c, 9 * sizeof(T));
            else
                z.R = TM::identity();
            if (Rinv)
                std::memcpy(z.Rinv.a, Rinv + 9 * c, 9 * sizeof(T));
            else
                z.Rinv = TM::identity();
            z.shouldRotate = slip ? slip[c] != 0 : false;
            z.has_dv = dvc != nullptr;
            if (dvc)
                for (int d = 0; d < 3; ++d) z.dv(d) = dvc[3 * c + d];
        }
        return 0;
    }
    // device-side convenience of the product, restated: static STICKY half spaces.  Follows
    // AnalyticCollisionObject::multiObjectCollision for STICKY objects (CollisionObject.cpp:107-148):
    // normal_basis = I  =>  P = 0, vi = collider velocity = 0.
    void eval_halfspaces()
    {
        if (hs_origin.empty()) return;
        collision_nodes.clear();
        for (int n = 0; n < num_nodes; ++n) {
            bool inside = false;
            for (size_t h = 0; h < hs_origin.size() / 3; ++h) {
                double s = 0;
                for (int d = 0; d < 3; ++d) s += ((double)((T)id2coord[n][d] * dx) - hs_origin[3 * h + d]) * hs_normal[3 * h + d];
                if (s <= 0) inside = true;
            }
            if (inside) {
                CollisionNode z;
                z.node_id = n;
                z.P = TM::zero();
                z.R = z.Rinv = TM::identity();
                z.shouldRotate = false;
                z.has_dv = false;
                collision_nodes.push_back(z);
            }
        }
    }

    // ---- analytic collision objects: the collision query of buildInitialDvAndVnForNewton (MpmSimulationBase.cpp:1139-1184)
    // with the object transform x = R s X + b and its rates (CollisionObject.h:63-69)
    // AnalyticCollisionObject::detectAndResolveCollision (Lib/Ziran/Math/Geometry/CollisionObject.cpp:384-447) over
    // HalfSpace (AnalyticLevelSet.cpp:264-288), Sphere::queryInside (:435-452), AxisAlignedAnalyticBox (:353-363,504-529)
    // signedDistance / normal of a primitive level set as DisjointUnionLevelSet / DifferenceLevelSet call them on their members:
    // HalfSpace AnalyticLevelSet.cpp:272-288, Sphere :403-421, AxisAlignedAnalyticBox / AnalyticBox :359-363,504-529, Torus :580-608,
    // CappedCylinder AnalyticLevelSet.h:262-287 (boxes and the cylinder: distance only, their automatic-differentiation normal is not restated)
    static T signed_distance(const hot_collision_object& o, const TV& X, TV& N)
    {
        TV p0{ { (T)o.p0[0], (T)o.p0[1], (T)o.p0[2] } }, p1{ { (T)o.p1[0], (T)o.p1[1], (T)o.p1[2] } };
        N = TV::zero();
        if (o.shape == HOT_SHAPE_HALFSPACE) {
            const T nn = std::sqrt(p1.squaredNorm());
            N = TV{ { p1(0) / nn, p1(1) / nn, p1(2) / nn } };
            return N.dot(X - p0);
        }
        if (o.shape == HOT_SHAPE_SPHERE) {
            TV t = X - p0;
            const T d2 = t.squaredNorm(), dist = std::sqrt(d2);
            N = d2 < (T)1e-7 ? TV{ { 1, 0, 0 } } : TV{ { t(0) / dist, t(1) / dist, t(2) / dist } };
            return dist - p1(0);
        }
        if (o.shape == HOT_SHAPE_BOX) {
            T dd = -(T)3.4e38, q2 = 0;
            for (int k = 0; k < 3; ++k) {
                T c = (p0(k) + p1(k)) / (T)2, h = (p1(k) - p0(k)) / (T)2;
                T d = std::abs(X(k) - c) - h;
                dd = std::max(dd, d);
                T q = d < (T)0 ? (T)0 : d;
                q2 += q * q;
            }
            return std::min(dd, (T)0) + std::sqrt(q2);
        }
        double qn = std::sqrt(o.lsq[0] * o.lsq[0] + o.lsq[1] * o.lsq[1] + o.lsq[2] * o.lsq[2] + o.lsq[3] * o.lsq[3]);
        if (!(qn > 0)) qn = 1;
        const double w = o.lsq[0] / qn, qx = o.lsq[1] / qn, qy = o.lsq[2] / qn, qz = o.lsq[3] / qn;
        TM Rl;
        Rl(0, 0) = (T)(1 - 2 * (qy * qy + qz * qz)), Rl(0, 1) = (T)(2 * (qx * qy - w * qz)), Rl(0, 2) = (T)(2 * (qx * qz + w * qy));
        Rl(1, 0) = (T)(2 * (qx * qy + w * qz)), Rl(1, 1) = (T)(1 - 2 * (qx * qx + qz * qz)), Rl(1, 2) = (T)(2 * (qy * qz - w * qx));
        Rl(2, 0) = (T)(2 * (qx * qz - w * qy)), Rl(2, 1) = (T)(2 * (qy * qz + w * qx)), Rl(2, 2) = (T)(1 - 2 * (qx * qx + qy * qy));
        TV P = Rl.transpose() * (X - p0);
        if (o.shape == HOT_SHAPE_ROTATED_BOX) {
            T dd = -(T)3.4e38, q2 = 0;
            for (int k = 0; k < 3; ++k) {
                T d = std::abs(P(k)) - p1(k);
                dd = std::max(dd, d);
                T q = d < (T)0 ? (T)0 : d;
                q2 += q * q;
            }
            return std::min(dd, (T)0) + std::sqrt(q2);
        }
        T rho = std::sqrt(P(0) * P(0) + P(2) * P(2));
        if (o.shape == HOT_SHAPE_TORUS) {
            T q0 = rho - p1(0), L = std::sqrt(q0 * q0 + P(1) * P(1)), gr = q0 / L;
            N = Rl * TV{ { gr * P(0) / rho, P(1) / L, gr * P(2) / rho } };
            return L - p1(1);
        }
        T d0 = rho - p1(0), d1 = std::abs(P(1)) - (T)0.5 * p1(1);
        T m0 = std::max(d0, (T)0), m1 = std::max(d1, (T)0);
        return std::min(std::max(d0, d1), (T)0) + std::sqrt(m0 * m0 + m1 * m1);
    }
    static int members_of(const hot_collision_object& o) { return (o.shape == HOT_SHAPE_UNION || o.shape == HOT_SHAPE_DIFFERENCE) ? (int)o.p1[0] : 0; }
    static bool detect_and_resolve(const hot_collision_object* po, const TV& x, TV& v, TV& n)
    {
        const hot_collision_object& o = *po; // a composite's members are po[1 .. members_of(o)]
        TV b{ { (T)o.b[0], (T)o.b[1], (T)o.b[2] } }, p0{ { (T)o.p0[0], (T)o.p0[1], (T)o.p0[2] } }, p1{ { (T)o.p1[0], (T)o.p1[1], (T)o.p1[2] } };
        TV xb = x - b, X, N = TV::zero();
        const T one_over_s = (T)1 / (T)o.s;
        for (int k = 0; k < 3; ++k) X.a[k] = ((T)o.R[3 * k] * xb(0) + (T)o.R[3 * k + 1] * xb(1) + (T)o.R[3 * k + 2] * xb(2)) * one_over_s; // R^T (x - b) / s
        bool colliding = false;
        if (o.shape == HOT_SHAPE_UNION) { // DisjointUnionLevelSet::signedDistance / normal (AnalyticLevelSet.cpp:148-190), AnalyticLevelSet::queryInside (:111-120)
            T best = std::numeric_limits<T>::max();
            for (int m = 1; m <= members_of(o); ++m) {
                TV Nm;
                T d = signed_distance(po[m], X, Nm);
                if (d < best) best = d, N = Nm;
            }
            colliding = best <= (T)0;
        }
        else if (o.shape == HOT_SHAPE_DIFFERENCE) { // DifferenceLevelSet (:220-236)
            TV Na, Nb;
            T a = signed_distance(po[1], X, Na), nb = -signed_distance(po[2], X, Nb);
            N = nb > a ? Nb * (T)-1 : Na;
            colliding = std::max(a, nb) <= (T)0;
        }
        else if (o.shape == HOT_SHAPE_HALFSPACE) {
            {
                const T nn = std::sqrt(p1.squaredNorm()); // HalfSpace stores outward_normal.normalized() (AnalyticLevelSet.cpp:111-115)
                p1 = TV{ { p1(0) / nn, p1(1) / nn, p1(2) / nn } };
            }
            T phi = p1.dot(X - p0);
            colliding = phi <= (T)0;
            N = p1;
        }
        else if (o.shape == HOT_SHAPE_SPHERE) {
            TV to_center = X - p0;
            T d2 = to_center.squaredNorm(), r2 = p1(0) * p1(0);
            if (d2 < r2) {
                colliding = true;
                T dist = std::sqrt(d2);
                if (dist < (T)1e-7)
                    N = TV{ { 1, 0, 0 } };
                else
                    N = to_center * ((T)1 / dist);
            }
        }
        else if (o.shape == HOT_SHAPE_CAPPED_CYLINDER || o.shape == HOT_SHAPE_TORUS || o.shape == HOT_SHAPE_ROTATED_BOX) {
            // CappedCylinder (AnalyticLevelSet.h:221-297) / Torus (AnalyticLevelSet.cpp:565-608): y-axis primitive behind the level
            // set's own rotation (Eigen quaternion w,x,y,z, normalised) and translation; the torus normal is the gradient the
            // reference obtains by automatic differentiation
            double qn = std::sqrt(o.lsq[0] * o.lsq[0] + o.lsq[1] * o.lsq[1] + o.lsq[2] * o.lsq[2] + o.lsq[3] * o.lsq[3]);
            if (!(qn > 0)) qn = 1;
            const double w = o.lsq[0] / qn, qx = o.lsq[1] / qn, qy = o.lsq[2] / qn, qz = o.lsq[3] / qn;
            TM Rl;
            Rl(0, 0) = (T)(1 - 2 * (qy * qy + qz * qz)), Rl(0, 1) = (T)(2 * (qx * qy - w * qz)), Rl(0, 2) = (T)(2 * (qx * qz + w * qy));
            Rl(1, 0) = (T)(2 * (qx * qy + w * qz)), Rl(1, 1) = (T)(1 - 2 * (qx * qx + qz * qz)), Rl(1, 2) = (T)(2 * (qy * qz - w * qx));
            Rl(2, 0) = (T)(2 * (qx * qz - w * qy)), Rl(2, 1) = (T)(2 * (qy * qz + w * qx)), Rl(2, 2) = (T)(1 - 2 * (qx * qx + qy * qy));
            TV P = Rl.transpose() * (X - p0);
            T rho = std::sqrt(P(0) * P(0) + P(2) * P(2));
            if (o.shape == HOT_SHAPE_ROTATED_BOX) { // AnalyticBox::signedDistancePrimitive (AnalyticLevelSet.cpp:502-522)
                T dd = -(T)3.4e38, q2 = 0;
                for (int k = 0; k < 3; ++k) {
                    T d = std::abs(P(k)) - p1(k);
                    dd = std::max(dd, d);
                    T q = d < (T)0 ? (T)0 : d;
                    q2 += q * q;
                }
                colliding = std::min(dd, (T)0) + std::sqrt(q2) <= (T)0;
            }
            else if (o.shape == HOT_SHAPE_TORUS) {
                T q0 = rho - p1(0), L = std::sqrt(q0 * q0 + P(1) * P(1));
                colliding = L - p1(1) <= (T)0;
                T gr = q0 / L;
                N = Rl * TV{ { gr * P(0) / rho, P(1) / L, gr * P(2) / rho } };
            }
            else {
                T d0 = rho - p1(0), d1 = std::abs(P(1)) - (T)0.5 * p1(1);
                T m0 = std::max(d0, (T)0), m1 = std::max(d1, (T)0);
                colliding = std::min(std::max(d0, d1), (T)0) + std::sqrt(m0 * m0 + m1 * m1) <= (T)0;
            }
        }
        else {
            T dd = -(T)3.4e38, q2 = 0;
            for (int k = 0; k < 3; ++k) {
                T c = (p0(k) + p1(k)) / (T)2, h = (p1(k) - p0(k)) / (T)2;
                T d = std::abs(X(k) - c) - h;
                dd = std::max(dd, d);
                T q = d < (T)0 ? (T)0 : d;
                q2 += q * q;
            }
            colliding = std::min(dd, (T)0) + std::sqrt(q2) <= (T)0;
        }
        if (!colliding) return false;
        const T ss = (T)o.dsdt * one_over_s, w0 = (T)o.omega[0], w1 = (T)o.omega[1], w2 = (T)o.omega[2];
        TV v_object{ { w1 * xb(2) - w2 * xb(1) + ss * xb(0) + (T)o.dbdt[0], w2 * xb(0) - w0 * xb(2) + ss * xb(1) + (T)o.dbdt[1], w0 * xb(1) - w1 * xb(0) + ss * xb(2) + (T)o.dbdt[2] } };
        { // world normal R N
            TV Nw;
            for (int k = 0; k < 3; ++k) Nw.a[k] = (T)o.R[k] * N(0) + (T)o.R[3 + k] * N(1) + (T)o.R[6 + k] * N(2);
            N = Nw;
        }
        v = v - v_object;
        if (o.type == HOT_COLLISION_STICKY)
            v = TV::zero();
        else if (o.type == HOT_COLLISION_SLIP) {
            n = N;
            T dot = v.dot(n);
            v = v - n * dot;
            if (o.friction != 0 && dot < 0) {
                T vn = std::sqrt(v.squaredNorm());
                if (-dot * (T)o.friction < vn)
                    for (int k = 0; k < 3; ++k) v.a[k] += (v.a[k] / vn) * dot * (T)o.friction;
                else
                    v = TV::zero();
            }
        }
        else { // SEPARATE
            n = N;
            T dot = v.dot(n);
            if (dot < 0) {
                v = v - n * dot;
                if (o.friction != 0) {
                    T vn = std::sqrt(v.squaredNorm());
                    if (-dot * (T)o.friction < vn)
                        for (int k = 0; k < 3; ++k) v.a[k] += (v.a[k] / vn) * dot * (T)o.friction;
                    else
                        v = TV::zero();
                }
            }
        }
        v = v + v_object;
        return true;
    }
    // RotationExtractor<T,3>::rotate (MpmSimulationBase.h:271-281): Eigen::Quaternion::setFromTwoVectors(a, e_x) as a matrix
    static TM rotate_to_x(const TV& a)
    {
        T la = std::sqrt(a.squaredNorm());
        TV v0{ { a(0) / la, a(1) / la, a(2) / la } };
        T c = v0(0), qx, qy, qz, qw;
        const T eps = sizeof(T) == 8 ? (T)1e-12 : (T)1e-5;
        if (c < (T)-1 + eps) {
            c = std::max(c, (T)-1);
            T ax1 = v0(2), ax2 = -v0(1), l = std::sqrt(ax1 * ax1 + ax2 * ax2);
            if (l < (T)1e-30) ax1 = 1, ax2 = 0, l = 1;
            T w2 = ((T)1 + c) * (T)0.5, s = std::sqrt((T)1 - w2);
            qw = std::sqrt(w2), qx = 0, qy = ax1 / l * s, qz = ax2 / l * s;
        }
        else {
            T s = std::sqrt(((T)1 + c) * (T)2), invs = (T)1 / s;
            qx = (T)0 * invs, qy = v0(2) * invs, qz = -v0(1) * invs, qw = s * (T)0.5;
        }
        T tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
        T twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        TM Rm;
        Rm(0, 0) = 1 - (tyy + tzz), Rm(0, 1) = txy - twz, Rm(0, 2) = txz + twy;
        Rm(1, 0) = txy + twz, Rm(1, 1) = 1 - (txx + tzz), Rm(1, 2) = tyz - twx;
        Rm(2, 0) = txz - twy, Rm(2, 1) = tyz + twx, Rm(2, 2) = 1 - (txx + tyy);
        return Rm;
    }
    // multiObjectCollision with wn (CollisionObject.cpp:107-148) + the CollisionNode construction (:1153-1171)
    void eval_collision_objects()
    {
        if (cobjs.empty()) return;
        collision_nodes.clear();
        std::vector<TV> gv(num_nodes);
        iterate_grid([&](const int*, Node& g) { gv[(size_t)g.idx] = g.v; });
        for (int id = 0; id < num_nodes; ++id) { // serial: the reference pushes into a concurrent_vector (order irrelevant)
            TV xi{ { (T)id2coord[id][0] * dx, (T)id2coord[id][1] * dx, (T)id2coord[id][2] * dx } };
            TV old_v = gv[id], vi = gv[id], wn = TV::zero();
            TM nb = TM::zero();
            bool any = false;
            int slip_count = 0;
            for (size_t ko = 0; ko < cobjs.size(); ko += 1 + members_of(cobjs[ko])) { // a composite's members ride along behind it
                const auto& o = cobjs[ko];
                TV n = TV::zero();
                bool collide = detect_and_resolve(&cobjs[ko], xi, vi, n);
                any = any || collide;
                if (!collide) continue;
                if (o.type == HOT_COLLISION_STICKY) {
                    wn = TV::zero();
                    nb = TM::identity();
                    break;
                }
                for (int c = 0; c < slip_count; ++c) {
                    TV n_old{ { nb(0, c), nb(1, c), nb(2, c) } };
                    T dot = n_old.dot(n);
                    n = n - n_old * dot;
                }
                wn = n;
                T len = std::sqrt(n.squaredNorm());
                if (len) {
                    for (int k = 0; k < 3; ++k) nb(k, slip_count) = n(k) / len;
                    if (++slip_count == 3) break;
                }
            }
            if (!any) continue;
            CollisionNode z;
            z.node_id = id;
            z.P = TM::identity() - nb * nb.transpose();
            z.shouldRotate = wn(0) != 0 || wn(1) != 0 || wn(2) != 0;
            z.R = z.shouldRotate ? rotate_to_x(wn) : TM::identity();
            z.Rinv = inverse(z.R);
            z.has_dv = true;
            z.dv = vi - old_v;
            collision_nodes.push_back(z);
        }
    }

    // reference MultigridSimulation::startBackwardEuler (MultigridSimulation.h:167-186) +
    // buildInitialDvAndVnForNewton (MpmSimulationBase.cpp:1139-1184) + backupStrain (FBasedMpmForceHelper.cpp:24-33)
    void begin_step(T dt_)
    {
        dt = dt_;
        eval_halfspaces();
        eval_collision_objects();
        dv.resize(num_nodes), vn.resize(num_nodes);
        bc_of_node.assign(num_nodes, -1);
        for (size_t c = 0; c < collision_nodes.size(); ++c) bc_of_node[collision_nodes[c].node_id] = (int)c;
        iterate_grid([&](const int*, Node& g) {
            int id = (int)g.idx;
            int c = bc_of_node[id];
            if (c >= 0)
                dv[id] = collision_nodes[c].has_dv ? collision_nodes[c].dv : g.v * (T)-1;
            else
                dv[id] = gravity * dt;
            vn[id] = g.v;
        });
        Fn = F;
        // resetLSFlag (ImplicitSolver.h:277-282)
        updated = false;
        dv0 = dv;
    }

    // project lambda: reference MultigridSimulation.h:105-124
    void project(std::vector<TV>& v) const
    {
        bool slipmode = cfg.systemBCProject && cfg.boundaryType == 1;
        for (const auto& z : collision_nodes) {
            if (slipmode) {
                if (z.shouldRotate)
                    v[z.node_id](0) = 0;
                else
                    v[z.node_id] = TV::zero();
            }
            else
                v[z.node_id] = z.P * v[z.node_id];
        }
    }
    // reference ImplicitSolver.h:106-125
    void recover_solution(std::vector<TV>& ddv) const
    {
        if (cfg.systemBCProject && cfg.boundaryType == 1)
            for (const auto& z : collision_nodes)
                if (z.shouldRotate) ddv[z.node_id] = z.Rinv * ddv[z.node_id];
    }
    void transform_residual(std::vector<TV>& r) const
    {
        if (cfg.systemBCProject && cfg.boundaryType == 1)
            for (const auto& z : collision_nodes)
                if (z.shouldRotate) r[z.node_id] = z.R * r[z.node_id];
    }

    // =============================================================== gridToParticles
    // reference MpmSimulationBase.cpp:891-901 (constructNewVelocityFromNewtonResult), :930-1042
    // (gridToParticlesHelper<true,false,false>), evolveStrain (FBasedMpmForceHelper.cpp:99-114),
    // applyPlasticity (:1044-1064)
    int grid_to_particles(double dt_)
    {
        F = Fn; // force->restoreStrain(), MultigridSimulation.h:231
#pragma omp parallel for schedule(static)
        for (size_t s = 0; s < nodes.size(); ++s) nodes[s].new_v = TV::zero();
        iterate_grid([&](const int*, Node& g) { g.new_v = g.v + dv[g.idx]; });
        T D_inverse = (T)4 / (dx * dx); // MpmSimulationBase.cpp:113-118
        T r = (T)cfg.apic_rpic_ratio;
        int flags = 0;
        T dtT = (T)dt_;
#pragma omp parallel for schedule(dynamic, 4) reduction(| : flags)
        for (int g = 0; g < (int)particle_group.size(); ++g)
            for (int idx = particle_group[g].first; idx <= particle_group[g].second; ++idx) {
                int i = particle_order[idx];
                TV& Xp = X[i];
                TV picV = TV::zero();
                Spline s;
                compute_spline(Xp, s);
                TM Bp = TM::zero();
                TM gradVp = TM::zero();
                iterate_kernel(s, g, particle_base_offset[i], [&](const int* node, T w, const TV& dw, Node& gs) {
                    picV += gs.new_v * w;
                    TV d{ { node[0] * dx - Xp(0), node[1] * dx - Xp(1), node[2] * dx - Xp(2) } };
                    Bp += outer(gs.new_v * w, d);
                    gradVp += outer(gs.new_v, dw);
                });
                scratch_gradV[i] = gradVp;
                Vel[i] = picV;
                TM CC = Bp * D_inverse;
                C[i] = CC * ((r + 1) * (T)0.5) + CC.transpose() * ((r - 1) * (T)0.5);
                TV increment = picV * dtT;
                Xp += increment;
                T inc = increment.squaredNorm();
                T dx2 = dx * dx;
                if (inc > dx2) flags |= 1;
                if (inc > dx2 * (T)0.25 * (T)(cfg.cfl * cfg.cfl)) flags |= 2;
            }
        if (sharded()) {
            int32_t f[2] = { flags & 1, (flags >> 1) & 1 };
            allreduce(f, 2, HOT_COMM_I32, HOT_COMM_MAX);
            flags = f[0] | (f[1] << 1);
        }
        // evolveStrain
#pragma omp parallel for schedule(static)
        for (int64_t p = 0; p < Np; ++p) F[p] = (TM::identity() + scratch_gradV[p] * dtT) * F[p];
        // applyPlasticity
        if (cfg.plasticity == 1) {
#pragma omp parallel for schedule(static)
            for (int64_t p = 0; p < Np; ++p) von_mises_project(F[p], mu[p], lambda[p], (T)cfg.yield_stress);
        }
        else if (cfg.plasticity == 2) {
#pragma omp parallel for schedule(static)
            for (int64_t p = 0; p < Np; ++p)
                snow_project(F[p], mu[p], lambda[p], Jp[p], (T)cfg.snow[0], (T)cfg.snow[1], (T)cfg.snow[2], (T)cfg.snow[3], (T)cfg.snow[4]);
        }
        return flags;
    }

    // members defined in sim_force.hpp / sim_matrix.hpp / sim_solve.hpp
    void eval_interpolant_and_gradient(const std::vector<TV>& f);
    void update_position_based_state();
    double force_total_energy();
    double total_energy();
    void update_state(const std::vector<TV>& dv_in);
    void rasterize_force(T scale, std::vector<TV>& force);
    void compute_residual(std::vector<TV>& residual);
    void evaluate_cn_tolerance();
    void matfree_multiply(const std::vector<TV>& x, std::vector<TV>& b);
    void build_matrix();
    static void build_diagonal(EllMat<T>& m, int opt);
    void setup_ic(EllMat<T>& m, const std::vector<std::array<int, 3>>& coords);
    void solve_ic(const EllMat<T>& m, const std::vector<TV>& r, std::vector<TV>& u) const;
    static void multiply(const EllMat<T>& m, const std::vector<TV>& x, std::vector<TV>& b);
    static void mark_colors(const std::vector<std::array<int, 3>>& coords, EllMat<T>& m);
    static void build_product(EllMat<T>& out, const EllMat<T>& l, const EllMat<T>& r);
    static void build_transpose(EllMat<T>& out, const EllMat<T>& l, int rowcnt);
    void build_mg();
    void estimate_2norm(EllMat<T>& A, T tol);
    int minres_solve(const std::function<void(const std::vector<TV>&, std::vector<TV>&)>& Amul, const std::function<void(const std::vector<TV>&, std::vector<TV>&)>& prec,
        std::vector<TV>& x, const std::vector<TV>& b, T relative_tolerance, T tolerance, int max_iterations);
    void scaler(const std::vector<TV>& r, std::vector<TV>& mr, const EllMat<T>& A) const;
    void smooth(int kind, int level, std::vector<TV>& u, std::vector<TV>& r, std::vector<TV>& du, std::vector<TV>& dAu, int iterations, T tolerance);
    void vcycle(const std::vector<TV>& in, std::vector<TV>& out);
    void precondition(const std::vector<TV>& in, std::vector<TV>& out);
    bool should_exit(const std::vector<TV>& residual);
    T line_search(std::vector<TV>& ddv, std::vector<TV>& residual, T alpha);
    bool lbfgs_solve();
    void compute_step(const std::vector<TV>& residual, std::vector<TV>& step);
    bool newton_solve();
    int solve();
    int advance(double dt_);
    double calculate_dt(double max_dt, double* max_speed, double* min_corner, double* max_corner);
    int advance_frame(double frame_dt, double min_dt, double max_dt, int* substeps, int* iterations_total);
};

} // namespace hot_oracle
