// libhotmi355x — typed context.  Member functions are defined in sort.hip / transfer.hip / force.hip /
// hessian.hip / mg_build.hip / mg_solve.hip / solve.hip and explicitly instantiated there for float and double.
#pragma once
#include "hot_ctx.h"
#include <atomic>
#include <functional>

namespace hot {

// In-block image of one colour block (k_gs_images -> k_gs_subst): everything the block's 64-row triangular solve reads of the matrix,
// premultiplied and packed in the order the substitution consumes it.  Per block: D[64][9] | D^-1[64][9] by position; per direction
// (0 forward: strictly lower in-block couplings, 1 backward: strictly upper) the entries -(D_r^-1 A_rc) column by column (column = position c
// of the block, in the direction's sweep order), inside a column by ascending row position, 9 scalars each, behind an all-zero entry 0.
// index table (16 bits): per direction idx[row][step] = the row's entry in the column walked at that step, 0 if none.
template <class T>
struct GsImg {
    static constexpr size_t hdr_elems = 2 * 64 * 9;
    // strictly lower (upper) in-block couplings of a 4^3 block under the 5^3 stencil: per axis 14 of the 16 ordered position pairs are within
    // reach 2, so (14^3 - 64) / 2 = 1340 of the 2016 node pairs can couple
    static constexpr size_t cap_entries = 1340 + 1; // + the all-zero entry 0
    static constexpr size_t per_dir = cap_entries * 9, per_block = hdr_elems + 2 * per_dir;
    static constexpr size_t idx_per_dir = 64 * 64; // 16-bit entry indices [row][step of the direction's walk] (0 = no entry in that column: the all-zero entry)
};

// Sharded runs, hot_config.shard_owner = 0: the rank whose particle range — the SPGrid pages [split[r - 1], split[r]) of the page order — contains the
// page of the 4^3 colour block around node (x, y, z) of level `level` (a level-l node sits at (x, y, z) << l on the finest grid).
template <class T>
__host__ __device__ inline int home_rank(const uint64_t* split, int R, int x, int y, int z, int level)
{
    using G = Geo<T>;
    const uint64_t page = G::linear_offset((x & ~3) << level, (y & ~3) << level, (z & ~3) << level) >> 12;
    int r = 0;
    while (r < R - 1 && page >= split[r]) ++r;
    return r;
}

// one multigrid level: system matrix in 125-slot stencil ELL + transfer tables to the next coarser level
template <class T>
struct Level {
    int n = 0; // rows (nodes)
    int id = 0; // level index
    bool built = false; // work vectors / colouring valid (set by build_mg)
    long long nnzb = 0; // structurally non-zero 3x3 blocks (for the roofline's algorithmic bytes)
    DBuf<int32_t> coord; // 3n
    DBuf<int32_t> col; // n*125   (entryCol)
    DBuf<T> val; // n*125*9 (entryVal, 3x3 column-major)
    DBuf<T> diagVal, diagInv; // n*9: D_i and the scaler (inverse of entries or of the block, by Ainv)
    DBuf<T> diagBlockInv; // n*9: D_i^{-1} (GS always uses the block inverse)
    // colouring (markColors): ckey = colour<<28 | blockId<<7 | indexInBlock(1-based), gs_order = nodes sorted by ckey
    DBuf<uint32_t> ckey;
    DBuf<int32_t> gs_order; // n
    DBuf<int32_t> gs_block_start; // nblocks+1 : offsets into gs_order
    // after build_mg every row's slots are regrouped as [nl entries preceding the row in the GS order | diagonal |
    // nu entries following it | structural zeros], so each half sweep streams only the half it needs
    double lMin = 1e-8, lMax = 1e2; // spectrum bounds for the Chebyshev smoother (SquareMatrix.h:37, estimate2norm :375-475)
    DBuf<T> apv; // n*64*9: A*P of this level (coarse window per row, packed: the na nb nc = 27 .. 64 slots that can be non-zero come first, k_ap), kept for the coarse-correction residual update
    DBuf<int32_t> apc; // n*64: coarse column by geometric window position 16 a + 4 b + c (0 where the coarse node does not exist: its block is 0; -1 at the structurally zero positions, which apv does not store)
    DBuf<int32_t> gs_nbr; // nblocks*26: the adjacent colour blocks (global block id | colour << 28, or -1): whose unknowns a block's rows read
    DBuf<int> gs_flag; // 4*nblocks: sweep number in which the (block, sub-block) was last finished (k_gs_sweep's point-to-point hand-off)
    DBuf<int32_t> gs_pad; // nblocks*64*8 (+ one sentinel record): per (colour block, position) {node or -1, the row's four class counts, first forward slot, first backward slot, pad}: the GS kernels' header in one load
    DBuf<int32_t> gs_col; // n*125: col after the regrouping with in-block columns replaced by -1 - (position in the colour block): k_gs_block2 tells triangle entries from gathers without fetching ckey[j]
    DBuf<T> gs_img; // nblocks * GsImg<T>::per_block: premultiplied in-block couplings in the order k_gs_subst consumes them (k_gs_images, mg_build.hip)
    DBuf<uint16_t> gs_imgi; // nblocks * 2 * GsImg<T>::idx_per_dir: the images' entry index of every (row, step)
    DBuf<T> gs_p1; // 3 per slot: the off-block products summed over the slot's (up to 16) entries (k_gs_offblock -> k_gs_subst, which subtracts a row's slots from its rhs in order)
    DBuf<int2> gs_slot; // {first stored entry (row * 125 + k), entries} per slot (k_gs_slot_fill); gs_pad[8 pos + 5 / 6]: the position's first forward / backward slot
    int gs_nslot = 0;
    int32_t gs_slot_rng[2][2][8] = {}; // [forward / backward][first / end][colour]: the off-block slots of the colour's blocks — on a row-partitioned level of the blocks THIS rank owns
    DBuf<int32_t> gs_rowpn; // n: of a row's two off-block runs, the entries in the colour swept just before the row's own: forward | backward << 16 (k_gs_split_rows)
    DBuf<int4> gs_srec; // one rank (k_gs_colour): nblocks*64 + 1: per (colour block, position) the first slot in each of the four lists {forward older, forward previous-colour, backward older, backward previous-colour} (k_gs_slot_fill2)
    int32_t gs_slot_rng2[4][2][8] = {}; // [list][first / end][colour]
    bool gs_fused_ready = false; // the slots are the four lists of k_gs_colour (else the two of k_gs_offblock)
    bool gs_img_ready = false;
    long long gs_img_shift[8] = {}; // per colour: image index of a block = its block id + this (row-partitioned level: only the blocks this rank owns — one contiguous run per colour — have images; else 0)
    DBuf<T> gs_w; // chained levels (k_gs_sweep<.., WINV>): nblocks * 2 * 9 * 2017: (I - N)^-1 - I of every colour block's in-block triangle, forward / backward (k_gs_winv, mg_build.hip)
    bool gs_w_ready = false;
    DBuf<int32_t> rowcnt; // 4n: (precede-off, precede-in, follow-in, follow-off) slot counts of the regrouped rows
    bool split = false;
    // coarseSolver 7 (mg_ic.hip): block incomplete Cholesky of a top level.  ic_l: the strictly lower blocks by stencil slot; (ic_col, ic_val)
    // the factor as a sweepable matrix (lower: L_ij, upper: L_ji^T), regrouped like (col, val) with its own class counts / block headers
    DBuf<int32_t> ic_col, ic_rowcnt, ic_pad;
    DBuf<T> ic_val, ic_l, ic_d, ic_dinv, ic_dinvT;
    bool ic_ready = false;
    double ic_shift = 0;
    int color_block_begin[9] = { 0 }; // blocks of colour c are [color_block_begin[c], color_block_begin[c+1])
    int nblocks = 0;
    // prolongation to this level from the next coarser one (P has 8 slots/row), restriction = P^T as child table
    DBuf<int32_t> pcol; // n*8  coarse ids (padded slots repeat slot 0, weight 0)
    DBuf<T> pw; // n*8
    DBuf<int32_t> child; // ncoarse*27 fine ids of the 3x3x3 children of each coarse node (-1 = absent), on the COARSE level object
    // coordinate -> id map of this level
    DBuf<uint64_t> hkeys;
    DBuf<unsigned long long> hrank;
    DBuf<int32_t> hid;
    HashMap map;
    // V-cycle work vectors (3n each)
    DBuf<T> residual, initialResidual, sol, du, dAu, tmp;
    // ---- sharded solve (hot_set_comm): row ownership of this level
    bool colored = false; // mark_colors ran for this level (sharded: already while the Hessian is built)
    bool part = false; // rows are partitioned over the ranks (else every rank computes every row: small coarse levels)
    std::vector<int> nstart; // [ranks + 1] id prefixes: rank r's particles first touch the nodes [nstart[r], nstart[r+1])
    DBuf<uint8_t> owner; // n: owning rank of every row = rank whose prefix holds the lowest node of the row's 4^3 colour block
    DBuf<uint8_t> own; // n: owner == this rank (row mask of the operator kernels)
    DBuf<uint8_t> block_owner; // [nblocks] sharded: owner of every colour block (mark_colors orders a colour's blocks by owner, then first touch)
    std::vector<uint8_t> block_owner_h;
    std::vector<int> csplit; // [8 * (ranks + 1)] colour c: blocks color_block_begin[c] + [csplit[c][r], csplit[c][r+1]) belong to rank r
    std::vector<int> xbeg, xcnt; // [ranks * 8] position range in gs_order of the nodes rank r owns of colour c
    DBuf<int32_t> dxtab; // the same two tables on the device (xbeg | xcnt)
    int xmax_full = 0, xmax_col[8] = { 0 }; // largest per-rank counts (padded all-gather slots)
    const uint8_t* mask() const { return part ? own.p : nullptr; }
    // hot_config.shard_gs = 2 (l1-scaled rank-local GS): the smoother's own diagonal blocks D' = D + diag(l1 norms of the row's off-rank couplings),
    // their inverses, and E = D' - D (3 per row) for the residual identity (k_l1_diag, mg_build.hip)
    DBuf<T> gsD, gsDinv, gsE;
    bool l1 = false;
    const T* gs_d() const { return l1 ? gsD.p : diagVal.p; }
    const T* gs_dinv() const { return l1 ? gsDinv.p : diagBlockInv.p; }
    // ---- halo mode (hot_config.shard_replicated == 0): the entries of a DOF vector of this level this rank reads but does not own, and
    // who owns them; both lists are ordered by (owner | reader, colour, position in gs_order), so one colour of a GS sweep is a sub-range
    struct Halo {
        bool built = false;
        std::vector<int64_t> scnt, soff, rcnt, roff; // [ranks] nodes this rank sends to / receives from every peer, offsets into send / recv
        std::vector<int64_t> scol, rcol; // [ranks * 9] colour segment offsets inside a peer's range
        DBuf<int32_t> send, recv; // node ids
        DBuf<uint8_t> need; // n: 1 = read here, owned elsewhere
        int64_t stot = 0, rtot = 0;
    } halo;
};

template <class T>
struct Ctx : CtxBase {
    using G = Geo<T>;
    static constexpr int EPB = G::EPB;
    static constexpr int TX = G::BX + 2, TY = G::BY + 2, TZ = G::BZ + 2, TILE = TX * TY * TZ; // nodes a particle group touches

    T dx = 0, dt = 0;
    // ---- particles (sorted order)
    int64_t Np = 0;
    DBuf<T> pX, pV, pM, pC, pF, pVol, pMu, pLam, pJp, pFn, pFt, pStress, pGradV;
    DBuf<int32_t> slot2orig;
    DBuf<int32_t> pGid; // global particle id (sort-key tie break of a sharded run, travels with a migrating particle)
    void reserve_particles(int64_t n);
    void migrate_particles(); // sharded: hand every particle to the rank of its SPGrid page range (shard.hip)
    void set_particle_ids(const int32_t* ids) override;
    void get_particle_ids(int32_t* ids) override;
    DBuf<T> spare1, spare3, spare9;
    DBuf<int32_t> sparei;
    bool keep_debug = true; // store stress/gradV for hot_get_particle_state
    // ---- sort
    DBuf<uint64_t> keys, keys2;
    DBuf<uint32_t> vals, vals2;
    DBuf<char> sort_tmp;
    size_t sort_tmp_bytes = 0;
    DBuf<int32_t> flags, scan; // generic flag/scan scratch
    DBuf<char> scan_tmp;
    size_t scan_tmp_bytes = 0;
    // ---- groups / blocks
    int Ng = 0, Nb = 0, Nn = 0;
    DBuf<int32_t> group_first; // Ng+1
    DBuf<uint64_t> group_page; // Ng page ids
    DBuf<int32_t> group_nb; // Ng*8
    DBuf<int32_t> group_origin; // Ng*3 node coords of the page's first node
    DBuf<uint64_t> blocks; // Nb page byte offsets
    DBuf<uint64_t> bh_keys;
    DBuf<unsigned long long> bh_rank;
    DBuf<int32_t> bh_id;
    HashMap block_map;
    // ---- per-cell particle ranges (particles are sorted by page, then by base cell inside the page)
    int Ncell = 0;
    DBuf<int32_t> cell_first; // Ncell+1
    DBuf<int32_t> group_cell0; // Ng+1: rank of the first base cell of every particle group
    DBuf<uint64_t> ch_keys;
    DBuf<unsigned long long> ch_rank;
    DBuf<int32_t> ch_id;
    HashMap cell_map; // (Linear_Offset(base cell) >> data_bits) -> cell id
    DBuf<T> pDP; // Hessian assembly: 64*Np particle records (k_dpdf_rec, hessian_rows.hip); matrix-free diagonal: 45*Np symmetric 9x9 V_p dt^2 dP/dF
    void build_cell_table();
    void matfree_diagonal(T* dinv); // 9 Nn: inverse (Ainv) of the block diagonal of the matrix-free operator
    void assemble_tiles(Level<T>& L); // A/B build: the LDS-staged kernels of rounds 1 - 4
    void assemble_rows(Level<T>& L); // production (hessian_rows.hip)
    void build_gs_winv(Level<T>& L); // inverse images of the in-block GS triangles of a chained level (mg_solve.hip)
    DBuf<int32_t> tile_tab; // its per-tile tables: [tile][64] {first particle, count} of the base cells around the tile, [tile][8] row DOFs
    // ---- atomic-free scatter: every particle group writes its (BX+2)(BY+2)(BZ+2) partial tile, then each node sums
    //      the <= 8 partial tiles that cover it in a fixed order (deterministic; global fp64 atomics top out at ~2e10/s)
    DBuf<int32_t> block_group; // Nb: group whose page is this block, or -1
    DBuf<int32_t> block_rev; // Nb*8: group at page(b) - (a BX, b BY, c BZ), or -1
    DBuf<T> gPart; // Ng * Q * TILE
    void reduce_tiles(int Q, T* o0, T* o1, T* o2, T* o3, T* o4, const char* name);
    // ---- node tiles (Nb*EPB)
    DBuf<T> gM, gMV, gF, gCN; // gMV/gF: 3 components, component-major over slots
    DBuf<int32_t> gIdx;
    DBuf<int32_t> tileDof; // Ng * (BX+2)(BY+2)(BZ+2): DOF id (or -1) of every node of a particle group's tile, so that the per-trial
                           // state pass gathers vn + dv after one index load instead of the nb8 -> gIdx -> value chain
    DBuf<int32_t> block_count; // Nb+1
    // ---- DOFs (Nn)
    DBuf<int32_t> dofSlot, id2coord, bcIdx;
    DBuf<T> mass, vn, dv, dv0, cnTol, nodeV;
    // ---- BC (Nc)
    int Nc = 0;
    DBuf<int32_t> bcNode;
    DBuf<T> bcP, bcR, bcRinv, bcDv;
    DBuf<uint8_t> bcSlip, bcHasDv;
    std::vector<double> hs_origin, hs_normal;
    std::vector<hot_collision_object> cobjs; // analytic collision objects (hot_set_collision_objects)
    DBuf<char> d_cobjs;
    DBuf<double> d_hs;
    // ---- objective
    double Ek = 0, Ek_sigma = 0; // incremental potential at the current iterate: the reference's formula, and with psi summed from the singular values (state_pass)
    bool updated = false;
    int ls_prev_trials = 1; // trials the previous line search of this step took (hot_config.ls_energy_only = 0 starts energy-only after a search that halved)
    T max_cn_tolerance = 0;
    DBuf<T> rhs, work0, work1, work2, work3, solve_keep;
    // ---- sharded solve: one connected body over several ranks (include/hot_mi355x.h hot_comm)
    hot_comm comm{};
    bool sharded() const { return comm.size > 1; }
    std::vector<int> block_first; // [ranks + 1] first global block first touched by each rank's particle groups
    std::vector<uint64_t> page_split; // [ranks - 1] sharded: rank r holds the SPGrid pages [page_split[r - 1], page_split[r]) of the page order (migrate_particles)
    std::vector<int> nstart0; // [ranks + 1] level-0 id prefixes (nodes of those blocks)
    DBuf<char> xsend, xrecv; // staging of the collectives
    DBuf<uint8_t> written; // level-0 rows this rank's tile kernel has written (its partial rows)
    void set_comm(const hot_comm* c) override;
    void write_partio(const char* path) override;
    void write_restart(const char* path) override;
    void read_restart(const char* path) override;
    void c_allreduce(void* buf, int64_t n, int dtype, int op, bool on_device);
    void c_allgather(const void* send, void* recv, int64_t bytes, bool on_device);
    void c_alltoallv(const void* send, const int64_t* soff, const int64_t* sbytes, void* recv, const int64_t* roff, const int64_t* rbytes);
    void merge_block_lists(); // sort(): the ranks' first-touch block lists -> the global Set_Page order
    void color_level(Level<T>& L); // markColors of one level (mg_build.hip)
    void level_ownership(Level<T>& L); // owner / own / colour splits / exchange tables from L.nstart and the colouring
    void exchange(Level<T>& L, T* x, int colour, int ncomp = 3); // owners' entries of x (ncomp values per node; all colours: colour < 0) to every rank
    void exchange_rows(Level<T>& L, const uint8_t* touched); // partial matrix rows -> their owners, summed there
    void allreduce_tiles(T* tiles, int q); // q * Nb * EPB node-tile values, summed over the ranks
    // ---- halo mode
    bool halo_mode() const { return sharded() && !cfg.shard_replicated; }
    int64_t comm_calls = 0, comm_calls_index = 0, comm_bytes_index = 0, comm_bytes_data = 0; // since the last hot_sort
    bool comm_index_phase = false; // the collectives called now carry integers that describe the grid, not field data
    struct IndexPhase {
        Ctx<T>* c;
        bool old;
        IndexPhase(Ctx<T>* c_) : c(c_), old(c_->comm_index_phase) { c->comm_index_phase = true; }
        ~IndexPhase() { c->comm_index_phase = old; }
    };
    const char* comm_tag = "other"; // what the collectives called now carry (profile records "commMB_<tag>": calls, megabytes handed in)
    struct CommTag {
        Ctx<T>* c;
        const char* old;
        CommTag(Ctx<T>* c_, const char* t) : c(c_), old(c_->comm_tag) { c->comm_tag = t; }
        ~CommTag() { c->comm_tag = old; }
    };
    void account(int64_t bytes)
    {
        ++comm_calls, comm_calls_index += comm_index_phase ? 1 : 0, (comm_index_phase ? comm_bytes_index : comm_bytes_data) += bytes;
        if (prof.on) {
            auto& r = prof.recs[std::string("commMB_") + (comm_index_phase ? "index" : comm_tag)];
            r.calls++, r.ms += (double)bytes * 1e-6;
        }
    }
    void export_comm_stats() { stats.comm_calls = comm_calls, stats.comm_calls_index = comm_calls_index, stats.comm_bytes_index = comm_bytes_index, stats.comm_bytes_data = comm_bytes_data; }
    // node tiles: the ranks whose particle groups cover a block ("sharers") exchange their partial tiles and add them in rank order
    DBuf<uint8_t> touch; // Nb: this rank's tiles cover the block
    DBuf<uint64_t> sharers; // Nb: bit r = rank r covers the block
    DBuf<int32_t> tpos; // ranks * Nb: position of the block in the list shared with rank q, or -1
    DBuf<int32_t> tlist; // the shared-block lists, peer after peer (ascending block id)
    std::vector<int64_t> tcnt, toff; // [ranks] blocks shared with every peer, offsets into tlist
    void build_tile_plan(); // sort(): after the global block list
    void tile_exchange(T* const* arrays, int q); // q slot arrays (Nb * EPB each): summed over the sharers of every block
    void build_halo(Level<T>& L, const std::function<void(uint8_t*)>& mark_extra); // mark_extra: additional readers of the level's vectors (device flags)
    void halo_gather(Level<T>& L, T* x, int colour = -1, int ncomp = 3); // owners' values of x -> this rank's halo entries (one colour, or all)
    void gather_colour(Level<T>& L, T* x, int colour, int ncomp); // one padded all-gather: every rank gets every owner's entries of one colour (or of all)
    void gather_all(Level<T>& L, T* x, int ncomp = 3); // every rank gets every owner's entries (C ABI getters, first-generation mode)
    void replicate_numbering(); // p2g(): node coordinates of the blocks this rank does not cover, slot <-> id tables
    void level0_ownership(); // p2g(): level 0 exists (coordinates, colouring, row ownership, halo) before the first vector operation of the step
    void mark_stencil(Level<T>& L, uint8_t* need); // the 125-stencil neighbours of the rows this rank owns
    // partitioned vector algebra (sharded, halo mode): level-0 solver vectors are valid on the rows this rank owns; kernels skip the
    // other rows (vmask) and every batch of inner products is summed over the ranks once (reduce_scalars)
    const uint8_t* vmask = nullptr;
    struct MaskScope {
        Ctx<T>* c;
        const uint8_t* old;
        MaskScope(Ctx<T>* c_, const uint8_t* m) : c(c_), old(c_->vmask) { c->vmask = m; }
        ~MaskScope() { c->vmask = old; }
    };
    void reduce_scalars(double* dev, int n)
    {
        if (!vmask || n <= 0) return;
        CommTag tag(this, "scalars");
        c_allreduce(dev, n, HOT_COMM_F64, HOT_COMM_SUM, true);
    }
    static constexpr int REAL = sizeof(T) == 4 ? HOT_COMM_F32 : HOT_COMM_F64;
    DBuf<double> dscal; // device scalars
    DBuf<T> speed_part; // block maxima of calculate_dt
    DBuf<double> red_part; // grid_sum_store deposits (2 per workgroup)
    DBuf<unsigned> red_count; // its arrival counter (always 0 between launches)
    // mirror: a slot of hscal (pinned, device-visible) that receives the result too.  ticket: the launch also stamps hscal[251] with a
    // fresh number after the results, and the host waits for that stamp with wait_ticket() instead of a stream synchronisation
    GridRed gred(size_t grid, double* mirror = nullptr, bool ticket = false)
    {
        if (2 * grid > red_part.cap) {
            HOT_HIP(hipStreamSynchronize(stream)); // a launch still summing the old deposits must be done before they are freed
            red_part.reserve(2 * grid, 1.5);
        }
        GridRed g{ red_part.p, red_count.p, mirror, nullptr, 0.0 };
        if (ticket) g.ticket = hscal + 251, g.ticket_val = new_ticket();
        return g;
    }
    GridRed gred2(size_t grid, double* mirror, double* mirror1) // two sums, the second one mirrored to its own host slot
    {
        GridRed g = gred(grid, mirror);
        g.mirror1 = mirror1;
        return g;
    }
    GridRed gred_n(size_t grid, int nv) // grid_sum_store_n: nv deposits per workgroup
    {
        if ((size_t)nv * grid > red_part.cap) {
            HOT_HIP(hipStreamSynchronize(stream));
            red_part.reserve((size_t)nv * grid, 1.5);
        }
        return GridRed{ red_part.p, red_count.p, nullptr, nullptr, 0.0 };
    }
    double last_ticket = 0;
    double new_ticket() { return last_ticket += 1.0; }
    // Wait until the launch that carries the newest ticket has delivered its results to the pinned host slots.  Every earlier launch on
    // the stream is complete by then.  Spinning on host memory costs ~2 us after the store; hipStreamSynchronize wakes the thread up
    // 30-45 us after the kernel ends.  Falls back to the stream synchronisation if the stamp does not arrive (device error, stall).
    void wait_ticket()
    {
        const double want = last_ticket;
        volatile double* t = hscal + 251;
        const double t0 = wall_ms();
        for (unsigned spin = 0; *t != want; ++spin) {
            __builtin_ia32_pause();
            if ((spin & 1023u) == 1023u && wall_ms() - t0 > 20.0) {
                HOT_HIP(hipStreamSynchronize(stream));
                if (*(volatile int*)(hscal + 250) != 0) sync(); // a spinning kernel (k_gs_sweep, k_cg_persist) gave up: throws ERR_RETRY, the caller redoes the operation with launches
                HOT_CHECK(*t == want, HOT_ERR_DEVICE, "a reduction launch did not deliver its result");
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*(volatile int*)(hscal + 250) != 0) sync(); // k_gs_sweep timed out somewhere before: the usual path (throws ERR_RETRY)
    }
    DBuf<uint64_t> col_hk; // mark_colors scratch (block hash map, colour block heads)
    DBuf<unsigned long long> col_hr;
    DBuf<int32_t> col_hi, col_cb;
    bool cg_bar_dirty = false; // a spinning kernel timed out since the deposit slots were last reset (Ctx::sync): they may hold a partial phase
    DBuf<double> cg_dep; // k_cg_persist: its dot-product deposits = barrier flags, four rotating sets of two per workgroup
    unsigned cg_phase = 0; // barriers passed by the persistent solves since the slots were reset, mod 4 (which set the next launch starts with)
    int cg_G = 0; // the grid the slots are laid out for
    bool attr_cg_set = false;
    int cg_group = 2; // iterations the last fused top-level PCG took: size of the first group of launches of the next one
    int gs_epoch = 0; // sweep number, never reused inside a context
    bool attr_tiles_set = false, attr_rows_set = false, attr_gs_set = false, attr_winv_set = false; // dynamic-LDS limits raised on this context's device (hipFuncSetAttribute is per device)
    DBuf<int> gs_done; // [0,40) pass counters of k_gs_sweep (the sticky wait-timeout flag lives in pinned host memory, hscal[250])
    double* hscal = nullptr; // pinned host mirror
    // ---- L-BFGS history
    DBuf<T> hist_dx[9], hist_dg[9];
    // ---- multigrid
    std::vector<Level<T>*> levels;
    // Level objects (and their multi-GB device buffers) are recycled across time steps: hipMalloc/hipFree of the
    // 2.6 GB level-0 matrix every step costs more than assembling it
    std::vector<Level<T>*> level_pool[12];
    Level<T>* acquire_level(int id)
    {
        Level<T>* l;
        if (!level_pool[id].empty()) {
            l = level_pool[id].back();
            level_pool[id].pop_back();
        }
        else
            l = new Level<T>();
        l->id = id, l->n = 0, l->nnzb = 0, l->nblocks = 0, l->built = false, l->split = false, l->part = false, l->colored = false, l->halo.built = false, l->ic_ready = false;
        return l;
    }
    void release_levels(size_t keep = 0)
    {
        while (levels.size() > keep) {
            Level<T>* l = levels.back();
            levels.pop_back();
            level_pool[l->id < 12 ? l->id : 11].push_back(l);
        }
    }
    DBuf<T> ap; // A*P scratch (n*64*9)
    // ---- baseline geometric multigrid (--baseline): one whole grid context per coarse level (spacing 2^l dx)
    std::vector<Ctx<T>*> gmg;
    DBuf<int32_t> orig2slot; // inverse of slot2orig
    Ctx<T>* build_gmg_grid(int level); // sort / mass P2G / boundaries / re-rasterised matrix of coarse level `level`

    Ctx(const hot_config& c);
    ~Ctx();
    template <class U>
    void upload(DBuf<U>& dst, const void* src, size_t n)
    {
        dst.reserve(n);
        if (n) HOT_HIP(hipMemcpyAsync(dst.p, src, n * sizeof(U), hipMemcpyDefault, stream));
    }
    template <class U>
    void download(void* dst, const U* src, size_t n)
    {
        if (dst && n) HOT_HIP(hipMemcpyAsync(dst, src, n * sizeof(U), hipMemcpyDefault, stream));
    }
    // A chained coarse-level sweep (k_gs_sweep) bounds its spin on the neighbour flags; if it ever gives up it raises the
    // flag in pinned host memory and leaves a half-updated iterate behind.  The next sync() then switches this context to the
    // launch-per-pass path for good and throws ERR_RETRY, which the operations that can contain such a sweep (solve, vcycle,
    // smooth) catch to redo themselves from their saved inputs — the context is never left poisoned.
    // A time-out is not for life: one transient event (another process holding compute units while a spinning kernel waited) must not cost a
    // long-running context its chained sweeps and persistent PCG for good.  After REARM_STEPS clean time steps the chained path is tried again;
    // the third time-out on a context is final.
    static constexpr int ERR_RETRY = -100; // internal, never crosses the C ABI
    static constexpr int REARM_STEPS = 32, MAX_TIMEOUTS = 3;
    bool gs_no_chain = false, gs_chain_timed_out = false;
    int gs_timeouts = 0, steps_since_timeout = 0;
    void rearm_chain() // hot_begin_step
    {
        if (!gs_chain_timed_out || gs_timeouts >= MAX_TIMEOUTS || ++steps_since_timeout < REARM_STEPS) return;
        gs_chain_timed_out = false, steps_since_timeout = 0;
        if (!sharded()) gs_no_chain = false; // (several ranks: launch-per-pass sweeps whatever happened, set_comm)
    }
    int n_cu = 0; // compute units of the device (persistent kernels size their grids by it)
    int device_cus()
    {
        if (n_cu == 0) {
            int dev = 0;
            HOT_HIP(hipGetDevice(&dev));
            HOT_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        return n_cu;
    }
    int unset_level = -1; // the level whose GS forward target (Level::tmp) carries "not written yet" marks from the kernel launched last on it (restrict_dev / vcycle_dev -> smooth_dev)
    int retry_scope = 0, fake_syncs = 0; // inside with_gs_retry; A/B build: synchronisations counted for HOT_GS_FAKE_TIMEOUT
    void sync()
    {
        HOT_HIP(hipStreamSynchronize(stream));
        // A/B build: HOT_GS_FAKE_TIMEOUT=n raises the time-out flag at the n-th synchronisation inside an operation that can retry — the
        // redo-from-saved-inputs path then runs in a test (tests/test_gpu_variants.py) without a kernel that really hangs
        if (retry_scope > 0 && !gs_no_chain && ab_int("HOT_GS_FAKE_TIMEOUT", 0) > 0 && ++fake_syncs == ab_int("HOT_GS_FAKE_TIMEOUT", 0)) *(volatile int*)(hscal + 250) = 1;
        if (*(volatile int*)(hscal + 250) != 0) {
            *(volatile int*)(hscal + 250) = 0;
            gs_no_chain = gs_chain_timed_out = true;
            cg_bar_dirty = true; // a k_cg_persist workgroup that gave up did not re-arm the barrier counters: cleared before the next persistent launch (after rearm_chain)
            ++gs_timeouts, steps_since_timeout = 0;
            throw Error{ ERR_RETRY, "k_gs_sweep: wait on a neighbouring block timed out; redoing the operation with one launch per pass" };
        }
    }
    template <class Fn>
    void with_gs_retry(Fn&& fn)
    {
        struct Scope {
            int& n;
            Scope(int& n_) : n(n_) { ++n; }
            ~Scope() { --n; }
        } scope(retry_scope);
        try {
            fn();
        }
        catch (const Error& e) {
            if (e.code != ERR_RETRY) throw;
            fn(); // gs_no_chain is set now: no chained sweep can occur, so this cannot throw ERR_RETRY again
        }
    }
    int32_t exclusive_scan_i32(const int32_t* in, int32_t* out, size_t n); // returns total (syncs)
    void need(bool cond, const char* what) { HOT_CHECK(cond, HOT_ERR_INVALID, what); }

    // CtxBase
    void set_particles(int64_t Np, const void* X, const void* V, const void* mass, const void* C, const void* F, const void* vol, const void* mu, const void* lambda, const void* Jp) override;
    void get_particles(void* X, void* V, void* C, void* F, void* mu, void* lambda, void* Jp) override;
    void sort() override;
    void get_counts(int64_t* Np, int32_t* Ng, int32_t* Nb, int32_t* Nn) override;
    void get_indexing(int32_t* order, uint64_t* base_offset, int32_t* group, uint64_t* block_offset, uint64_t* blocks) override;
    void p2g() override;
    void get_grid(int32_t* id2coord, void* mass, void* v) override;
    void set_bc(int32_t Nc, const int32_t* node_id, const void* P, const void* R, const void* Rinv, const uint8_t* slip, const void* dvc) override;
    void set_halfspaces(int32_t n, const double* origin, const double* normal) override;
    void set_collision_objects(int32_t n, const hot_collision_object* objs) override;
    void begin_step(double dt) override;
    void get_dv(void* dv) override;
    void set_dv(const void* dv) override;
    void update_state(const void* dv, double* energy) override;
    void get_particle_state(void* F, void* stress, void* gradV) override;
    void residual(void* r) override;
    void project(void* v) override;
    void cn_tolerance(void* tol) override;
    void build_hessian() override;
    void matfree_multiply(const void* x, void* y) override;
    void build_mg() override;
    void get_level(int32_t level, int32_t* nrows, int32_t* colsize, int32_t* id2coord) override;
    void get_matrix(int32_t level, int32_t* entryCol, void* entryVal) override;
    long long get_level_nnzb(int32_t level) override
    {
        need(level >= 0 && level < (int)levels.size(), "level out of range");
        if (levels[level]->nnzb < 0) count_nnzb(*levels[level]); // one pass over the values, only when somebody asks
        return levels[level]->nnzb;
    }
    long long get_level_inblock_nnzb(int32_t level) override
    {
        need(level >= 0 && level < (int)levels.size() && levels[level]->split, "hot_get_level_inblock_nnzb: level out of range or not coloured (hot_build_mg)");
        Level<T>& L = *levels[level];
        std::vector<int32_t> rc(4 * (size_t)L.n);
        HOT_HIP(hipMemcpyAsync(rc.data(), L.rowcnt.p, rc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        sync();
        long long tot = 0;
        for (int i = 0; i < L.n; ++i) tot += rc[4 * (size_t)i + 1] + rc[4 * (size_t)i + 2]; // precede-in + follow-in (rows of other ranks hold zeros)
        return tot;
    }
    void get_prolongation(int32_t level, int32_t* entryCol, void* weight) override;
    void spmv(int32_t level, const void* x, void* y) override;
    void restrict_(int32_t level, const void* fine, void* coarse) override;
    void prolong(int32_t level, const void* coarse, void* fine) override;
    void smooth(int32_t level, int32_t kind, int32_t iterations, double tol, void* u, void* r, const void* r0) override;
    void vcycle(const void* in, void* out) override;
    void solve(hot_stats* st) override;
    void g2p(double dt, int32_t* flags) override;
    void constitutive_eval(int32_t n, const void* F, const void* mu, const void* lambda, int32_t project, void* psi, void* P, void* dPdF) override;
    void plasticity_eval(int32_t kind, int32_t n, void* F, void* mu, void* lambda, void* Jp) override;
    void advance(double dt, hot_stats* st) override;
    void calculate_dt(double max_dt, double* dt, double* max_speed, double* min_corner, double* max_corner) override;
    void advance_frame(double frame_dt, double min_dt, double max_dt, int32_t* substeps, int32_t* iterations_total, hot_stats* st) override;

    // ---- device-side building blocks (device pointers)
    void eval_halfspaces();
    void eval_collision_objects();
    void trial_batch(const T* ddv, T alpha, int K, double* Ek_out); // the energies of K line-search trials (alpha, alpha / 2, ...) from one pass: what K energy-only state_pass calls return
    double state_pass(const T* dv_in, bool want_force, bool energy_only = false); // G2P(vn+dv) -> F, energy, force scatter; returns total energy (syncs)
    void force_pass(); // force scatter from the stresses of the last state_pass
    void residual_dev(T* r); // from the force tiles of the last state_pass / force_pass
    void project_dev(T* v);
    void transform_dev(T* v, bool inverse); // transformResidual / recoverSolution
    void cn_tolerance_dev();
    void build_diagonal(Level<T>& L);
    void build_ic(Level<T>& L); // coarseSolver 7: block incomplete Cholesky of the top level (mg_ic.hip)
    void count_nnzb(Level<T>& L);
    std::string lname(const char* base, int level) { return std::string(base) + "_L" + std::to_string(level); }
    void spmv_dev(Level<T>& L, const T* x, T* y);
    void scale_dev(Level<T>& L, const T* in, T* out); // out_i = Dinv_i in_i
    void block_apply_dev(const T* D, const T* in, T* out, int n);
    void estimate_2norm(Level<T>& L, double tol);
    int minres_dev(const std::function<void(const T*, T*)>& Amul, const std::function<void(const T*, T*)>& prec, T* x, const T* b, T relative_tolerance, T tolerance, int max_iterations);
    void scal(size_t n, T a, T* x); // x *= a
    bool gs_marks_wanted(int level) const; // the level's next smoother is the chained GS sweep that takes its forward target's "not written yet" marks from the kernel before it
    void restrict_dev(int level, const T* fine, T* coarse, T* zero_coarse = nullptr);
    void prolong_dev(int level, const T* coarse, T* fine);
    void smooth_dev(int level, int kind, int iterations, T tol, T* u, T* r, T* du, T* dAu, bool final_residual = true);
    void vcycle_dev(const T* in, T* out);
    void precondition_dev(const T* in, T* out);
    void matfree_dev(const T* x, T* y);
    // vector helpers on 3n-long arrays
    void axpy(size_t n, T a, const T* x, T* y); // y += a x
    void axpy_dev(size_t n, const double* a, double sign, const T* x, T* y); // y += sign * (*a) * x, scalar on device
    void copy(size_t n, const T* x, T* y);
    void zero(size_t n, T* y);
    void dot_to(size_t n, const T* x, const T* y, double* out, double* mirror = nullptr); // *out = <x,y> (device scalar; mirror: pinned host slot that receives it too)
    double dot_host(size_t n, const T* x, const T* y);
    bool should_exit(const T* r);
    T line_search(T* ddv, T* residual_out, T alpha);
    bool lbfgs_solve();
    bool newton_solve();
    void compute_step_dev(const T* residual, T* step);
    DBuf<T> nw_step, nw_r, nw_p, nw_q, nw_t, nw_diag; // projected-Newton work vectors
    void line_search_api(void* ddv, void* residual, double alpha, double* alpha_out) override;
    void should_exit_api(const void* residual, int32_t* exit_now, double* scaled) override;
    void transform_api(void* v, bool inverse) override;
    void compute_step_api(const void* residual, void* step) override;
};

// launch helpers
inline int div_up(size_t a, size_t b) { return (int)((a + b - 1) / b); }

#define HOT_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                       \
    do {                                                                             \
        (ctx)->prof.begin(name, (ctx)->stream);                                      \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, (ctx)->stream, __VA_ARGS__); \
        (ctx)->prof.end((ctx)->stream);                                              \
    } while (0)

} // namespace hot
