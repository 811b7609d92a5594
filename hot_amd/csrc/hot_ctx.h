// libhotmi355x — context: owns the HIP stream and every device-resident array of the hot path.
//
// HBM data layout:
//   particles   SoA, component-major (`a[c*Np + p]`), kept physically in SORTED order (slot == rank in the
//               reference's particle_sorter); `slot2orig[rank]` is the reference's particle_order.
//   grid        per touched SPGrid page ("block", 2x4x4 nodes fp64 / 4x4x4 fp32) one dense tile, tiles in the
//               reference's Set_Page insertion order; node slot = block*EPB + in-page element index.
//   DOF vectors TVStack layout (xyz interleaved), node ids = reference g.idx.
//   matrices    padded ELL, row-major slots, 3x3 column-major blocks (== SquareMatrix entryCol/entryVal), every
//               level uses the 125-slot stencil layout slot = (dI+2)*25+(dJ+2)*5+(dK+2), dI = row - col coords.
#pragma once
#include "hot_common.h"

namespace hot {

struct Profiler {
    struct Rec {
        int64_t calls = 0;
        double ms = 0;
    };
    bool on = false;
    std::map<std::string, Rec> recs;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    void begin(const char* name, hipStream_t s)
    {
        if (!on) return;
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (!pool.empty()) {
            ev = pool.back();
            pool.pop_back();
        }
        else {
            (void)hipEventCreate(&ev.first);
            (void)hipEventCreate(&ev.second);
        }
        (void)hipEventRecord(ev.first, s);
        pending.emplace_back(name, ev);
    }
    void end(hipStream_t s)
    {
        if (!on) return;
        (void)hipEventRecord(pending.back().second.second, s);
    }
    void count(const std::string& name) // event-free counter record (0 ms): lets bench.py relate launches to sweeps
    {
        if (on) recs[name].calls++;
    }
    void collect()
    {
        if (!on) return;
        for (auto& p : pending) {
            (void)hipEventSynchronize(p.second.second);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, p.second.first, p.second.second);
            auto& r = recs[p.first];
            r.calls++;
            r.ms += ms;
            pool.push_back(p.second);
        }
        pending.clear();
    }
};

struct CtxBase {
    void* native_comm = nullptr; // owned communicator state of hot_rccl_attach, released with the context
    void (*native_comm_free)(void*) = nullptr;
    hot_config cfg;
    std::string err;
    hipStream_t stream = nullptr;
    Profiler prof;
    hot_stats stats;
    virtual ~CtxBase()
    {
        if (native_comm && native_comm_free) native_comm_free(native_comm);
    }
    virtual void set_particles(int64_t Np, const void* X, const void* V, const void* mass, const void* C, const void* F, const void* vol, const void* mu, const void* lambda, const void* Jp) = 0;
    virtual void get_particles(void* X, void* V, void* C, void* F, void* mu, void* lambda, void* Jp) = 0;
    virtual void sort() = 0;
    virtual void get_counts(int64_t* Np, int32_t* Ng, int32_t* Nb, int32_t* Nn) = 0;
    virtual void get_indexing(int32_t* order, uint64_t* base_offset, int32_t* group, uint64_t* block_offset, uint64_t* blocks) = 0;
    virtual void p2g() = 0;
    virtual void get_grid(int32_t* id2coord, void* mass, void* v) = 0;
    virtual void set_bc(int32_t Nc, const int32_t* node_id, const void* P, const void* R, const void* Rinv, const uint8_t* slip, const void* dvc) = 0;
    virtual void set_halfspaces(int32_t n, const double* origin, const double* normal) = 0;
    virtual void set_collision_objects(int32_t n, const hot_collision_object* objs) = 0;
    virtual void begin_step(double dt) = 0;
    virtual void get_dv(void* dv) = 0;
    virtual void set_dv(const void* dv) = 0;
    virtual void update_state(const void* dv, double* energy) = 0;
    virtual void get_particle_state(void* F, void* stress, void* gradV) = 0;
    virtual void residual(void* r) = 0;
    virtual void project(void* v) = 0;
    virtual void cn_tolerance(void* tol) = 0;
    virtual void build_hessian() = 0;
    virtual void matfree_multiply(const void* x, void* y) = 0;
    virtual void build_mg() = 0;
    virtual void get_level(int32_t level, int32_t* nrows, int32_t* colsize, int32_t* id2coord) = 0;
    virtual long long get_level_nnzb(int32_t level) = 0;
    virtual long long get_level_inblock_nnzb(int32_t level) = 0;
    virtual void get_matrix(int32_t level, int32_t* entryCol, void* entryVal) = 0;
    virtual void get_prolongation(int32_t level, int32_t* entryCol, void* weight) = 0;
    virtual void spmv(int32_t level, const void* x, void* y) = 0;
    virtual void restrict_(int32_t level, const void* fine, void* coarse) = 0;
    virtual void prolong(int32_t level, const void* coarse, void* fine) = 0;
    virtual void smooth(int32_t level, int32_t kind, int32_t iterations, double tol, void* u, void* r, const void* r0) = 0;
    virtual void vcycle(const void* in, void* out) = 0;
    virtual void solve(hot_stats* st) = 0;
    virtual void g2p(double dt, int32_t* flags) = 0;
    virtual void set_comm(const hot_comm* c) = 0;
    virtual void set_particle_ids(const int32_t* ids) = 0;
    virtual void get_particle_ids(int32_t* ids) = 0;
    virtual void write_partio(const char* path) = 0;
    virtual void write_restart(const char* path) = 0;
    virtual void read_restart(const char* path) = 0;
    virtual void line_search_api(void* ddv, void* residual, double alpha, double* alpha_out) = 0;
    virtual void should_exit_api(const void* residual, int32_t* exit_now, double* scaled) = 0;
    virtual void transform_api(void* v, bool inverse) = 0;
    virtual void compute_step_api(const void* residual, void* step) = 0;
    virtual void constitutive_eval(int32_t n, const void* F, const void* mu, const void* lambda, int32_t project, void* psi, void* P, void* dPdF) = 0;
    virtual void plasticity_eval(int32_t kind, int32_t n, void* F, void* mu, void* lambda, void* Jp) = 0;
    virtual void advance(double dt, hot_stats* st) = 0;
    virtual void calculate_dt(double max_dt, double* dt, double* max_speed, double* min_corner, double* max_corner) = 0;
    virtual void advance_frame(double frame_dt, double min_dt, double max_dt, int32_t* substeps, int32_t* iterations_total, hot_stats* st) = 0;
};

CtxBase* make_ctx_f32(const hot_config& cfg);
CtxBase* make_ctx_f64(const hot_config& cfg);

} // namespace hot
