// libhotmi355x — the extern "C" boundary declared in include/hot_mi355x.h.  Exceptions never cross it: every
// entry point converts hot::Error into a negative hot_status and stores the message for hot_last_error().
#include "hot_ctx.h"

struct hot_ctx {
    hot::CtxBase* impl = nullptr;
    std::string err;
};

#define HOT_API_BEGIN                                   \
    if (!ctx || !ctx->impl) return HOT_ERR_INVALID;     \
    try {                                               \
        (void)hipSetDevice(ctx->impl->cfg.device);
#define HOT_API_END                                     \
        return HOT_OK;                                  \
    }                                                   \
    catch (const hot::Error& e) {                       \
        ctx->err = e.msg;                               \
        return e.code;                                  \
    }                                                   \
    catch (const std::exception& e) {                   \
        ctx->err = e.what();                            \
        return HOT_ERR_INVALID;                         \
    }                                                   \
    catch (...) {                                       \
        ctx->err = "unknown exception";                 \
        return HOT_ERR_INVALID;                         \
    }

extern "C" {

void hot_default_config(hot_config* c)
{
    std::memset(c, 0, sizeof(*c));
    c->dtype = 1;
    c->dx = 0.01;
    c->gravity[1] = -9.8;
    c->apic_rpic_ratio = 1;
    c->cfl = 0.6;
    c->lsolver = 3;
    c->Ainv = 1;
    c->smoother = 5;
    c->coarseSolver = 2;
    c->levelCnt = 3;
    c->times = 1;
    c->levelscale = 0;
    c->omega = 1;
    c->topomega = 0.1;
    c->cneps = 1e-7;
    c->useCN = 1;
    c->project = 1;
    c->systemBCProject = 1;
    c->linesearch = 1;
    c->max_iterations = 10000;
    c->snow[0] = 10, c->snow[1] = 2e-2, c->snow[2] = 7.5e-3, c->snow[3] = 0.6, c->snow[4] = 20;
}

int hot_create(const hot_config* cfg, hot_ctx** out)
{
    if (!cfg || !out) return HOT_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "libhotmi355x: no HIP device visible — this library has no CPU fallback\n");
        return HOT_ERR_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev || (cfg->dtype != 0 && cfg->dtype != 1) || !(cfg->dx > 0)) return HOT_ERR_INVALID;
    if (cfg->shard_owner < 0 || cfg->shard_owner > 2 || cfg->shard_gs < 0 || cfg->shard_gs > 2) {
        fprintf(stderr, "libhotmi355x: hot_create: hot_config.shard_owner must be 0 (by the sweep), 1 (first touch) or 2 (page range), shard_gs 0 (colour-synchronous), 1 (rank-local) or 2 (rank-local, l1-scaled)\n");
        return HOT_ERR_INVALID;
    }
    hot_ctx* c = new hot_ctx;
    try {
        c->impl = cfg->dtype == 0 ? hot::make_ctx_f32(*cfg) : hot::make_ctx_f64(*cfg);
    }
    catch (const hot::Error& e) {
        fprintf(stderr, "libhotmi355x: hot_create failed: %s\n", e.msg.c_str());
        delete c;
        return e.code;
    }
    catch (const std::exception& e) { // std::bad_alloc and friends must not cross the extern "C" boundary either
        fprintf(stderr, "libhotmi355x: hot_create failed: %s\n", e.what());
        delete c;
        return HOT_ERR_INVALID;
    }
    catch (...) {
        delete c;
        return HOT_ERR_INVALID;
    }
    *out = c;
    return HOT_OK;
}
void hot_destroy(hot_ctx* ctx)
{
    if (!ctx) return;
    delete ctx->impl;
    delete ctx;
}
const char* hot_last_error(hot_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int hot_sync(hot_ctx* ctx)
{
    HOT_API_BEGIN
    HOT_HIP(hipStreamSynchronize(ctx->impl->stream));
    ctx->impl->prof.collect();
    HOT_API_END
}
int hot_set_particles(hot_ctx* ctx, int64_t Np, const void* X, const void* V, const void* mass, const void* C, const void* F, const void* vol, const void* mu, const void* lambda, const void* Jp)
{
    HOT_API_BEGIN
    ctx->impl->set_particles(Np, X, V, mass, C, F, vol, mu, lambda, Jp);
    HOT_API_END
}
int hot_get_particles(hot_ctx* ctx, void* X, void* V, void* C, void* F, void* mu, void* lambda, void* Jp)
{
    HOT_API_BEGIN
    ctx->impl->get_particles(X, V, C, F, mu, lambda, Jp);
    HOT_API_END
}
int hot_sort(hot_ctx* ctx)
{
    HOT_API_BEGIN
    ctx->impl->sort();
    HOT_API_END
}
int hot_get_counts(hot_ctx* ctx, int64_t* Np, int32_t* Ng, int32_t* Nb, int32_t* Nn)
{
    HOT_API_BEGIN
    ctx->impl->get_counts(Np, Ng, Nb, Nn);
    HOT_API_END
}
int hot_get_indexing(hot_ctx* ctx, int32_t* order, uint64_t* base_offset, int32_t* group, uint64_t* block_offset, uint64_t* blocks)
{
    HOT_API_BEGIN
    ctx->impl->get_indexing(order, base_offset, group, block_offset, blocks);
    HOT_API_END
}
int hot_p2g(hot_ctx* ctx)
{
    HOT_API_BEGIN
    ctx->impl->p2g();
    HOT_API_END
}
int hot_get_grid(hot_ctx* ctx, int32_t* id2coord, void* mass, void* v)
{
    HOT_API_BEGIN
    ctx->impl->get_grid(id2coord, mass, v);
    HOT_API_END
}
int hot_set_bc(hot_ctx* ctx, int32_t Nc, const int32_t* node_id, const void* P, const void* R, const void* Rinv, const uint8_t* slip, const void* dvc)
{
    HOT_API_BEGIN
    ctx->impl->set_bc(Nc, node_id, P, R, Rinv, slip, dvc);
    HOT_API_END
}
int hot_set_sticky_halfspaces(hot_ctx* ctx, int32_t n, const double* origin, const double* normal)
{
    HOT_API_BEGIN
    ctx->impl->set_halfspaces(n, origin, normal);
    HOT_API_END
}
int hot_set_collision_objects(hot_ctx* ctx, int32_t n, const hot_collision_object* objects)
{
    HOT_API_BEGIN
    ctx->impl->set_collision_objects(n, objects);
    HOT_API_END
}
int hot_begin_step(hot_ctx* ctx, double dt)
{
    HOT_API_BEGIN
    ctx->impl->begin_step(dt);
    HOT_API_END
}
int hot_get_dv(hot_ctx* ctx, void* dv)
{
    HOT_API_BEGIN
    ctx->impl->get_dv(dv);
    HOT_API_END
}
int hot_set_dv(hot_ctx* ctx, const void* dv)
{
    HOT_API_BEGIN
    ctx->impl->set_dv(dv);
    HOT_API_END
}
int hot_update_state(hot_ctx* ctx, const void* dv, double* energy)
{
    HOT_API_BEGIN
    ctx->impl->update_state(dv, energy);
    HOT_API_END
}
int hot_get_particle_state(hot_ctx* ctx, void* F, void* stress, void* gradV)
{
    HOT_API_BEGIN
    ctx->impl->get_particle_state(F, stress, gradV);
    HOT_API_END
}
int hot_residual(hot_ctx* ctx, void* r)
{
    HOT_API_BEGIN
    ctx->impl->residual(r);
    HOT_API_END
}
int hot_project(hot_ctx* ctx, void* v)
{
    HOT_API_BEGIN
    ctx->impl->project(v);
    HOT_API_END
}
int hot_cn_tolerance(hot_ctx* ctx, void* tol)
{
    HOT_API_BEGIN
    ctx->impl->cn_tolerance(tol);
    HOT_API_END
}
int hot_build_hessian(hot_ctx* ctx)
{
    HOT_API_BEGIN
    ctx->impl->build_hessian();
    HOT_API_END
}
int hot_matfree_multiply(hot_ctx* ctx, const void* x, void* y)
{
    HOT_API_BEGIN
    ctx->impl->matfree_multiply(x, y);
    HOT_API_END
}
int hot_build_mg(hot_ctx* ctx)
{
    HOT_API_BEGIN
    ctx->impl->build_mg();
    HOT_API_END
}
int hot_get_level(hot_ctx* ctx, int32_t level, int32_t* nrows, int32_t* colsize, int32_t* id2coord)
{
    HOT_API_BEGIN
    ctx->impl->get_level(level, nrows, colsize, id2coord);
    HOT_API_END
}
int hot_get_matrix(hot_ctx* ctx, int32_t level, int32_t* entryCol, void* entryVal)
{
    HOT_API_BEGIN
    ctx->impl->get_matrix(level, entryCol, entryVal);
    HOT_API_END
}
int hot_get_level_nnzb(hot_ctx* ctx, int32_t level, int64_t* nnzb)
{
    HOT_API_BEGIN
    *nnzb = ctx->impl->get_level_nnzb(level);
    HOT_API_END
}
int hot_get_level_inblock_nnzb(hot_ctx* ctx, int32_t level, int64_t* nnzb)
{
    HOT_API_BEGIN
    *nnzb = ctx->impl->get_level_inblock_nnzb(level);
    HOT_API_END
}
// ---- what this box streams: a device-to-device copy KERNEL on the context's stream (bench.py roofline.peak_measured)
typedef float hot_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const hot_f4* __restrict__ src, hot_f4* __restrict__ dst, size_t n16)
{
    // ONE 16-byte piece per thread, non-temporal both ways, consecutive lanes on consecutive pieces: the shape that streams fastest here
    // (tools/micro/copy_bw.hip, 1 GiB: 6.5 TB/s; 6.2 without the non-temporal hint; grid-stride loops of 4096 workgroups 4.3 - 4.7; hipMemcpyAsync 5.0)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
int hot_copy_bandwidth(hot_ctx* ctx, int64_t bytes, int32_t reps, double* gbytes_per_s)
{
    HOT_API_BEGIN
    HOT_CHECK(bytes >= (1 << 20) && reps > 0 && gbytes_per_s, HOT_ERR_INVALID, "hot_copy_bandwidth: at least 1 MiB, one repetition");
    hipStream_t st = ctx->impl->stream;
    hot::DBuf<hot_f4> a, b;
    const size_t n16 = (size_t)bytes / 16;
    a.reserve(n16), b.reserve(n16);
    HOT_HIP(hipMemsetAsync(a.p, 0x11, n16 * 16, st));
    hipEvent_t e0, e1;
    HOT_HIP(hipEventCreate(&e0));
    HOT_HIP(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n16 + 255) / 256);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, a.p, b.p, n16);
    HOT_HIP(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, a.p, b.p, n16);
    HOT_HIP(hipEventRecord(e1, st));
    HOT_HIP(hipEventSynchronize(e1));
    float ms = 0;
    HOT_HIP(hipEventElapsedTime(&ms, e0, e1));
    HOT_HIP(hipEventDestroy(e0));
    HOT_HIP(hipEventDestroy(e1));
    *gbytes_per_s = 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9; // bytes read + bytes written
    HOT_API_END
}
int hot_get_prolongation(hot_ctx* ctx, int32_t level, int32_t* entryCol, void* weight)
{
    HOT_API_BEGIN
    ctx->impl->get_prolongation(level, entryCol, weight);
    HOT_API_END
}
int hot_spmv(hot_ctx* ctx, int32_t level, const void* x, void* y)
{
    HOT_API_BEGIN
    ctx->impl->spmv(level, x, y);
    HOT_API_END
}
int hot_restrict(hot_ctx* ctx, int32_t level, const void* fine, void* coarse)
{
    HOT_API_BEGIN
    ctx->impl->restrict_(level, fine, coarse);
    HOT_API_END
}
int hot_prolong(hot_ctx* ctx, int32_t level, const void* coarse, void* fine)
{
    HOT_API_BEGIN
    ctx->impl->prolong(level, coarse, fine);
    HOT_API_END
}
int hot_smooth(hot_ctx* ctx, int32_t level, int32_t kind, int32_t iterations, double tolerance, void* u, void* r, const void* r0)
{
    HOT_API_BEGIN
    ctx->impl->smooth(level, kind, iterations, tolerance, u, r, r0);
    HOT_API_END
}
int hot_vcycle(hot_ctx* ctx, const void* in, void* out)
{
    HOT_API_BEGIN
    ctx->impl->vcycle(in, out);
    HOT_API_END
}
int hot_solve(hot_ctx* ctx, hot_stats* stats)
{
    HOT_API_BEGIN
    ctx->impl->solve(stats);
    HOT_API_END
}
int hot_g2p(hot_ctx* ctx, double dt, int32_t* flags)
{
    HOT_API_BEGIN
    ctx->impl->g2p(dt, flags);
    HOT_API_END
}
int hot_line_search(hot_ctx* ctx, void* ddv, void* residual, double alpha, double* alpha_out)
{
    HOT_API_BEGIN
    ctx->impl->line_search_api(ddv, residual, alpha, alpha_out);
    HOT_API_END
}
int hot_should_exit(hot_ctx* ctx, const void* residual, int32_t* exit_now, double* scaled_residual)
{
    HOT_API_BEGIN
    ctx->impl->should_exit_api(residual, exit_now, scaled_residual);
    HOT_API_END
}
int hot_recover_solution(hot_ctx* ctx, void* v)
{
    HOT_API_BEGIN
    ctx->impl->transform_api(v, true);
    HOT_API_END
}
int hot_transform_residual(hot_ctx* ctx, void* v)
{
    HOT_API_BEGIN
    ctx->impl->transform_api(v, false);
    HOT_API_END
}
int hot_compute_step(hot_ctx* ctx, const void* residual, void* step)
{
    HOT_API_BEGIN
    ctx->impl->compute_step_api(residual, step);
    HOT_API_END
}
int hot_write_partio(hot_ctx* ctx, const char* path)
{
    HOT_API_BEGIN
    ctx->impl->write_partio(path);
    HOT_API_END
}
int hot_write_restart(hot_ctx* ctx, const char* path)
{
    HOT_API_BEGIN
    ctx->impl->write_restart(path);
    HOT_API_END
}
int hot_read_restart(hot_ctx* ctx, const char* path)
{
    HOT_API_BEGIN
    ctx->impl->read_restart(path);
    HOT_API_END
}
int hot_set_particle_ids(hot_ctx* ctx, const int32_t* ids)
{
    HOT_API_BEGIN
    ctx->impl->set_particle_ids(ids);
    HOT_API_END
}
int hot_get_particle_ids(hot_ctx* ctx, int32_t* ids)
{
    HOT_API_BEGIN
    ctx->impl->get_particle_ids(ids);
    HOT_API_END
}
int hot_get_stream(hot_ctx* ctx, void** hip_stream)
{
    HOT_API_BEGIN
    if (hip_stream) *hip_stream = (void*)ctx->impl->stream;
    HOT_API_END
}
int hot_set_comm(hot_ctx* ctx, const hot_comm* comm)
{
    HOT_API_BEGIN
    ctx->impl->set_comm(comm);
    HOT_API_END
}
int hot_constitutive_eval(hot_ctx* ctx, int32_t n, const void* F, const void* mu, const void* lambda, int32_t project, void* psi, void* P, void* dPdF)
{
    HOT_API_BEGIN
    ctx->impl->constitutive_eval(n, F, mu, lambda, project, psi, P, dPdF);
    HOT_API_END
}
int hot_plasticity_eval(hot_ctx* ctx, int32_t kind, int32_t n, void* F, void* mu, void* lambda, void* Jp)
{
    HOT_API_BEGIN
    ctx->impl->plasticity_eval(kind, n, F, mu, lambda, Jp);
    HOT_API_END
}
int hot_advance(hot_ctx* ctx, double dt, hot_stats* stats)
{
    HOT_API_BEGIN
    ctx->impl->advance(dt, stats);
    HOT_API_END
}
int hot_calculate_dt(hot_ctx* ctx, double max_dt, double* dt, double* max_speed, double* min_corner, double* max_corner)
{
    HOT_API_BEGIN
    ctx->impl->calculate_dt(max_dt, dt, max_speed, min_corner, max_corner);
    HOT_API_END
}
int hot_advance_frame(hot_ctx* ctx, double frame_dt, double min_dt, double max_dt, int32_t* substeps, int32_t* iterations_total, hot_stats* stats)
{
    HOT_API_BEGIN
    ctx->impl->advance_frame(frame_dt, min_dt, max_dt, substeps, iterations_total, stats);
    HOT_API_END
}
int hot_profile_reset(hot_ctx* ctx)
{
    HOT_API_BEGIN
    HOT_HIP(hipStreamSynchronize(ctx->impl->stream));
    ctx->impl->prof.collect();
    ctx->impl->prof.recs.clear();
    HOT_API_END
}
int hot_profile_count(hot_ctx* ctx, int32_t* n)
{
    HOT_API_BEGIN
    HOT_HIP(hipStreamSynchronize(ctx->impl->stream));
    ctx->impl->prof.collect();
    *n = (int32_t)ctx->impl->prof.recs.size();
    HOT_API_END
}
int hot_profile_get(hot_ctx* ctx, int32_t i, char* name, int64_t* calls, double* total_ms)
{
    HOT_API_BEGIN
    auto& recs = ctx->impl->prof.recs;
    HOT_CHECK(i >= 0 && i < (int)recs.size(), HOT_ERR_INVALID, "profile index out of range");
    auto it = recs.begin();
    std::advance(it, i);
    if (name) {
        std::strncpy(name, it->first.c_str(), 63);
        name[63] = 0;
    }
    if (calls) *calls = it->second.calls;
    if (total_ms) *total_ms = it->second.ms;
    HOT_API_END
}
const char* hot_version(void) { return "libhotmi355x 0.2 (gfx950)"; }
int hot_abi_version(void) { return HOT_ABI_VERSION; }

} // extern "C"
