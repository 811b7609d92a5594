// ORACLE (test infrastructure only — never linked into the product library).
//
// C entry points of liboracle.so: the CPU restatement behind exactly the signatures of
// include/hot_mi355x.h with the prefix `hoto_` instead of `hot_`, so the parity tests drive the oracle and
// the HIP library through the same Python wrapper.  All pointers are host pointers here.
#include "sim_solve.hpp"
#include <cstdio>
#include <map>
#include <cstring>
#include <cstdlib>

using namespace hot_oracle;

struct hoto_ctx {
    int dtype;
    Sim<float>* f = nullptr;
    Sim<double>* d = nullptr;
    std::string err;
};

#define DISPATCH(ctx, ...)                \
    do {                                  \
        if ((ctx)->dtype == 0) {          \
            auto& S = *(ctx)->f;          \
            using T = float;              \
            (void)sizeof(T);              \
            __VA_ARGS__;                  \
        }                                 \
        else {                            \
            auto& S = *(ctx)->d;          \
            using T = double;             \
            (void)sizeof(T);              \
            __VA_ARGS__;                  \
        }                                 \
    } while (0)

template <class T>
static void copy_tv(const std::vector<V3<T>>& v, void* out)
{
    if (out) std::memcpy(out, v.data(), v.size() * sizeof(V3<T>));
}
template <class T>
static std::vector<V3<T>> load_tv(const void* in, size_t n)
{
    std::vector<V3<T>> v(n);
    std::memcpy(v.data(), in, n * sizeof(V3<T>));
    return v;
}

template <class U>
static void o_put(FILE* f, U v) { fwrite(&v, sizeof(U), 1, f); }
template <class T, class Get>
static void o_array(FILE* f, const char* name, int64_t n, int comps, const Get& get)
{
    o_put<uint64_t>(f, strlen(name));
    fwrite(name, 1, strlen(name), f);
    o_put<int32_t>(f, 7);
    o_put<uint64_t>(f, 1), o_put<uint64_t>(f, 8), o_put<int32_t>(f, 0), o_put<int32_t>(f, (int32_t)n);
    o_put<uint64_t>(f, (uint64_t)n), o_put<uint64_t>(f, (uint64_t)comps * sizeof(T));
    for (int64_t p = 0; p < n; ++p)
        for (int k = 0; k < comps; ++k) o_put<T>(f, get(p, k));
}

extern "C" {

void hoto_default_config(hot_config* c)
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

void hoto_set_wide(int on) { hot_oracle::wide_flag() = on != 0; } // tests/oracle_lib.py wide_sums(): restore the process-wide flag
void hoto_set_psi_invariants(int on) { hot_oracle::psi_invariants_flag() = on != 0; } // tests/oracle_lib.py psi_invariants()

int hoto_create(const hot_config* cfg, hoto_ctx** out)
{
    hoto_ctx* c = new hoto_ctx;
    c->dtype = cfg->dtype;
    hot_oracle::fair_flag() = getenv("HOT_ORACLE_FAIR") != nullptr; // CPU-baseline variant (sim_core.hpp), timing only
    hot_oracle::wide_flag() = getenv("HOT_ORACLE_WIDE") != nullptr; // fp32: node sums and dot products accumulated in double (sim_core.hpp)
    hot_oracle::psi_invariants_flag() = getenv("HOT_ORACLE_PSI_INVARIANTS") != nullptr; // every energy in the product's trial form (corotated.hpp corotated_psi_product_form)
    DISPATCH(c, {
        auto* s = new Sim<T>();
        s->cfg = *cfg;
        s->dx = (T)cfg->dx;
        for (int k = 0; k < 3; ++k) s->gravity(k) = (T)cfg->gravity[k];
        std::memset(&s->stats, 0, sizeof(s->stats));
        if (c->dtype == 0)
            c->f = (Sim<float>*)(void*)s;
        else
            c->d = (Sim<double>*)(void*)s;
    });
    *out = c;
    return 0;
}
void hoto_destroy(hoto_ctx* c)
{
    if (!c) return;
    delete c->f;
    delete c->d;
    delete c;
}
const char* hoto_last_error(hoto_ctx* c)
{
    DISPATCH(c, c->err = S.err);
    return c->err.c_str();
}
int hoto_sync(hoto_ctx*) { return 0; }

int hoto_set_particles(hoto_ctx* c, int64_t Np, const void* X, const void* V, const void* mass, const void* C, const void* F, const void* vol, const void* mu, const void* lambda, const void* Jp)
{
    DISPATCH(c, S.set_particles(Np, (const T*)X, (const T*)V, (const T*)mass, (const T*)C, (const T*)F, (const T*)vol, (const T*)mu, (const T*)lambda, (const T*)Jp));
    return 0;
}
int hoto_get_particles(hoto_ctx* c, void* X, void* V, void* C, void* F, void* mu, void* lambda, void* Jp)
{
    DISPATCH(c, {
        size_t n = S.Np;
        if (X) std::memcpy(X, S.X.data(), n * 3 * sizeof(T));
        if (V) std::memcpy(V, S.Vel.data(), n * 3 * sizeof(T));
        if (C) std::memcpy(C, S.C.data(), n * 9 * sizeof(T));
        if (F) std::memcpy(F, S.F.data(), n * 9 * sizeof(T));
        if (mu) std::memcpy(mu, S.mu.data(), n * sizeof(T));
        if (lambda) std::memcpy(lambda, S.lambda.data(), n * sizeof(T));
        if (Jp) std::memcpy(Jp, S.Jp.data(), n * sizeof(T));
    });
    return 0;
}
int hoto_sort(hoto_ctx* c)
{
    int rc = 0;
    DISPATCH(c, rc = S.sort_particles());
    return rc;
}
int hoto_get_counts(hoto_ctx* c, int64_t* Np, int32_t* Ng, int32_t* Nb, int32_t* Nn)
{
    DISPATCH(c, {
        if (Np) *Np = S.Np;
        if (Ng) *Ng = (int32_t)S.particle_group.size();
        if (Nb) *Nb = (int32_t)S.blocks.size();
        if (Nn) *Nn = S.num_nodes;
    });
    return 0;
}
int hoto_get_indexing(hoto_ctx* c, int32_t* order, uint64_t* base_offset, int32_t* group, uint64_t* block_offset, uint64_t* blocks)
{
    DISPATCH(c, {
        if (order) std::memcpy(order, S.particle_order.data(), S.Np * sizeof(int32_t));
        if (base_offset) std::memcpy(base_offset, S.particle_base_offset.data(), S.Np * sizeof(uint64_t));
        if (group)
            for (size_t g = 0; g < S.particle_group.size(); ++g) group[2 * g] = S.particle_group[g].first, group[2 * g + 1] = S.particle_group[g].second;
        if (block_offset) std::memcpy(block_offset, S.block_offset.data(), S.block_offset.size() * sizeof(uint64_t));
        if (blocks) std::memcpy(blocks, S.blocks.data(), S.blocks.size() * sizeof(uint64_t));
    });
    return 0;
}
int hoto_p2g(hoto_ctx* c)
{
    DISPATCH(c, S.particles_to_grid());
    return 0;
}
int hoto_get_grid(hoto_ctx* c, int32_t* id2coord, void* mass, void* v)
{
    DISPATCH(c, {
        if (id2coord) std::memcpy(id2coord, S.id2coord.data(), (size_t)S.num_nodes * 3 * sizeof(int32_t));
        if (mass) std::memcpy(mass, S.mass_matrix.data(), (size_t)S.num_nodes * sizeof(T));
        if (v)
            for (int n = 0; n < S.num_nodes; ++n)
                for (int k = 0; k < 3; ++k) ((T*)v)[3 * n + k] = S.nodes[S.dof_slot[n]].v(k);
    });
    return 0;
}
int hoto_set_bc(hoto_ctx* c, int32_t Nc, const int32_t* node_id, const void* P, const void* R, const void* Rinv, const uint8_t* slip, const void* dvc)
{
    int rc = 0;
    DISPATCH(c, {
        S.hs_origin.clear(), S.hs_normal.clear();
        rc = S.set_bc(Nc, node_id, (const T*)P, (const T*)R, (const T*)Rinv, slip, (const T*)dvc);
    });
    return rc;
}
int hoto_set_sticky_halfspaces(hoto_ctx* c, int32_t n, const double* origin, const double* normal)
{
    DISPATCH(c, {
        S.hs_origin.assign(origin, origin + 3 * n);
        S.hs_normal.assign(normal, normal + 3 * n);
    });
    return 0;
}
int hoto_set_collision_objects(hoto_ctx* c, int32_t n, const hot_collision_object* objects)
{
    DISPATCH(c, {
        S.cobjs.assign(objects, objects + n);
        S.hs_origin.clear(), S.hs_normal.clear();
        if (n == 0) S.collision_nodes.clear();
    });
    return 0;
}
int hoto_begin_step(hoto_ctx* c, double dt)
{
    DISPATCH(c, S.begin_step((T)dt));
    return 0;
}
int hoto_get_dv(hoto_ctx* c, void* dv)
{
    DISPATCH(c, copy_tv(S.dv, dv));
    return 0;
}
int hoto_set_dv(hoto_ctx* c, const void* dv)
{
    DISPATCH(c, S.dv = load_tv<T>(dv, S.num_nodes));
    return 0;
}
int hoto_update_state(hoto_ctx* c, const void* dv, double* energy)
{
    DISPATCH(c, {
        if (dv) {
            auto v = load_tv<T>(dv, S.num_nodes);
            S.dv = v;
        }
        S.update_position_based_state();
        S.Ek = S.total_energy();
        if (energy) *energy = S.Ek;
    });
    return 0;
}
int hoto_get_particle_state(hoto_ctx* c, void* F, void* stress, void* gradV)
{
    DISPATCH(c, {
        size_t n = S.Np;
        if (F) std::memcpy(F, S.F.data(), n * 9 * sizeof(T));
        if (stress) std::memcpy(stress, S.scratch_stress.data(), n * 9 * sizeof(T));
        if (gradV) std::memcpy(gradV, S.scratch_gradV.data(), n * 9 * sizeof(T));
    });
    return 0;
}
int hoto_residual(hoto_ctx* c, void* r)
{
    DISPATCH(c, {
        std::vector<V3<T>> res;
        S.compute_residual(res);
        copy_tv(res, r);
    });
    return 0;
}
int hoto_project(hoto_ctx* c, void* v)
{
    DISPATCH(c, {
        auto x = load_tv<T>(v, S.num_nodes);
        S.project(x);
        copy_tv(x, v);
    });
    return 0;
}
int hoto_cn_tolerance(hoto_ctx* c, void* tol)
{
    DISPATCH(c, {
        S.evaluate_cn_tolerance();
        if (tol) std::memcpy(tol, S.nodeCNTol.data(), (size_t)S.num_nodes * sizeof(T));
    });
    return 0;
}
int hoto_build_hessian(hoto_ctx* c)
{
    DISPATCH(c, {
        S.build_matrix();
        Sim<T>::build_diagonal(S.sysmats[0], S.cfg.Ainv);
    });
    return 0;
}
int hoto_matfree_multiply(hoto_ctx* c, const void* x, void* y)
{
    DISPATCH(c, {
        auto xx = load_tv<T>(x, S.num_nodes);
        std::vector<V3<T>> b;
        S.matfree_multiply(xx, b);
        copy_tv(b, y);
    });
    return 0;
}
int hoto_build_mg(hoto_ctx* c)
{
    DISPATCH(c, S.build_mg());
    return 0;
}
int hoto_get_level(hoto_ctx* c, int32_t level, int32_t* nrows, int32_t* colsize, int32_t* id2coord)
{
    int rc = 0;
    DISPATCH(c, {
        if (level < 0 || level >= (int)S.sysmats.size())
            rc = HOT_ERR_INVALID;
        else {
            if (nrows) *nrows = S.sysmats[level].nrows;
            if (colsize) *colsize = S.sysmats[level].colsize;
            if (id2coord) std::memcpy(id2coord, S.level_coords[level].data(), S.level_coords[level].size() * 3 * sizeof(int32_t));
        }
    });
    return rc;
}
int hoto_get_matrix(hoto_ctx* c, int32_t level, int32_t* entryCol, void* entryVal)
{
    DISPATCH(c, {
        auto& m = S.sysmats[level];
        if (entryCol) std::memcpy(entryCol, m.entryCol.data(), m.entryCol.size() * sizeof(int32_t));
        if (entryVal) std::memcpy(entryVal, m.entryVal.data(), m.entryVal.size() * 9 * sizeof(T));
    });
    return 0;
}
int hoto_get_level_nnzb(hoto_ctx* c, int32_t level, int64_t* nnzb)
{
    DISPATCH(c, {
        auto& m = S.sysmats[level];
        int64_t cnt = 0;
        for (auto& v : m.entryVal) {
            bool nz = false;
            for (int k = 0; k < 9; ++k) nz = nz || v.a[k] != 0;
            cnt += nz;
        }
        *nnzb = cnt;
    });
    return 0;
}
int hoto_get_prolongation(hoto_ctx* c, int32_t level, int32_t* entryCol, void* weight)
{
    DISPATCH(c, {
        auto& m = S.promats[level];
        if (entryCol) std::memcpy(entryCol, m.entryCol.data(), m.entryCol.size() * sizeof(int32_t));
        if (weight)
            for (size_t k = 0; k < m.entryVal.size(); ++k) ((T*)weight)[k] = m.entryVal[k](0, 0);
    });
    return 0;
}
int hoto_spmv(hoto_ctx* c, int32_t level, const void* x, void* y)
{
    DISPATCH(c, {
        auto xx = load_tv<T>(x, S.sysmats[level].nrows);
        std::vector<V3<T>> b;
        Sim<T>::multiply(S.sysmats[level], xx, b);
        copy_tv(b, y);
    });
    return 0;
}
int hoto_restrict(hoto_ctx* c, int32_t level, const void* fine, void* coarse)
{
    DISPATCH(c, {
        auto xx = load_tv<T>(fine, S.sysmats[level].nrows);
        std::vector<V3<T>> b;
        Sim<T>::multiply(S.resmats[level], xx, b);
        copy_tv(b, coarse);
    });
    return 0;
}
int hoto_prolong(hoto_ctx* c, int32_t level, const void* coarse, void* fine)
{
    DISPATCH(c, {
        auto xx = load_tv<T>(coarse, S.sysmats[level + 1].nrows);
        std::vector<V3<T>> b;
        Sim<T>::multiply(S.promats[level], xx, b);
        copy_tv(b, fine);
    });
    return 0;
}
int hoto_smooth(hoto_ctx* c, int32_t level, int32_t kind, int32_t iterations, double tolerance, void* u, void* r, const void* r0)
{
    DISPATCH(c, {
        int n = S.sysmats[level].nrows;
        auto uu = load_tv<T>(u, n);
        auto rr = load_tv<T>(r, n);
        S.mg_initialResiduals[level] = r0 ? load_tv<T>(r0, n) : rr;
        std::vector<V3<T>> du(n, V3<T>::zero()), dAu(n, V3<T>::zero());
        S.smooth(kind, level, uu, rr, du, dAu, iterations, (T)tolerance);
        copy_tv(uu, u);
        copy_tv(rr, r);
    });
    return 0;
}
int hoto_vcycle(hoto_ctx* c, const void* in, void* out)
{
    DISPATCH(c, {
        auto xx = load_tv<T>(in, S.num_nodes);
        std::vector<V3<T>> b;
        S.vcycle(xx, b);
        copy_tv(b, out);
    });
    return 0;
}
int hoto_solve(hoto_ctx* c, hot_stats* stats)
{
    int rc = 0;
    DISPATCH(c, {
        rc = S.solve();
        if (stats) *stats = S.stats;
    });
    return rc;
}
int hoto_g2p(hoto_ctx* c, double dt, int32_t* flags)
{
    DISPATCH(c, {
        int f = S.grid_to_particles(dt);
        if (flags) *flags = f;
    });
    return 0;
}
// the remaining members of the solver-facing objective concept (ImplicitSolver.h): lineSearch :312-333, shouldExitByCN :174-211,
// recoverSolution / transformResidual :106-125, computeStep :355-432
int hoto_line_search(hoto_ctx* c, void* ddv, void* residual, double alpha, double* alpha_out)
{
    DISPATCH(c, {
        auto d = load_tv<T>(ddv, S.num_nodes);
        std::vector<V3<T>> r(S.num_nodes);
        T a = S.line_search(d, r, (T)alpha);
        copy_tv(d, ddv);
        copy_tv(r, residual);
        if (alpha_out) *alpha_out = (double)a;
    });
    return 0;
}
int hoto_should_exit(hoto_ctx* c, const void* residual, int32_t* exit_now, double* scaled_residual)
{
    DISPATCH(c, {
        auto r = load_tv<T>(residual, S.num_nodes);
        if (S.cfg.useCN && S.nodeCNTol.size() != (size_t)S.num_nodes) S.evaluate_cn_tolerance();
        bool e = S.should_exit(r);
        if (exit_now) *exit_now = e ? 1 : 0;
        if (scaled_residual) *scaled_residual = S.stats.final_scaled_residual;
    });
    return 0;
}
int hoto_recover_solution(hoto_ctx* c, void* v)
{
    DISPATCH(c, {
        auto x = load_tv<T>(v, S.num_nodes);
        S.recover_solution(x);
        copy_tv(x, v);
    });
    return 0;
}
int hoto_transform_residual(hoto_ctx* c, void* v)
{
    DISPATCH(c, {
        auto x = load_tv<T>(v, S.num_nodes);
        S.transform_residual(x);
        copy_tv(x, v);
    });
    return 0;
}
int hoto_compute_step(hoto_ctx* c, const void* residual, void* step)
{
    DISPATCH(c, {
        auto r = load_tv<T>(residual, S.num_nodes);
        std::vector<V3<T>> st(S.num_nodes, V3<T>::zero());
        if (S.cfg.useCN && S.nodeCNTol.size() != (size_t)S.num_nodes) S.evaluate_cn_tolerance();
        S.compute_step(r, st);
        copy_tv(st, step);
    });
    return 0;
}
// frame output, restated independently of the product's io.hip from the same reference sources: writePartio's .bgeo of the positions
// (Lib/Ziran/Math/Geometry/PartioIO.h:142-180; Bgeo v5 container, big-endian) and the particle DataManager container
// (Lib/Ziran/CS/DataStructure/DataManager.h:263-294, DataArray.h:100-105, Lib/Ziran/CS/Util/BinaryIO.h:82-88)
static void o_be32(FILE* f, uint32_t v)
{
    unsigned char b[4] = { (unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v };
    fwrite(b, 1, 4, f);
}
int hoto_write_partio(hoto_ctx* c, const char* path)
{
    FILE* f = fopen(path, "wb");
    if (!f) return HOT_ERR_INVALID;
    DISPATCH(c, {
        o_be32(f, 0x4267656fu); // "Bgeo"
        fputc('V', f);
        o_be32(f, 5), o_be32(f, (uint32_t)S.Np);
        for (int k = 0; k < 7; ++k) o_be32(f, 0);
        for (int64_t p = 0; p < S.Np; ++p) {
            for (int d = 0; d < 4; ++d) {
                float v = d < 3 ? (float)S.X[p](d) : 1.0f;
                uint32_t u;
                std::memcpy(&u, &v, 4);
                o_be32(f, u);
            }
        }
        fputc(0x00, f), fputc(0xff, f);
    });
    fclose(f);
    return 0;
}
int hoto_write_restart(hoto_ctx* c, const char* path)
{
    FILE* f = fopen(path, "wb");
    if (!f) return HOT_ERR_INVALID;
    DISPATCH(c, {
        const int64_t n = S.Np;
        o_put<int32_t>(f, (int32_t)n), o_put<uint64_t>(f, 9);
        o_array<T>(f, "m", n, 1, [&](int64_t p, int) { return S.mass[p]; });
        o_array<T>(f, "P", n, 3, [&](int64_t p, int k) { return S.X[p](k); });
        o_array<T>(f, "V", n, 3, [&](int64_t p, int k) { return S.Vel[p](k); });
        o_array<T>(f, "C", n, 9, [&](int64_t p, int k) { return S.C[p].a[k]; });
        o_array<T>(f, "F", n, 9, [&](int64_t p, int k) { return S.F[p].a[k]; });
        o_array<T>(f, "element measure", n, 1, [&](int64_t p, int) { return S.vol[p]; });
        o_array<T>(f, "mu", n, 1, [&](int64_t p, int) { return S.mu[p]; });
        o_array<T>(f, "lambda", n, 1, [&](int64_t p, int) { return S.lambda[p]; });
        o_array<T>(f, "Jp", n, 1, [&](int64_t p, int) { return S.Jp[p]; });
    });
    fclose(f);
    return 0;
}
int hoto_read_restart(hoto_ctx* c, const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) return HOT_ERR_INVALID;
    int rc = 0;
    DISPATCH(c, {
        int32_t n = 0;
        uint64_t narr = 0;
        if (fread(&n, 4, 1, f) != 1 || fread(&narr, 8, 1, f) != 1 || n <= 0 || narr > 64) rc = HOT_ERR_INVALID;
        std::map<std::string, std::vector<T>> col;
        for (uint64_t a = 0; a < narr && rc == 0; ++a) {
            uint64_t len = 0, nr = 0, rb = 0, cnt = 0, bytes = 0;
            int32_t lg = 0;
            if (fread(&len, 8, 1, f) != 1 || len > 255) { rc = HOT_ERR_INVALID; break; }
            std::string name(len, ' ');
            if (fread(&name[0], 1, len, f) != len || fread(&lg, 4, 1, f) != 1 || fread(&nr, 8, 1, f) != 1 || fread(&rb, 8, 1, f) != 1) { rc = HOT_ERR_INVALID; break; }
            fseek(f, (long)(nr * rb), SEEK_CUR);
            if (fread(&cnt, 8, 1, f) != 1 || fread(&bytes, 8, 1, f) != 1 || (int64_t)cnt != n || bytes % sizeof(T) || bytes > 9 * sizeof(T)) { rc = HOT_ERR_INVALID; break; }
            auto& v = col[name];
            v.resize(cnt * (bytes / sizeof(T)));
            if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) rc = HOT_ERR_INVALID;
        }
        const std::pair<const char*, int> want[] = { { "m", 1 }, { "P", 3 }, { "V", 3 }, { "C", 9 }, { "F", 9 }, { "element measure", 1 }, { "mu", 1 }, { "lambda", 1 }, { "Jp", 1 } };
        for (const auto& kw : want)
            if (rc == 0 && (!col.count(kw.first) || (int64_t)col[kw.first].size() != (int64_t)n * kw.second)) rc = HOT_ERR_INVALID;
        if (rc == 0 && col.size() == 9)
            S.set_particles(n, col["P"].data(), col["V"].data(), col["m"].data(), col["C"].data(), col["F"].data(), col["element measure"].data(), col["mu"].data(), col["lambda"].data(), col["Jp"].data());
        else
            rc = HOT_ERR_INVALID;
    });
    fclose(f);
    return rc;
}
// global particle ids: the oracle's shards never exchange particles (its sharded mode sums grid-sized arrays), ids are kept for the caller
int hoto_set_particle_ids(hoto_ctx* c, const int32_t* ids)
{
    DISPATCH(c, { S.particle_ids.assign(ids, ids + S.Np); });
    return 0;
}
int hoto_get_particle_ids(hoto_ctx* c, int32_t* ids)
{
    DISPATCH(c, {
        for (int64_t p = 0; p < S.Np; ++p) ids[p] = S.particle_ids.size() == (size_t)S.Np ? S.particle_ids[p] : (int32_t)p;
    });
    return 0;
}
int hoto_get_stream(hoto_ctx*, void** s)
{
    if (s) *s = nullptr; // host code: no stream
    return 0;
}
int hoto_set_comm(hoto_ctx* c, const hot_comm* comm)
{
    DISPATCH(c, {
        if (comm && comm->size > 1)
            S.comm = *comm;
        else
            S.comm = hot_comm{};
    });
    return 0;
}
// CorotatedIsotropic::updateScratch + psi + firstPiola + firstPiolaDerivative for caller-supplied F (typed by the context)
int hoto_constitutive_eval(hoto_ctx* c, int32_t n, const void* F, const void* mu, const void* lambda, int32_t project, void* psi, void* P, void* dPdF)
{
    DISPATCH(c, {
        (void)S;
        for (int k = 0; k < n; ++k) {
            CorotatedScratch<T> s;
            const T m = ((const T*)mu)[k], l = ((const T*)lambda)[k];
            corotated_update_scratch(((const M3<T>*)F)[k], m, l, project != 0, s);
            if (psi) ((T*)psi)[k] = corotated_psi(s, m, l);
            if (P) ((M3<T>*)P)[k] = corotated_first_piola(s, m, l);
            if (dPdF) corotated_first_piola_derivative(s, (T*)dPdF + 81 * (size_t)k);
        }
    });
    return 0;
}
int hoto_plasticity_eval(hoto_ctx* c, int32_t kind, int32_t n, void* F, void* mu, void* lambda, void* Jp)
{
    DISPATCH(c, {
        for (int k = 0; k < n; ++k) {
            if (kind == 1)
                von_mises_project(((M3<T>*)F)[k], ((T*)mu)[k], ((T*)lambda)[k], (T)S.cfg.yield_stress);
            else
                snow_project(((M3<T>*)F)[k], ((T*)mu)[k], ((T*)lambda)[k], ((T*)Jp)[k], (T)S.cfg.snow[0], (T)S.cfg.snow[1], (T)S.cfg.snow[2], (T)S.cfg.snow[3], (T)S.cfg.snow[4]);
        }
    });
    return 0;
}
int hoto_advance(hoto_ctx* c, double dt, hot_stats* stats)
{
    int rc = 0;
    DISPATCH(c, {
        rc = S.advance(dt);
        if (stats) *stats = S.stats;
    });
    return rc;
}
int hoto_calculate_dt(hoto_ctx* c, double max_dt, double* dt, double* max_speed, double* min_corner, double* max_corner)
{
    DISPATCH(c, {
        double d = S.calculate_dt(max_dt, max_speed, min_corner, max_corner);
        if (dt) *dt = d;
    });
    return 0;
}
int hoto_advance_frame(hoto_ctx* c, double frame_dt, double min_dt, double max_dt, int32_t* substeps, int32_t* iterations_total, hot_stats* stats)
{
    int rc = 0;
    DISPATCH(c, {
        int n = 0, its = 0;
        rc = S.advance_frame(frame_dt, min_dt, max_dt, &n, &its);
        if (substeps) *substeps = n;
        if (iterations_total) *iterations_total = its;
        if (stats) *stats = S.stats;
    });
    return rc;
}
int hoto_profile_reset(hoto_ctx*) { return 0; }
int hoto_profile_count(hoto_ctx*, int32_t* n)
{
    *n = 0;
    return 0;
}
int hoto_profile_get(hoto_ctx*, int32_t, char*, int64_t*, double*) { return HOT_ERR_INVALID; }
const char* hoto_version(void) { return "hot-oracle-cpu 0.1"; }
int hoto_abi_version(void) { return HOT_ABI_VERSION; } // the header this checker was compiled against

// ---- small stand-alone probes used by the oracle's own pin tests
void hoto_linear_offset(int dtype, int n, const int32_t* ijk, uint64_t* out)
{
    for (int c = 0; c < n; ++c) out[c] = dtype == 0 ? SpMask<6>::linear_offset(ijk[3 * c], ijk[3 * c + 1], ijk[3 * c + 2]) : SpMask<7>::linear_offset(ijk[3 * c], ijk[3 * c + 1], ijk[3 * c + 2]);
}
void hoto_linear_to_coord(int dtype, int n, const uint64_t* off, int32_t* ijk)
{
    for (int c = 0; c < n; ++c) {
        auto r = dtype == 0 ? SpMask<6>::linear_to_coord(off[c]) : SpMask<7>::linear_to_coord(off[c]);
        ijk[3 * c] = r[0], ijk[3 * c + 1] = r[1], ijk[3 * c + 2] = r[2];
    }
}
void hoto_packed_add(int dtype, int n, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
    for (int c = 0; c < n; ++c) out[c] = dtype == 0 ? SpMask<6>::packed_add(a[c], b[c]) : SpMask<7>::packed_add(a[c], b[c]);
}
// SVD of n 3x3 matrices (column-major): U, sigma, V
void hoto_svd3(int dtype, int n, const void* A, void* U, void* sigma, void* V)
{
    if (dtype == 0)
        for (int c = 0; c < n; ++c) svd3(((const M3<float>*)A)[c], ((M3<float>*)U)[c], ((V3<float>*)sigma)[c], ((M3<float>*)V)[c]);
    else
        for (int c = 0; c < n; ++c) svd3(((const M3<double>*)A)[c], ((M3<double>*)U)[c], ((V3<double>*)sigma)[c], ((M3<double>*)V)[c]);
}
void hoto_make_pd3(int n, double* S)
{
    for (int c = 0; c < n; ++c) make_pd3(((M3<double>*)S)[c]);
}
void hoto_make_pd2(int n, double* abd)
{
    for (int c = 0; c < n; ++c) make_pd2(abd[3 * c], abd[3 * c + 1], abd[3 * c + 2]);
}
// constitutive probe: psi, P (9), dPdF (81) for n deformation gradients
void hoto_corotated(int n, const double* F, double mu, double lambda, int project, double* psi, double* P, double* dPdF)
{
    for (int c = 0; c < n; ++c) {
        CorotatedScratch<double> s;
        corotated_update_scratch(((const M3<double>*)F)[c], mu, lambda, project != 0, s);
        if (psi) psi[c] = corotated_psi(s, mu, lambda);
        if (P) ((M3<double>*)P)[c] = corotated_first_piola(s, mu, lambda);
        if (dPdF) corotated_first_piola_derivative(s, dPdF + 81 * c);
    }
}
void hoto_corotated_differential(int n, const double* F, const double* dF, double mu, double lambda, int project, double* dP)
{
    for (int c = 0; c < n; ++c) {
        CorotatedScratch<double> s;
        corotated_update_scratch(((const M3<double>*)F)[c], mu, lambda, project != 0, s);
        ((M3<double>*)dP)[c] = corotated_first_piola_differential(s, ((const M3<double>*)dF)[c]);
    }
}
void hoto_plasticity(int kind, int n, double* F, double* mu, double* lambda, double* Jp, double yield, const double* snow)
{
    for (int c = 0; c < n; ++c) {
        if (kind == 1)
            von_mises_project(((M3<double>*)F)[c], mu[c], lambda[c], yield);
        else
            snow_project(((M3<double>*)F)[c], mu[c], lambda[c], Jp[c], snow[0], snow[1], snow[2], snow[3], snow[4]);
    }
}
int hoto_num_threads(void) { return omp_get_max_threads(); }

} // extern "C"
