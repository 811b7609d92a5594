// ORACLE (test infrastructure only).  Hessian assembly (padded ELL-125 of 3x3 blocks), Galerkin multigrid
// hierarchy, smoothers and the V-cycle.
#pragma once
#include <stdexcept>
#include "sim_force.hpp"
#include <map>

namespace hot_oracle {

static inline int linear_offset125(int dx, int dy, int dz) { return (dx + 2) * 25 + (dy + 2) * 5 + dz + 2; } // ImplicitSolver.h:465-468

// reference ImplicitSolverObjective::buildMatrix<true> (Projects/multigrid/ImplicitSolver.h:470-603) with
// FBasedMpmForceHelper::runLambdaWithDifferential (Lib/MPM/Force/FBasedMpmForceHelper.h:63-121)
template <class T>
void Sim<T>::build_matrix()
{
    sysmats.assign(1, EllMat<T>());
    EllMat<T>& A = sysmats[0];
    A.colsize = 125;
    A.nrows = num_nodes;
    A.entryCol.assign((size_t)num_nodes * 125, -1);
    A.entryVal.assign((size_t)num_nodes * 125, TM::zero());
    dRhs.assign(num_nodes, TV::zero());
    level_coords.assign(1, id2coord);
    // inertia term (sharded: contributed once, by rank 0, the shards' force terms are summed below)
#pragma omp parallel for schedule(static)
    for (int n = 0; n < num_nodes; ++n) {
        if (sharded() && comm.rank != 0) continue;
        A.entryCol[(size_t)n * 125 + linear_offset125(0, 0, 0)] = n;
        A.entryVal[(size_t)n * 125 + linear_offset125(0, 0, 0)] = TM::identity() * mass_matrix[n];
    }
    // force term
    T force_scale = dt * dt;
    bool proj = cfg.project != 0;
    const bool wd = wide();
    std::vector<double> wval; // wide-sums variant: the force term of every block in double, added to the inertia term and rounded once
    if (wd) wval.assign((size_t)num_nodes * 125 * 9, 0.0);
    for_each_particle_colored([&](int g, int i) {
        CorotatedScratch<T> s;
        corotated_update_scratch(F[i], mu[i], lambda[i], proj, s);
        T ddF[81];
        corotated_first_piola_derivative(s, ddF);
        const TM& Fn_local = Fn[i];
        TM FnT = Fn_local.transpose();
        Spline sp;
        compute_spline(X[i], sp);
        TV cached_w[27];
        int cached_node[27][3];
        int cached_idx[27];
        int cnt = 0;
        iterate_kernel(sp, g, particle_base_offset[i], [&](const int* node, T, const TV& dw, Node& gs) {
            if (gs.idx < 0) return;
            cached_w[cnt] = FnT * dw;
            cached_node[cnt][0] = node[0], cached_node[cnt][1] = node[1], cached_node[cnt][2] = node[2];
            cached_idx[cnt++] = (int)gs.idx;
        });
        for (int a = 0; a < cnt; ++a) {
            const TV& wi = cached_w[a];
            int dofi = cached_idx[a];
            for (int b = 0; b < cnt; ++b) {
                const TV& wj = cached_w[b];
                int dofj = cached_idx[b];
                if (dofj < dofi) continue;
                TM dFdX = TM::zero();
                for (int q = 0; q < 3; ++q)
                    for (int v = 0; v < 3; ++v) {
                        T ww = wi(v) * wj(q);
                        // ddF.block<3,3>(3*v, 3*q)
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) dFdX(r, c) += ddF[(3 * v + r) + 9 * (3 * q + c)] * ww;
                    }
                TM delta = dFdX * (force_scale * vol[i]);
                size_t sij = (size_t)dofi * 125 + linear_offset125(cached_node[a][0] - cached_node[b][0], cached_node[a][1] - cached_node[b][1], cached_node[a][2] - cached_node[b][2]);
                A.entryCol[sij] = dofj;
                if (wd)
                    for (int k = 0; k < 9; ++k) wval[sij * 9 + k] += (double)delta.a[k];
                else
                    A.entryVal[sij] += delta;
                if (dofi != dofj) {
                    size_t sji = (size_t)dofj * 125 + linear_offset125(cached_node[b][0] - cached_node[a][0], cached_node[b][1] - cached_node[a][1], cached_node[b][2] - cached_node[a][2]);
                    A.entryCol[sji] = dofi;
                    TM dT = delta.transpose();
                    if (wd)
                        for (int k = 0; k < 9; ++k) wval[sji * 9 + k] += (double)dT.a[k];
                    else
                        A.entryVal[sji] += dT;
                }
            }
        }
    });
    if (wd) {
#pragma omp parallel for schedule(static)
        for (size_t e = 0; e < A.entryVal.size(); ++e)
            for (int k = 0; k < 9; ++k) A.entryVal[e].a[k] = (T)((double)A.entryVal[e].a[k] + wval[e * 9 + k]);
        std::vector<double>().swap(wval);
    }
    if (sharded()) {
        allreduce(A.entryVal.data(), (int64_t)A.entryVal.size() * 9, REAL);
        allreduce(A.entryCol.data(), (int64_t)A.entryCol.size(), HOT_COMM_I32, HOT_COMM_MAX); // -1 where no rank's particle couples the pair
    }
    // BC projection of the assembled system (:554-593), or only the padding rule (:594-602)
#pragma omp parallel for schedule(static)
    for (int i = 0; i < num_nodes; ++i) {
        size_t st = (size_t)i * 125, ed = st + 125;
        int ic = cfg.systemBCProject ? bc_of_node[i] : -1;
        bool iCollide = ic >= 0;
        bool iSlip = iCollide ? collision_nodes[ic].shouldRotate : false;
        for (; st < ed; ++st) {
            int j = A.entryCol[st];
            if (j == -1) {
                A.entryCol[st] = i > 0 ? 0 : 1;
                continue;
            }
            if (!cfg.systemBCProject) continue;
            int jc = bc_of_node[j];
            bool jCollide = jc >= 0;
            if (!iCollide && !jCollide) continue;
            bool jSlip = jCollide ? collision_nodes[jc].shouldRotate : false;
            TM& val = A.entryVal[st];
            if ((iCollide && !iSlip) || (jCollide && !jSlip)) {
                val = (j == i) ? TM::identity() : TM::zero();
                continue;
            }
            if (iSlip) val = collision_nodes[ic].R * val;
            if (jSlip) val = val * collision_nodes[jc].Rinv;
            if (iSlip) val(0, 0) = 0, val(0, 1) = 0, val(0, 2) = 0;
            if (jSlip) val(0, 0) = 0, val(1, 0) = 0, val(2, 0) = 0;
            if (i == j) val(0, 0) = 1;
        }
    }
}

// reference SquareMatrix::buildDiagonal (Projects/multigrid/SquareMatrix.h:301-324)
template <class T>
void Sim<T>::build_diagonal(EllMat<T>& m, int opt)
{
    int n = m.nrows;
    m.diagonalVal.resize(n), m.diagonalBlock.resize(n);
    if (opt == 0) m.diagonalEntry.resize(n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        TM d = TM::zero();
        for (size_t idx = (size_t)i * m.colsize; idx < (size_t)(i + 1) * m.colsize; ++idx)
            if (m.entryCol[idx] == i) d += m.entryVal[idx];
        m.diagonalVal[i] = d;
        if (opt == 0) {
            TM e = TM::zero();
            for (int k = 0; k < 3; ++k) e(k, k) = 1 / d(k, k);
            m.diagonalEntry[i] = e;
        }
        m.diagonalBlock[i] = inverse(d);
    }
}

// reference SquareMatrix::multiply (SquareMatrix.h:477-487): every padded slot is multiplied
template <class T>
void Sim<T>::multiply(const EllMat<T>& m, const std::vector<TV>& x, std::vector<TV>& b)
{
    b.resize(m.nrows);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m.nrows; ++i) {
        TV sum = TV::zero();
        for (size_t idx = (size_t)i * m.colsize; idx < (size_t)(i + 1) * m.colsize; ++idx) sum += m.entryVal[idx] * x[m.entryCol[idx]];
        b[i] = sum;
    }
}

// reference markColors lambda (Projects/multigrid/MultigridPreconditioner.h:582-605): always 4^3 blocks,
// colour = parity bits of the block coordinates, block ids in first-touch order, 1-based index in block
template <class T>
void Sim<T>::mark_colors(const std::vector<std::array<int, 3>>& coords, EllMat<T>& m)
{
    using ULL = unsigned long long;
    constexpr ULL hash_seed = 100007;
    m.colorOrder.resize(coords.size());
    for (auto& b : m.coloredBlockDofs) b.clear();
    std::array<int, 8> blockCnts;
    blockCnts.fill(0);
    std::array<std::unordered_map<ULL, int>, 8> blockIds;
    for (int i = 0; i < (int)coords.size(); ++i) {
        int bi[3];
        for (int d = 0; d < 3; ++d) bi[d] = coords[i][d] >> 2;
        int color = ((bi[0] & 1) << 2) | ((bi[1] & 1) << 1) | (bi[2] & 1);
        ULL key = (ULL)bi[0] * hash_seed * hash_seed + (ULL)bi[1] * hash_seed + (ULL)bi[2];
        auto it = blockIds[color].find(key);
        int blockId;
        if (it == blockIds[color].end()) {
            blockId = blockCnts[color]++;
            blockIds[color][key] = blockId;
            m.coloredBlockDofs[color].emplace_back();
        }
        else
            blockId = it->second;
        m.coloredBlockDofs[color][blockId].push_back(i);
        m.colorOrder[i] = { color, blockId, (int)m.coloredBlockDofs[color][blockId].size() };
    }
}

// reference SquareMatrix::buildCoarseMatrix (SquareMatrix.h:526-571): out = l * r as padded ELL.  The
// reference orders the slots of a row by std::unordered_map iteration; here they are ordered by column
// (the set of (col, value) pairs per row is identical; slot order is not part of the parity contract).
template <class T>
void Sim<T>::build_product(EllMat<T>& out, const EllMat<T>& l, const EllMat<T>& r)
{
    int n = l.nrows;
    std::vector<std::vector<std::pair<int, TM>>> rows(n);
    int colsize = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(max : colsize)
    for (int i = 0; i < n; ++i) {
        std::vector<std::pair<int, TM>> tmp;
        tmp.reserve((size_t)l.colsize * r.colsize);
        for (size_t j = (size_t)i * l.colsize; j < (size_t)(i + 1) * l.colsize; ++j) {
            int jj = l.entryCol[j];
            for (size_t k = (size_t)jj * r.colsize; k < (size_t)(jj + 1) * r.colsize; ++k) tmp.emplace_back(r.entryCol[k], l.entryVal[j] * r.entryVal[k]);
        }
        std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, TM>& a, const std::pair<int, TM>& b) { return a.first < b.first; });
        auto& row = rows[i];
        for (auto& p : tmp) {
            if (!row.empty() && row.back().first == p.first)
                row.back().second += p.second;
            else
                row.push_back(p);
        }
        colsize = std::max(colsize, (int)row.size());
    }
    out.colsize = colsize;
    out.nrows = n;
    out.entryCol.resize((size_t)n * colsize);
    out.entryVal.resize((size_t)n * colsize);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        size_t idx = (size_t)i * colsize;
        for (auto& p : rows[i]) {
            out.entryCol[idx] = p.first;
            out.entryVal[idx] = p.second;
            ++idx;
        }
        for (; idx < (size_t)(i + 1) * colsize; ++idx) {
            out.entryCol[idx] = i > 0 ? 0 : 1;
            out.entryVal[idx] = TM::zero();
        }
    }
}

// reference SquareMatrix::buildTransposeMatrix (SquareMatrix.h:573-607): pattern transpose, 3x3 values are
// NOT transposed (they are w*I for the prolongation)
template <class T>
void Sim<T>::build_transpose(EllMat<T>& out, const EllMat<T>& l, int rowcnt)
{
    if (fair_flag()) {
        // "fair" CPU-baseline variant: the same transpose as a counting pass, a prefix sum and a parallel fill; every row is then sorted
        // by column and duplicates are added in ascending source-row order, which is the order the std::map version below adds them in
        std::vector<int> cnt(rowcnt + 1, 0);
        for (size_t j = 0; j < (size_t)l.nrows * l.colsize; ++j) ++cnt[l.entryCol[j] + 1];
        for (int r = 0; r < rowcnt; ++r) cnt[r + 1] += cnt[r];
        std::vector<std::pair<int, TM>> flat(cnt[rowcnt]);
        {
            std::vector<int> pos(cnt.begin(), cnt.end() - 1);
            for (int i = 0; i < l.nrows; ++i) // rows ascending: a column's candidates end up in ascending source-row order
                for (size_t j = (size_t)i * l.colsize; j < (size_t)(i + 1) * l.colsize; ++j) flat[pos[l.entryCol[j]]++] = { i, l.entryVal[j] };
        }
        std::vector<int> width(rowcnt);
#pragma omp parallel for schedule(static)
        for (int r = 0; r < rowcnt; ++r) { // merge equal columns in place
            int w = cnt[r];
            for (int k = cnt[r]; k < cnt[r + 1]; ++k) {
                if (w > cnt[r] && flat[w - 1].first == flat[k].first)
                    flat[w - 1].second += flat[k].second;
                else
                    flat[w++] = flat[k];
            }
            width[r] = w - cnt[r];
        }
        int colsize = 0;
        for (int r = 0; r < rowcnt; ++r) colsize = std::max(colsize, width[r]);
        out.colsize = colsize, out.nrows = rowcnt;
        out.entryCol.resize((size_t)rowcnt * colsize), out.entryVal.resize((size_t)rowcnt * colsize);
#pragma omp parallel for schedule(static)
        for (int r = 0; r < rowcnt; ++r) {
            size_t idx = (size_t)r * colsize;
            for (int k = 0; k < width[r]; ++k, ++idx) out.entryCol[idx] = flat[cnt[r] + k].first, out.entryVal[idx] = flat[cnt[r] + k].second;
            for (; idx < (size_t)(r + 1) * colsize; ++idx) out.entryCol[idx] = r > 0 ? 0 : 1, out.entryVal[idx] = TM::zero();
        }
        return;
    }
    std::vector<std::map<int, TM>> data(rowcnt);
    for (int i = 0; i < l.nrows; ++i)
        for (size_t j = (size_t)i * l.colsize; j < (size_t)(i + 1) * l.colsize; ++j) {
            int jj = l.entryCol[j];
            auto it = data[jj].find(i);
            if (it == data[jj].end())
                data[jj][i] = l.entryVal[j];
            else
                it->second += l.entryVal[j];
        }
    int colsize = 0;
    for (auto& d : data) colsize = std::max(colsize, (int)d.size());
    out.colsize = colsize;
    out.nrows = rowcnt;
    out.entryCol.resize((size_t)rowcnt * colsize);
    out.entryVal.resize((size_t)rowcnt * colsize);
    for (int i = 0; i < rowcnt; ++i) {
        size_t idx = (size_t)i * colsize;
        for (auto& p : data[i]) {
            out.entryCol[idx] = p.first;
            out.entryVal[idx] = p.second;
            ++idx;
        }
        for (; idx < (size_t)(i + 1) * colsize; ++idx) {
            out.entryCol[idx] = i > 0 ? 0 : 1;
            out.entryVal[idx] = TM::zero();
        }
    }
}

// reference SquareMatrix::estimate2norm (SquareMatrix.h:375-475, the active #else branch): power iteration on A*A from
// a +-1 start vector.  The reference seeds the signs with srand(time(NULL)); here they are a fixed hash of the entry
// index (the converged estimate does not depend on the start within the 1e-6 stopping tolerance).
static inline int cheb_sign(int i, int d) { return (((unsigned)(3 * i + d) * 2654435761u) >> 16) & 1u ? 1 : -1; }
template <class T>
void Sim<T>::estimate_2norm(EllMat<T>& A, T tol)
{
    constexpr int MaxIters = 512;
    int n = A.nrows;
    std::vector<TV> v(n), x(n);
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) v[i].a[d] = (T)cheb_sign(i, d);
    multiply(A, v, x);
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) x[i].a[d] = std::abs(x[i].a[d]);
    T e = std::sqrt(dot_product(x, x));
    if (e == 0) {
        A.lMin = A.lMax = 0;
        return;
    }
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) x[i].a[d] /= e;
    T e0 = 0;
    int iter = 0;
    for (; iter < MaxIters && std::abs(e - e0) > tol * e; ++iter) {
        e0 = e;
        multiply(A, x, v);
        multiply(A, v, x);
        T normx = std::sqrt(dot_product(x, x));
        e = normx / std::sqrt(dot_product(v, v));
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < 3; ++d) x[i].a[d] /= normx;
    }
    A.lMax = e;
    A.lMin = A.lMax / 30; // "experience" (:473)
}

// reference MultigridBuilder::build (MultigridPreconditioner.h:554-703), kernel_range == 2 (trilinear P,
// linear_weight_template :445-466), R = P^T, A_{l+1} = R (A_l P)
template <class T>
void Sim<T>::build_mg()
{
    using ULL = unsigned long long;
    constexpr ULL hash_seed = 100007;
    int levelCnt = cfg.levelCnt;
    sysmats.resize(1);
    promats.clear(), resmats.clear();
    level_coords.resize(1);
    const bool baseline = cfg.useBaselineMultigrid != 0;
    bool colors = baseline || (cfg.coarseSolver == 5 || cfg.smoother == 5) || cfg.coarseSolver == 7; // (7: the factorisation runs in the smoother's order)
    build_diagonal(sysmats[0], cfg.Ainv);
    if (colors) mark_colors(level_coords[0], sysmats[0]);
    if (cfg.coarseSolver == 7 && levelCnt == 1) setup_ic(sysmats[0], level_coords[0]); // :612-613
    if (!baseline && ((cfg.coarseSolver == 6 && levelCnt == 1) || (cfg.smoother == 6 && levelCnt > 1))) estimate_2norm(sysmats[0], (T)1e-6); // :610-611
    const T w1d[2][3] = { { 0, 1, 0 }, { 0, (T)0.5, (T)0.5 } };
    for (int level = 0; level < levelCnt - 1; ++level) {
        const auto& coords = level_coords[level];
        std::vector<std::array<int, 3>> new_coords;
        std::unordered_map<ULL, int> new_coord2id;
        Sim<T>* grid = nullptr;
        if (baseline) {
            // MultigridSimulation::particlesToMultigrids (Projects/multigrid/MultigridSimulation.inl:345-456): the coarse level is
            // an MPM grid of spacing 2^(level+1) dx — sortParticlesAndPolluteMultigrid (:40-124), mass P2G (:404-428),
            // buildMultigridBoundaries (:126-161), buildMultigridMatrices (:163-342) with the particles' dP/dF at the current F
            while ((int)gmg.size() <= level) gmg.emplace_back(new Sim<T>());
            grid = gmg[level].get();
            grid->cfg = cfg;
            grid->cfg.levelCnt = 1, grid->cfg.useBaselineMultigrid = 0;
            grid->cfg.dx = cfg.dx * (double)(1 << (level + 1));
            grid->dx = (T)grid->cfg.dx; // (curdx *= 2)
            grid->gravity = gravity;
            grid->Np = Np, grid->X = X, grid->Vel = Vel, grid->mass = mass, grid->vol = vol, grid->mu = mu, grid->lambda = lambda, grid->Jp = Jp, grid->C = C;
            grid->F = Fn;
            grid->cobjs = cobjs, grid->hs_origin = hs_origin, grid->hs_normal = hs_normal;
            grid->collision_nodes.clear();
            grid->sort_particles();
            grid->particles_to_grid();
            grid->begin_step(dt);
            grid->F = F;
            grid->build_matrix();
            new_coords = grid->id2coord;
            for (int j = 0; j < (int)new_coords.size(); ++j) new_coord2id[(ULL)new_coords[j][0] * hash_seed * hash_seed + (ULL)new_coords[j][1] * hash_seed + (ULL)new_coords[j][2]] = j;
        }
        EllMat<T> P;
        P.colsize = 8;
        P.nrows = (int)coords.size();
        P.entryCol.resize(coords.size() * 8);
        P.entryVal.resize(coords.size() * 8);
        std::vector<int> cstart;
        if (sharded() && (int)level_nstart.size() > level) cstart.assign(comm.size + 1, -1);
        for (int i = 0; i < (int)coords.size(); ++i) {
            for (int r = 0; r < (int)cstart.size(); ++r) // a rank's coarse id prefix: the coarse nodes first touched by fine ids below its fine prefix
                if (cstart[r] < 0 && i >= level_nstart[level][r]) cstart[r] = (int)new_coords.size();
            int x = coords[i][0], y = coords[i][1], z = coords[i][2];
            for (int new_x = x / 2; new_x <= x / 2 + 1; ++new_x)
                for (int new_y = y / 2; new_y <= y / 2 + 1; ++new_y)
                    for (int new_z = z / 2; new_z <= z / 2 + 1; ++new_z) {
                        int linear_idx = (new_x - x / 2) * 4 + (new_y - y / 2) * 2 + new_z - z / 2;
                        T weight = w1d[x & 1][new_x - x / 2 + 1] * w1d[y & 1][new_y - y / 2 + 1] * w1d[z & 1][new_z - z / 2 + 1];
                        if (weight == 0) {
                            P.entryCol[(size_t)i * 8 + linear_idx] = P.entryCol[(size_t)i * 8];
                            P.entryVal[(size_t)i * 8 + linear_idx] = TM::zero();
                            continue;
                        }
                        ULL key = (ULL)new_x * hash_seed * hash_seed + (ULL)new_y * hash_seed + (ULL)new_z;
                        auto it = new_coord2id.find(key);
                        int j;
                        if (it == new_coord2id.end()) {
                            if (baseline) throw std::runtime_error("baseline multigrid: a fine node's parent is not a coarse-grid DOF (MultigridSimulation.inl:207)");
                            new_coords.push_back({ new_x, new_y, new_z });
                            j = (int)new_coords.size() - 1;
                            new_coord2id[key] = j;
                        }
                        else
                            j = it->second;
                        P.entryCol[(size_t)i * 8 + linear_idx] = j;
                        P.entryVal[(size_t)i * 8 + linear_idx] = TM::identity() * weight;
                    }
        }
        if (!cstart.empty()) {
            for (auto& c : cstart)
                if (c < 0) c = (int)new_coords.size();
            level_nstart.resize(level + 1);
            level_nstart.push_back(cstart);
        }
        EllMat<T> R;
        build_transpose(R, P, (int)new_coords.size());
        EllMat<T> AP, RAP;
        if (baseline)
            RAP = std::move(grid->sysmats[0]);
        else {
            build_product(AP, sysmats[level], P);
            build_product(RAP, R, AP);
        }
        build_diagonal(RAP, cfg.Ainv);
        if (colors) mark_colors(new_coords, RAP);
        if (cfg.coarseSolver == 7 && level + 2 == levelCnt) setup_ic(RAP, new_coords); // :684-685
        if (!baseline && ((cfg.coarseSolver == 6 && level + 2 == levelCnt) || (cfg.smoother == 6 && level + 2 < levelCnt))) estimate_2norm(RAP, (T)1e-6); // :682-683
        promats.push_back(std::move(P));
        resmats.push_back(std::move(R));
        sysmats.push_back(std::move(RAP));
        level_coords.push_back(std::move(new_coords));
    }
    compute_owners(); // sharded: who sweeps which colour block (hot_config.shard_owner)
    int L = (int)sysmats.size();
    mg_residuals.resize(L), mg_initialResiduals.resize(L), mg_sols.resize(L), mg_dus.resize(L), mg_dAus.resize(L), mg_tmps.resize(L);
    for (int l = 0; l < L; ++l) {
        size_t n = sysmats[l].nrows;
        mg_residuals[l].assign(n, TV::zero()), mg_initialResiduals[l].assign(n, TV::zero()), mg_sols[l].assign(n, TV::zero());
        mg_dus[l].assign(n, TV::zero()), mg_dAus[l].assign(n, TV::zero()), mg_tmps[l].assign(n, TV::zero());
    }
}

// scale_diagonal_{entry,block}_inverse: MultigridPreconditioner.h:143-154, selected by Ainv (:485-495)
template <class T>
void Sim<T>::scaler(const std::vector<TV>& r, std::vector<TV>& mr, const EllMat<T>& A) const
{
    mr.resize(r.size());
    const auto& D = cfg.Ainv == 0 ? A.diagonalEntry : A.diagonalBlock;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < (int)r.size(); ++i) mr[i] = D[i] * r[i];
}

template <class T>
static inline T dot_product(const std::vector<V3<T>>& a, const std::vector<V3<T>>& b)
{
    // reference dotProduct is a serial Eigen reduction (MultigridPreconditioner.h:155-158)
    T s = 0;
    if (sizeof(T) == 4 && wide_flag()) { // wide-sums variant (sim_core.hpp): the products in T, their sum in double
        double w = 0;
        for (size_t i = 0; i < a.size(); ++i) w += (double)(a[i].a[0] * b[i].a[0]) + (double)(a[i].a[1] * b[i].a[1]) + (double)(a[i].a[2] * b[i].a[2]);
        return (T)w;
    }
    if (fair_flag()) {
#pragma omp parallel for schedule(static) reduction(+ : s)
        for (size_t i = 0; i < a.size(); ++i) s += a[i].dot(b[i]);
        return s;
    }
    for (size_t i = 0; i < a.size(); ++i) s += a[i].dot(b[i]);
    return s;
}

template <class T>
static inline int color_comp(const std::array<int, 3>& a, const std::array<int, 3>& b)
{
    for (int i = 0; i < 3; ++i) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0; // the reference falls off the end here (UB, SquareMatrix.h:39-46); equal keys multiply a zero vector
}

// coarseSolver 7 (top level only).  The reference hands the top-level matrix to Eigen::IncompleteCholesky (SquareMatrix.h:35,224-256;
// IC_smooth MultigridPreconditioner.h:320-323: u = ICSolver.solve(r), once): Eigen's left-looking IC with AMD ordering, row / column
// scaling and a shift-and-retry loop.  Eigen is absent here and its AMD tie-breaking cannot be restated, so this is NOT that
// factorisation but one of the same family, shared bit for bit by the HIP library (hot_amd/csrc/mg_ic.hip): incomplete Cholesky with zero
// fill by 3x3 BLOCKS on the 125-stencil pattern, rows in the smoother's order (colour, first-touch 4^3 block, node: the order gs_smooth
// sweeps in), Eigen's shift strategy (factor A + shift * diag(A), shift 0 first, then 1e-3 doubled until every 3x3 pivot is positive
// definite).  Parity with the reference can therefore only be claimed on the converged solution of the outer solve.
template <class T>
void Sim<T>::setup_ic(EllMat<T>& m, const std::vector<std::array<int, 3>>& coords)
{
    using ULL = unsigned long long;
    const int n = m.nrows;
    std::unordered_map<ULL, int> id;
    id.reserve((size_t)n * 2);
    auto key = [](int x, int y, int z) { return ((ULL)(unsigned)x << 42) | ((ULL)(unsigned)y << 21) | (ULL)(unsigned)z; };
    for (int i = 0; i < n; ++i) id[key(coords[i][0], coords[i][1], coords[i][2])] = i;
    m.icNbr.assign((size_t)n * 125, -1);
    std::vector<TM> A((size_t)n * 125, TM::zero()); // the matrix by stencil slot: slot (dx+2)*25 + (dy+2)*5 + dz+2 holds column coord_i - d
    for (int i = 0; i < n; ++i) {
        for (int s = 0; s < 125; ++s) {
            const int x = coords[i][0] - (s / 25 - 2), y = coords[i][1] - ((s / 5) % 5 - 2), z = coords[i][2] - (s % 5 - 2);
            if ((x | y | z) < 0) continue;
            auto it = id.find(key(x, y, z));
            if (it != id.end()) m.icNbr[(size_t)i * 125 + s] = it->second;
        }
        for (size_t e = (size_t)i * m.colsize; e < (size_t)(i + 1) * m.colsize; ++e) {
            const int j = m.entryCol[e];
            const int dx = coords[i][0] - coords[j][0], dy = coords[i][1] - coords[j][1], dz = coords[i][2] - coords[j][2];
            if (std::abs(dx) > 2 || std::abs(dy) > 2 || std::abs(dz) > 2) continue; // padding slot (aliases column 0 / 1 with a zero block)
            A[(size_t)i * 125 + (dx + 2) * 25 + (dy + 2) * 5 + dz + 2] += m.entryVal[e];
        }
    }
    // factorisation order = the smoother's: (colour, block, index in block)
    m.icOrder.resize(n);
    for (int i = 0; i < n; ++i) m.icOrder[i] = i;
    std::sort(m.icOrder.begin(), m.icOrder.end(), [&](int a, int b) { return m.colorOrder[a] < m.colorOrder[b]; });
    std::vector<int> rank(n);
    for (int p = 0; p < n; ++p) rank[m.icOrder[p]] = p;
    auto chol3 = [](const TM& D, TM& L) { // lower Cholesky factor of a symmetric 3x3; false if a pivot is not positive
        L = TM::zero();
        T l00 = D(0, 0);
        if (!(l00 > 0)) return false;
        l00 = std::sqrt(l00);
        const T l10 = D(1, 0) / l00, l20 = D(2, 0) / l00;
        T l11 = D(1, 1) - l10 * l10;
        if (!(l11 > 0)) return false;
        l11 = std::sqrt(l11);
        const T l21 = (D(2, 1) - l20 * l10) / l11;
        T l22 = D(2, 2) - l20 * l20 - l21 * l21;
        if (!(l22 > 0)) return false;
        l22 = std::sqrt(l22);
        L(0, 0) = l00, L(1, 0) = l10, L(2, 0) = l20, L(1, 1) = l11, L(2, 1) = l21, L(2, 2) = l22;
        return true;
    };
    T shift = 0;
    for (int attempt = 0; attempt < 60; ++attempt) {
        m.icL.assign((size_t)n * 125, TM::zero());
        m.icD.assign(n, TM::zero()), m.icDinv.assign(n, TM::zero());
        bool ok = true;
        for (int p = 0; p < n && ok; ++p) {
            const int i = m.icOrder[p];
            // the row's lower neighbours in factorisation order
            std::vector<std::pair<int, int>> lower; // (rank, slot)
            for (int s = 0; s < 125; ++s) {
                const int j = m.icNbr[(size_t)i * 125 + s];
                if (j >= 0 && rank[j] < p) lower.emplace_back(rank[j], s);
            }
            std::sort(lower.begin(), lower.end());
            for (auto& ls : lower) {
                const int s = ls.second, j = m.icNbr[(size_t)i * 125 + s];
                TM S = A[(size_t)i * 125 + s];
                for (int t = 0; t < 125; ++t) { // common earlier neighbours k of i and j (slot order: the order the device's lanes hold them in)
                    const int k = m.icNbr[(size_t)j * 125 + t];
                    if (k < 0 || rank[k] >= rank[j]) continue;
                    const int dx = coords[i][0] - coords[k][0], dy = coords[i][1] - coords[k][1], dz = coords[i][2] - coords[k][2];
                    if (std::abs(dx) > 2 || std::abs(dy) > 2 || std::abs(dz) > 2) continue;
                    S = S - m.icL[(size_t)i * 125 + (dx + 2) * 25 + (dy + 2) * 5 + dz + 2] * m.icL[(size_t)j * 125 + t].transpose();
                }
                m.icL[(size_t)i * 125 + s] = S * m.icDinv[j].transpose(); // L_ij L_jj^T = S
            }
            TM D = A[(size_t)i * 125 + 62] * ((T)1 + shift);
            for (auto& ls : lower) D = D - m.icL[(size_t)i * 125 + ls.second] * m.icL[(size_t)i * 125 + ls.second].transpose();
            TM L;
            if (!chol3(D, L)) {
                ok = false;
                break;
            }
            m.icD[i] = L, m.icDinv[i] = inverse(L);
        }
        if (ok) {
            m.icShift = shift;
            return;
        }
        shift = shift == 0 ? (T)1e-3 : shift * 2;
    }
    throw std::runtime_error("incomplete Cholesky: no positive definite factorisation found");
}
template <class T>
void Sim<T>::solve_ic(const EllMat<T>& m, const std::vector<TV>& r, std::vector<TV>& u) const
{
    const int n = m.nrows;
    std::vector<int> rank(n);
    for (int p = 0; p < n; ++p) rank[m.icOrder[p]] = p;
    std::vector<TV> y(n);
    for (int p = 0; p < n; ++p) { // L y = r
        const int i = m.icOrder[p];
        TV s = r[i];
        for (int t = 0; t < 125; ++t) {
            const int j = m.icNbr[(size_t)i * 125 + t];
            if (j >= 0 && rank[j] < p) s = s - m.icL[(size_t)i * 125 + t] * y[j];
        }
        y[i] = m.icDinv[i] * s;
    }
    u.resize(n);
    for (int p = n - 1; p >= 0; --p) { // L^T u = y
        const int i = m.icOrder[p];
        TV s = y[i];
        for (int t = 0; t < 125; ++t) { // rows j after i that hold i in their lower part: L_ji sits in row j at the mirrored slot
            const int j = m.icNbr[(size_t)i * 125 + t];
            if (j >= 0 && rank[j] > p) s = s - m.icL[(size_t)j * 125 + (124 - t)].transpose() * u[j];
        }
        u[i] = m.icDinv[i].transpose() * s;
    }
}

// smoothers: reference MultigridPreconditioner.h:160-173 (jacobi), :174-189 (optimal jacobi), :190-226 (cg),
// :227-264 (chebyshev), :266-318 (gs)
template <class T>
void Sim<T>::smooth(int kind, int level, std::vector<TV>& u, std::vector<TV>& r, std::vector<TV>& du, std::vector<TV>& dAu, int iterations, T tolerance)
{
    EllMat<T>& A = sysmats[level];
    int n = A.nrows;
    // A.project is the objective's projection only on level 0 when the system is NOT BC-projected (:690-693)
    auto Aproject = [&](std::vector<TV>& v) {
        if (level == 0 && !cfg.systemBCProject) project(v);
    };
    if (kind == 7) { // IC_smooth (MultigridPreconditioner.h:320-323): u = IC^-1 r, once; r is left alone
        solve_ic(A, r, u);
        return;
    }
    if (kind == 0) {
        for (; iterations--;) {
            scaler(r, du, A);
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) du[i] = du[i] * (T)cfg.topomega;
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) u[i] += du[i];
            multiply(A, du, dAu);
            Aproject(dAu);
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) r[i] -= dAu[i];
        }
    }
    else if (kind == 1) {
        for (; iterations--;) {
            if (std::sqrt(dot_product(r, r)) < tolerance) break;
            scaler(r, du, A);
            multiply(A, du, dAu);
            Aproject(dAu);
            T omega = dot_product(du, r) / dot_product(du, dAu);
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) u[i] += du[i] * omega, r[i] -= dAu[i] * omega;
        }
    }
    else if (kind == 2) {
        std::vector<TV>& z = mg_tmps[level];
        scaler(mg_initialResiduals[level], z, A);
        T zTrk0 = dot_product(z, mg_initialResiduals[level]);
        scaler(r, z, A);
        du = z;
        T zTrk = dot_product(z, r);
        double cgratio = 0.5;
        tolerance = (T)(zTrk0 * cgratio * cgratio);
        int cnt = 0;
        for (; iterations--;) {
            if (zTrk < tolerance) break;
            multiply(A, du, dAu);
            Aproject(dAu);
            T omega = zTrk / dot_product(dAu, du);
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) u[i] += du[i] * omega, r[i] -= dAu[i] * omega;
            scaler(r, z, A);
            T zTrkPre = zTrk;
            zTrk = dot_product(z, r);
            T beta = zTrk / zTrkPre;
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) du[i] = z[i] + du[i] * beta;
            ++cnt;
        }
        stats.linear_iterations += cnt;
    }
    else if (kind == 5) {
        std::vector<TV>& hdu = mg_tmps[level];
        iterations = ((iterations + 1) >> 1);
        // sharded, partitioned level: a colour block is processed by its owner (owner_of_block: one of the ranks whose particles touch it); after each colour the owners' values are handed to everybody (colour-synchronous: the
        // sequence of updates every node sees is the single-rank one).  The exchange here is the simplest possible: an all-reduce
        // of a vector that is zero off the rank's own blocks.
        // hot_config.shard_gs = 1 (processor-block GS): no hand-off between the colours — a rank's rows see the other ranks' unknowns as
        // the zeros the sweep started from, i.e. the sweep is the symmetric GS of the rank's own diagonal block of A — and ONE exchange of
        // du after the backward sweep.  The couplings across ranks enter through the residual update r -= A du below (full rows).
        const bool part = partitioned(level);
        const bool rank_local = part && cfg.shard_gs != 0;
        auto mine = [&](const std::vector<int>& blockNodes) { return !part || owner_of_block(level, blockNodes) == comm.rank; };
        // hot_config.shard_gs = 2: the l1-scaled processor-block sweep (the product's rule, hot_amd/csrc/mg_build.hip k_l1_diag; Baker, Falgout, Kolev, Yang,
        // "Multigrid smoothers for ultraparallel computing", SIAM J. Sci. Comput. 33 (2011), section 6.2): the diagonal block of a row that couples to other
        // ranks' rows is D' = D + diag(sum over the off-rank columns j of the absolute row sums of A_ij) — the symmetric GS of the rank's own diagonal block
        // with that diagonal is a convergent smoother for every symmetric positive definite A, which the unscaled one is not.
        std::vector<TM> Dl1, Dl1inv;
        if (rank_local && cfg.shard_gs == 2) {
            std::vector<int> own_of(n, -1);
            for (int c = 0; c < 8; ++c)
                for (const auto& blockNodes : A.coloredBlockDofs[c]) {
                    const int o = owner_of_block(level, blockNodes);
                    for (int i : blockNodes) own_of[i] = o;
                }
            Dl1.assign(A.diagonalVal.begin(), A.diagonalVal.end());
            Dl1inv.assign(A.diagonalBlock.begin(), A.diagonalBlock.end());
            for (int i = 0; i < n; ++i) {
                if (own_of[i] != comm.rank) continue;
                T e[3] = { 0, 0, 0 };
                for (size_t st = (size_t)i * A.colsize; st < (size_t)(i + 1) * A.colsize; ++st) {
                    const int col = A.entryCol[st];
                    if (own_of[col] == comm.rank) continue;
                    for (int r = 0; r < 3; ++r)
                        for (int cc = 0; cc < 3; ++cc) e[r] += std::abs(A.entryVal[st](r, cc));
                }
                if (e[0] == 0 && e[1] == 0 && e[2] == 0) continue;
                for (int r = 0; r < 3; ++r) Dl1[i](r, r) += e[r];
                Dl1inv[i] = inverse(Dl1[i]);
            }
        }
        const std::vector<TM>& gsD = Dl1.empty() ? A.diagonalVal : Dl1;
        const std::vector<TM>& gsDinv = Dl1inv.empty() ? A.diagonalBlock : Dl1inv;
        auto exchange_colour = [&](std::vector<TV>& x, int c) {
            if (!part) return;
            std::vector<T> buf((size_t)n * 3, (T)0);
            for (const auto& blockNodes : A.coloredBlockDofs[c])
                if (mine(blockNodes))
                    for (int i : blockNodes)
                        for (int d = 0; d < 3; ++d) buf[3 * (size_t)i + d] = x[i](d);
            allreduce(buf.data(), (int64_t)buf.size(), REAL);
            for (const auto& blockNodes : A.coloredBlockDofs[c])
                for (int i : blockNodes) x[i] = TV{ { buf[3 * (size_t)i], buf[3 * (size_t)i + 1], buf[3 * (size_t)i + 2] } };
        };
        for (; iterations--;) {
            hdu.assign(n, TV::zero());
            for (int c = 0; c < 8; ++c) {
#pragma omp parallel for schedule(dynamic, 4)
                for (int bid = 0; bid < (int)A.coloredBlockDofs[c].size(); ++bid) {
                    const auto& blockNodes = A.coloredBlockDofs[c][bid];
                    if (!mine(blockNodes)) continue;
                    for (int ii = 0; ii < (int)blockNodes.size(); ++ii) {
                        int i = blockNodes[ii];
                        TV sigma = TV::zero();
                        for (size_t st = (size_t)i * A.colsize; st < (size_t)(i + 1) * A.colsize; ++st) {
                            int col = A.entryCol[st];
                            if (color_comp<T>(A.colorOrder[col], A.colorOrder[i]) < 0) sigma += A.entryVal[st] * hdu[col];
                        }
                        hdu[i] = gsDinv[i] * (r[i] - sigma);
                    }
                }
                if (!rank_local) exchange_colour(hdu, c);
            }
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) hdu[i] = gsD[i] * hdu[i];
            du.assign(n, TV::zero());
            for (int c = 7; c >= 0; --c) {
#pragma omp parallel for schedule(dynamic, 4)
                for (int bid = 0; bid < (int)A.coloredBlockDofs[c].size(); ++bid) {
                    const auto& blockNodes = A.coloredBlockDofs[c][bid];
                    if (!mine(blockNodes)) continue;
                    for (int ii = (int)blockNodes.size() - 1; ii >= 0; --ii) {
                        int i = blockNodes[ii];
                        TV sigma = TV::zero();
                        for (size_t st = (size_t)i * A.colsize; st < (size_t)(i + 1) * A.colsize; ++st) {
                            int col = A.entryCol[st];
                            if (color_comp<T>(A.colorOrder[col], A.colorOrder[i]) > 0) sigma += A.entryVal[st] * du[col];
                        }
                        du[i] = gsDinv[i] * (hdu[i] - sigma);
                    }
                }
                if (!rank_local) exchange_colour(du, c);
            }
            if (rank_local) { // the one hand-off of a rank-local symmetric sweep: every rank's du (zero off its own blocks) summed
                std::vector<T> buf((size_t)n * 3);
                for (int i = 0; i < n; ++i)
                    for (int d = 0; d < 3; ++d) buf[3 * (size_t)i + d] = du[i](d);
                allreduce(buf.data(), (int64_t)buf.size(), REAL);
                for (int i = 0; i < n; ++i) du[i] = TV{ { buf[3 * (size_t)i], buf[3 * (size_t)i + 1], buf[3 * (size_t)i + 2] } };
            }
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) u[i] += du[i];
            multiply(A, du, dAu);
            Aproject(dAu);
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) r[i] -= dAu[i];
        }
    }
    else if (kind == 6) {
        std::vector<TV>& p = mg_tmps[level];
        T d = (A.lMax + A.lMin) / 2, c = (A.lMax - A.lMin) / 2;
        int cnt = 1;
        iterations--;
        scaler(r, p, A);
        T alpha = 1 / d, beta;
        du = p;
        multiply(A, du, dAu);
        Aproject(dAu);
        HOT_FAIR_FOR
        for (int i = 0; i < n; ++i) u[i] += du[i] * alpha, r[i] -= dAu[i] * alpha;
        for (; iterations-- > 0; ++cnt) {
            scaler(r, p, A);
            beta = (T)0.5 * c * c * alpha * alpha;
            if (cnt > 1) beta *= (T)0.5;
            alpha = 1 / (d - beta / alpha);
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) du[i] = p[i] + du[i] * beta;
            multiply(A, du, dAu);
            Aproject(dAu);
            HOT_FAIR_FOR
            for (int i = 0; i < n; ++i) u[i] += du[i] * alpha, r[i] -= dAu[i] * alpha;
        }
    }
}

// reference MultigridOperator::operator() (MultigridPreconditioner.h:362-421) with setup_parameters (:525-551)
template <class T>
void Sim<T>::vcycle(const std::vector<TV>& in, std::vector<TV>& out)
{
    int levelCnt = (int)sysmats.size();
    int times = cfg.times, levelscale = cfg.levelscale;
    int splitLevel;
    auto downIter = [&](int level) { return times + level * levelscale; };
    std::function<int(int)> upIter, topIter;
    const bool baseline = cfg.useBaselineMultigrid != 0; // gs_smooth / cg_smooth / 10000 (MultigridSimulation.inl:446-453)
    if (baseline) {
        splitLevel = cfg.levelCnt - 1;
        upIter = downIter;
        topIter = [](int) { return 10000; };
    }
    else if (cfg.topDownMGS) {
        splitLevel = 1;
        upIter = [](int) { return 0; };
        topIter = [](int) { return 10000; };
    }
    else {
        splitLevel = cfg.levelCnt - 1;
        upIter = downIter;
        if (cfg.levelCnt == 1)
            topIter = upIter;
        else if (!(cfg.coarseSolver == 2 || cfg.coarseSolver == 6))
            topIter = [&](int level) { return (times + level * levelscale) * 3; };
        else
            topIter = [](int) { return 10000; };
    }
    auto tolTop = [&](int) { return (T)(cfg.cneps * cfg.cneps); };
    auto run = [&](bool regular, int level, std::vector<TV>& sol, int its) {
        smooth(regular ? (baseline ? 5 : cfg.smoother) : (baseline ? 2 : cfg.coarseSolver), level, sol, mg_residuals[level], mg_dus[level], mg_dAus[level], its, regular ? (T)0 : tolTop(level));
    };
    stats.vcycles++;
    mg_residuals[0] = in;
    if (cfg.systemBCProject && !baseline) // correctResidualProjection (MultigridPreconditioner.h:695 ; the baseline installs a no-op, .inl:455)
        for (int i = 0; i < (int)in.size(); ++i) mg_residuals[0][i] += dRhs[i];
    out.assign(in.size(), TV::zero());
    if (levelCnt > 1)
        multiply(resmats[0], mg_residuals[0], mg_initialResiduals[1]);
    else
        mg_initialResiduals[0] = mg_residuals[0];
    for (int l = 1; l < levelCnt - 1; ++l) multiply(resmats[l], mg_initialResiduals[l], mg_initialResiduals[l + 1]);
    int level;
    for (level = 0; level < levelCnt - 1; ++level) {
        std::vector<TV>& sol = level == 0 ? out : mg_sols[level];
        mg_level = level;
        run(level < splitLevel, level, sol, level < splitLevel ? upIter(level) : topIter(level));
        multiply(resmats[level], mg_residuals[level], mg_residuals[level + 1]);
        mg_sols[level + 1].assign(sysmats[level + 1].nrows, TV::zero());
    }
    mg_level = level;
    run(false, level, level == 0 ? out : mg_sols[level], topIter(level));
    for (--level; level >= 0; --level) {
        std::vector<TV>& sol = level == 0 ? out : mg_sols[level];
        mg_level = level;
        multiply(promats[level], mg_sols[level + 1], mg_dus[level]);
        HOT_FAIR_FOR
        for (size_t i = 0; i < sol.size(); ++i) sol[i] += mg_dus[level][i];
        multiply(sysmats[level], mg_dus[level], mg_dAus[level]);
        HOT_FAIR_FOR
        for (size_t i = 0; i < sol.size(); ++i) mg_residuals[level][i] -= mg_dAus[level][i];
        run(level < splitLevel, level, sol, level < splitLevel ? downIter(level) : topIter(level));
    }
}

} // namespace hot_oracle
