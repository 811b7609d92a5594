/*
 * hot_mi355x.h — C ABI of libhotmi355x.so: the MI355X-native (gfx950 / HIP) implementation of the HOT
 * per-timestep hot path (APIC P2G/G2P over the SPGrid block grid + Galerkin-multigrid-preconditioned
 * L-BFGS / projected-Newton inner solve).
 *
 * The reference (penn-graphics-research/HOT) has no FFI: its seams are C++ virtuals, std::function members
 * and function pointers inside one process.  Every entry point below names the reference
 * member function(s) it replaces (paths relative to the reference tree).  The header-only C++ adapter
 * include/hot_adapter.hpp re-exposes these as the reference's operator concept
 * (multiply / precondition / project / smoother function pointer, TVStack = 3 x N column-major).
 *
 * Conventions
 *   - Opaque context, one per host thread; no global state.  All functions return 0 on success and a
 *     negative hot_status on failure (the reference throws / asserts instead: Lib/Ziran/CS/Util/Debug.h:19,
 *     SPGrid_Utilities.cpp:75-85); hot_last_error() gives the message.
 *   - Scalars of arrays are `real` = float (dtype 0) or double (dtype 1) as chosen in hot_config.dtype; the
 *     executable of the reference hard-codes double (Projects/multigrid/main.cpp:12-13).
 *   - Array arguments may be HOST or DEVICE pointers (resolved with hipMemcpyDefault); outputs are written
 *     to wherever the pointer lives.  NULL output pointers are skipped.
 *   - Vectors over grid DOFs are "TVStack" layout: 3 x Nn column-major == xyz interleaved per node, node ids
 *     are the reference's g.idx numbering (Lib/MPM/MpmGrid.h:148-161).  3x3 matrices are column-major
 *     (Eigen default), particle attributes are array-of-structs in the caller's particle index order.
 */
#ifndef HOT_MI355X_H
#define HOT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hot_ctx hot_ctx;

enum hot_status {
    HOT_OK = 0,
    HOT_ERR_INVALID = -1, /* bad argument / call order (reference: ZIRAN_ASSERT) */
    HOT_ERR_DEVICE = -2, /* HIP runtime error */
    HOT_ERR_CAPACITY = -3, /* exceeds the reference's own int32 limits (MpmSimulationBase.cpp:1071-1072, ImplicitSolver.h:479-480) */
    HOT_ERR_NUMERIC = -4 /* NaN encountered (reference: FPE trap, main.cpp:36) */
};

/* Mirrors the knobs of HOTSettings (Projects/multigrid/Configurations.h:18-42) plus the few
 * MpmSimulationBase members the path reads (dx, gravity, apic_rpic_ratio, cfl: MpmSimulationBase.cpp:28-40). */
typedef struct hot_config {
    int32_t dtype; /* 0 = fp32 (GridState 64 B, 4x4x4 blocks), 1 = fp64 (128 B, 2x4x4 blocks) */
    int32_t device; /* HIP device ordinal */
    double dx;
    double gravity[3];
    double apic_rpic_ratio; /* 1 */
    double cfl; /* 0.6 */
    int32_t lsolver; /* 1 = projected Newton + MINRES, 2 = projected Newton + inexact (MG-)PCG, 3 = L-BFGS with MG initial Hessian (HOT) */
    int32_t Ainv; /* 0 inverse diagonal entries, 1 inverse 3x3 diagonal block, 2 lumped mass (lsolver 1 / 2 only, no hierarchy) */
    int32_t smoother; /* 0 damped Jacobi, 1 optimal Jacobi, 2 PCG, 5 symmetric coloured GS, 6 Chebyshev; 7 (incomplete Cholesky) is a top solver only: rejected as smoother inside a hierarchy, like the reference (MultigridPreconditioner.h:614) */
    int32_t coarseSolver; /* same option space, applied on the top level; 7 = incomplete Cholesky applied once (IC_smooth, MultigridPreconditioner.h:320-323): NOT Eigen's AMD-ordered IC, which cannot be
                             restated without Eigen, but block IC(0) on the stencil pattern in the smoother's order with Eigen's shift strategy (hot_amd/csrc/mg_ic.hip): parity on converged solutions only */
    int32_t levelCnt; /* -mg_level */
    int32_t times; /* -mg_times */
    int32_t levelscale; /* -mg_scale */
    double omega; /* -mg_omega (unused by the reference smoothers, kept for parity) */
    double topomega; /* damped-Jacobi factor, 0.1 */
    double cneps; /* -cneps */
    int32_t useCN; /* --usecn: characteristic-norm termination */
    int32_t project; /* --project: PSD-project dP/dF */
    int32_t systemBCProject; /* --bcproject */
    int32_t linesearch; /* --linesearch */
    int32_t matrixFree; /* --matfree (lsolver 1 / 2): H x by G2P -> dP -> P2G, block-diagonal preconditioner */
    int32_t boundaryType; /* -bc: 0 all sticky, 1 has slip */
    int32_t useAdaptiveHessian; /* --adaptiveH */
    int32_t topDownMGS;
    int32_t max_iterations; /* nonlinear iteration cap (reference scenes set 10000) */
    int32_t plasticity; /* 0 none, 1 VonMisesFixedCorotated, 2 SnowPlasticity */
    double yield_stress; /* von Mises */
    double snow[5]; /* psi, theta_c, theta_s, min_Jp, max_Jp */
    int32_t profile; /* 1: bracket every kernel launch with HIP events on the launch stream */
    int32_t debug_store; /* 1: also keep per-particle grad v for hot_get_particle_state (extra 9 stores per particle and pass) */
    int32_t useBaselineMultigrid; /* --baseline: geometric multigrid, every coarse level a real MPM grid of spacing 2^l dx whose matrix is re-rasterised from the particles */
    int32_t gs_chain; /* tuning override of the coloured-GS launch structure: 0 = by level size (default), 1 = one launch per colour, 2 = one chained launch per half sweep */
    int32_t gs_sub_block; /* tuning override: nodes of a 4^3 colour block one workgroup substitutes at a time, 0 = by level size, or 16 / 32 / 64 */
    int32_t shard_gs; /* sharded runs (hot_set_comm), coloured GS on a row-partitioned level: 0 = colour-synchronous (default): after every colour the
                         owners' new values are handed to all ranks, i.e. the reference's update order and single-rank iterates; 1 = processor-block:
                         a rank sweeps its own rows against its own rows only (couplings to other ranks' rows enter through the residual), one
                         exchange per symmetric sweep instead of sixteen; a different (still symmetric positive definite) smoother that needs large
                         sub-domains to converge; 2 = the same with the l1 norms of a row's off-rank couplings added to its diagonal block (D' = D +
                         diag(sum_j |A_ij| 1), Baker / Falgout / Kolev / Yang 2011): convergent for every SPD matrix whatever the sub-domain size — see DESIGN.md §7 */
    int32_t shard_replicated; /* sharded runs: 0 (default) = halo mode: DOF vectors live on the rows a rank owns plus the halo it reads, node tiles are summed
                                 between the ranks that share a block, inner products are summed with one small all-reduce per batch; 1 = the first-generation
                                 decomposition: every DOF vector replicated, whole-array all-reduce per scatter and all-gather per operator (kept for A/B) */
    int32_t ls_energy_only; /* line-search trials that evaluate nothing but the energy (psi from the invariants of F^T F — no SVD, no stress / trial-F stores —, up to 16
                               trials per pass: the trial F is affine in the step length), full state pass once at the accepted step: 0 = adaptive (default: from the second trial of a search on, and from the first when the previous search had to halve),
                               1 = never (every trial is a full pass), 2 = always */
    int32_t linear_iteration_cap; /* lsolver 1 / 2: iterations of one MINRES / PCG solve at most; 0 = the reference's 10000 (ImplicitSolver.h: the solvers' max_iterations).  A fixed
                                     small count makes two implementations stop at the same Lanczos step, whatever round-off does to the stopping test (parity tests) */
    int32_t shard_owner; /* sharded runs: which rank owns the rows of a 4^3 colour block.  2 = the rank whose particle range (a contiguous range of the SPGrid
                            page order) contains the block's own page, on a coarse level the page at the block's position on the finest grid: every rank owns the
                            rows inside its own particle range, both sides of a cut send the same amount of partial matrix rows, and the boundary between the ranks'
                            rows is as compact as the boundary between their particles; 1 = the rank whose particles first touch the block (rounds 2 - 4): the lower
                            rank of every cut owns all blocks the two share; 0 (default) = by the smoother: 2 under colour-synchronous sweeps (shard_gs = 0), 1 under
                            rank-local sweeps (shard_gs = 1), whose iteration counts stay within 15 % of the single-rank ones only under it (measured,
                            profiles/r05_shard_ownership.txt).  Other values are rejected */
    int32_t reserved[5];
} hot_config;

/* Version of the structures below and of hot_config: it changes whenever a field is added (fields are only ever appended).  hot_solve / hot_advance /
 * hot_advance_frame write a WHOLE hot_stats through the caller's pointer, so a caller compiled against an older header would be written past the end
 * of its structure: check hot_abi_version() == HOT_ABI_VERSION once after loading the library (hot_amd/binding.py does).
 *   5: hot_stats.comm_calls_index appended (round 5); 6: hot_config.shard_owner takes 0 / 1 / 2, hot_copy_bandwidth (round 6) */
#define HOT_ABI_VERSION 6
int hot_abi_version(void);

typedef struct hot_stats {
    int32_t iterations; /* nonlinear (L-BFGS / Newton) iterations of the last solve */
    int32_t converged;
    int32_t linesearch_trials; /* total updateState calls inside lineSearch */
    int32_t linear_iterations; /* MINRES / PCG iterations (lsolver 1 / 2) or total top-level PCG iterations (lsolver 3) */
    int32_t vcycles;
    int32_t dropped_pairs; /* L-BFGS curvature pairs dropped (y^T s <= 0, LBFGS.h:420-425) */
    int32_t num_nodes;
    int32_t num_levels;
    double final_scaled_residual; /* sqrt(sum |r_i|^2/tol_i^2 / Nn) if useCN else |r|_2 */
    double energy; /* incremental potential at the last accepted line-search point (0 when cfg.linesearch == 0: nothing evaluates it then) */
    double ms_sort, ms_p2g, ms_begin, ms_hessian, ms_mg_build, ms_solve, ms_g2p, ms_total; /* host wall clock, device-synchronised */
    /* sharded runs: what this rank handed to the collectives of hot_comm since the last hot_sort (hot_advance: during the step).  "index": integers that
     * describe the grid (block lists, node numbering, exchange lists: once per step); "data": floating-point payloads (tiles, halos, matrix rows, scalars) */
    int64_t comm_calls, comm_bytes_index, comm_bytes_data;
    int64_t comm_calls_index; /* of comm_calls: the collectives of the index structure (block lists, numbering, exchange lists) — the host waits for each of them */
} hot_stats;

void hot_default_config(hot_config* cfg); /* HOT's tog.sh command set: -lsolver 3 -Ainv 1 --project --linesearch --bcproject -mg_level 3 -mg_times 1 -coarseSolver 2 -smoother 5 --usecn -cneps 1e-7 */
int hot_create(const hot_config* cfg, hot_ctx** out);
void hot_destroy(hot_ctx* ctx);
const char* hot_last_error(hot_ctx* ctx);
int hot_sync(hot_ctx* ctx); /* hipStreamSynchronize on the context's stream */

/* ---- particles (the reference's DataManager columns X, V, "m", C, F, "element measure", and the
 *      CorotatedIsotropic (mu, lambda) per particle; Jp only for SnowPlasticity; Lib/Ziran/Math/Geometry/Particles.h:8-45) */
int hot_set_particles(hot_ctx*, int64_t Np, const void* X /*3Np*/, const void* V /*3Np*/, const void* mass /*Np*/,
    const void* C /*9Np, may be NULL = 0*/, const void* F /*9Np, NULL = I*/, const void* vol /*Np*/,
    const void* mu /*Np*/, const void* lambda /*Np*/, const void* Jp /*Np or NULL = 1*/);
int hot_get_particles(hot_ctx*, void* X, void* V, void* C, void* F, void* mu, void* lambda, void* Jp);

/* ---- MpmSimulationBase::sortParticlesAndPolluteGrid (Lib/MPM/MpmSimulationBase.cpp:1066-1137) */
int hot_sort(hot_ctx*);
int hot_get_counts(hot_ctx*, int64_t* Np, int32_t* Ng /*particle groups*/, int32_t* Nb /*touched blocks*/, int32_t* Nn /*active nodes, valid after hot_p2g*/);
/* bit-exact with the reference containers of the same names */
int hot_get_indexing(hot_ctx*, int32_t* particle_order /*Np*/, uint64_t* particle_base_offset /*Np*/,
    int32_t* particle_group /*2Ng: first,last*/, uint64_t* block_offset /*Ng page ids*/, uint64_t* blocks /*Nb page byte offsets, insertion order*/);

/* ---- particlesToGrid (MpmSimulationBase.cpp:461-533,611-656) + getNumNodes (MpmGrid.h:148-161) + buildMassMatrix (:817-826) */
int hot_p2g(hot_ctx*);
int hot_get_grid(hot_ctx*, int32_t* id2coord /*3Nn*/, void* mass /*Nn*/, void* v /*3Nn*/);

/* ---- collision nodes: the output of buildInitialDvAndVnForNewton's collision query
 *      (MpmSimulationBase.cpp:1139-1184, CollisionObject.h:16-45).  dv_collide = (v_collider - v_node) per
 *      collision node, NULL = static collider (-v_node).  Call after hot_p2g, before hot_begin_step. */
int hot_set_bc(hot_ctx*, int32_t Nc, const int32_t* node_id, const void* P /*9Nc*/, const void* R /*9Nc or NULL=I*/,
    const void* Rinv /*9Nc or NULL=I*/, const uint8_t* slip /*Nc or NULL=0*/, const void* dv_collide /*3Nc or NULL*/);
/* Built-in device-side collision query for static STICKY analytic half spaces {x : (x-o).n <= 0}
 * (AnalyticCollisionObject + HalfSpace, re-evaluated every hot_begin_step; replaces hot_set_bc). */
int hot_set_sticky_halfspaces(hot_ctx*, int32_t n, const double* origin /*3n*/, const double* normal /*3n*/);

/* ---- analytic collision objects evaluated per grid node on the device at hot_begin_step: the collision query of
 *      buildInitialDvAndVnForNewton (MpmSimulationBase.cpp:1139-1184) = AnalyticCollisionObject::multiObjectCollision
 *      (Lib/Ziran/Math/Geometry/CollisionObject.cpp:107-148) + detectAndResolveCollision (:384-447) over HalfSpace /
 *      Sphere / AxisAlignedAnalyticBox level sets (AnalyticLevelSet.cpp), RotationExtractor for slip nodes
 *      (MpmSimulationBase.h:271-281).  Objects carry the reference's full transform x = R s X + b with rates (omega, ds/dt, db/dt);
 *      the caller's updateState callback refreshes them with another hot_set_collision_objects call.  type uses the reference's
 *      enum values (CollisionObject.h:52-57).  Boxes (both kinds) and capped cylinders must be STICKY (their automatic-differentiation normal is not defined inside them).
 *      Replaces any half spaces / explicit collision nodes set before; n = 0 clears. */
enum hot_collision_type { HOT_COLLISION_STICKY = 1, HOT_COLLISION_SLIP = 2, HOT_COLLISION_SEPARATE = 3 };
enum hot_collision_shape { HOT_SHAPE_HALFSPACE = 0, HOT_SHAPE_SPHERE = 1, HOT_SHAPE_BOX = 2, HOT_SHAPE_CAPPED_CYLINDER = 3, HOT_SHAPE_TORUS = 4, HOT_SHAPE_ROTATED_BOX = 5,
    /* composite level sets (Lib/Ziran/Math/Geometry/AnalyticLevelSet.h:58-120, AnalyticLevelSet.cpp:148-236): the record is followed by its member records in
     * the array — primitives of which only shape / p0 / p1 / lsq are read; the composite's own transform, type and friction apply.  UNION: p1[0] = number of
     * members, signed distance = the smallest of the members', normal of the member that attains it first.  DIFFERENCE: exactly two members A, B (p1[0] = 2),
     * signed distance max(phi_A, -phi_B), normal -n_B where -phi_B > phi_A, else n_A.  Collision where the distance is <= 0 (AnalyticLevelSet::queryInside).
     * A composite with a box / capped-cylinder member must be STICKY, one with a half-space member cannot turn or scale (as for those shapes alone). */
    HOT_SHAPE_UNION = 6, HOT_SHAPE_DIFFERENCE = 7 };
typedef struct hot_collision_object {
    int32_t shape; /* hot_collision_shape */
    int32_t type; /* hot_collision_type */
    double p0[3]; /* half space: origin ; sphere: centre ; box: min corner ; capped cylinder / torus / rotated box: centre b of the level set */
    double p1[3]; /* half space: outward normal ; sphere: (radius, -, -) ; box: max corner ; capped cylinder: (radius, height, -) ;
                     torus: (r0, r1, -) ; rotated box (AnalyticBox): half edges */
    double friction;
    double b[3]; /* translation of the object */
    double dbdt[3]; /* its velocity */
    double R[9]; /* rotation matrix, column-major (Rotation<T,3>::rotation); identity when the object does not turn */
    double omega[3]; /* angular velocity (world frame) */
    double s; /* uniform scaling, > 0 (1 = none) */
    double dsdt; /* its rate */
    double lsq[4]; /* capped cylinder / torus / rotated box: the level set's own rotation, quaternion (w, x, y, z) as in their constructors
                      (AnalyticLevelSet.h:241-254, AnalyticLevelSet.cpp:568-576); the primitive's axis is y.  (1,0,0,0) = none */
} hot_collision_object; /* world x = R s X + b  (CollisionObject.h:63-69); p0 / p1 are given in material space X */
int hot_set_collision_objects(hot_ctx*, int32_t n, const hot_collision_object* objects);

/* ---- MultigridSimulation::startBackwardEuler (Projects/multigrid/MultigridSimulation.h:167-186):
 *      dv0 = g dt (collider dv on collision nodes), vn = v, Fn = F (backupStrain), resetLSFlag */
int hot_begin_step(hot_ctx*, double dt);
int hot_get_dv(hot_ctx*, void* dv /*3Nn*/);
int hot_set_dv(hot_ctx*, const void* dv /*3Nn*/);

/* ---- ImplicitSolverObjective (Projects/multigrid/ImplicitSolver.h) */
int hot_update_state(hot_ctx*, const void* dv /*3Nn or NULL = current*/, double* energy); /* updateState :237-252 + totalEnergy :254-275 */
int hot_get_particle_state(hot_ctx*, void* F /*9Np trial F*/, void* stress /*9Np  V_p P Fn^T*/, void* gradV /*9Np*/);
int hot_residual(hot_ctx*, void* r /*3Nn*/); /* computeResidual :128-155 at the current state */
int hot_project(hot_ctx*, void* v /*3Nn in/out*/); /* project lambda, MultigridSimulation.h:105-124 */
int hot_cn_tolerance(hot_ctx*, void* node_tol /*Nn*/); /* evaluatePerNodeCNTolerance :667-696 */
int hot_build_hessian(hot_ctx*); /* buildMatrix<true> :470-603 (+ buildDiagonal of level 0) */
int hot_matfree_multiply(hot_ctx*, const void* x, void* y); /* multiply :741-758 with matrix_free */

/* ---- the remaining members the reference's solver templates call on the objective (LBFGS.h:344,410-414, ExtendedNewtonsMethod.h:48-61):
 *      lineSearch :312-333 (energy backtracking from alpha; ddv comes back scaled by the accepted alpha and transformed, residual is the
 *      residual at the accepted point; as in the reference the nodal dv of the context moves to the accepted point and the objective's
 *      `updated` flag stays set for the rest of the step), shouldExitByCN :174-211 (or the l2 test without useCN), recoverSolution /
 *      transformResidual :106-125 (slip nodes to / from their normal frame), computeStep :355-432 (projected Newton, lsolver 1 / 2:
 *      rebuilds matrix + hierarchy, then MINRES / inexact PCG). */
int hot_line_search(hot_ctx*, void* ddv /*3Nn in/out*/, void* residual /*3Nn out*/, double alpha, double* alpha_out);
int hot_should_exit(hot_ctx*, const void* residual /*3Nn*/, int32_t* exit_now, double* scaled_residual /*NULL ok*/);
int hot_recover_solution(hot_ctx*, void* v /*3Nn in/out*/);
int hot_transform_residual(hot_ctx*, void* v /*3Nn in/out*/);
int hot_compute_step(hot_ctx*, const void* residual /*3Nn*/, void* step /*3Nn out*/);

/* ---- MultigridBuilder::build (Projects/multigrid/MultigridPreconditioner.h:554-703) */
int hot_build_mg(hot_ctx*);
int hot_get_level(hot_ctx*, int32_t level, int32_t* nrows, int32_t* colsize, int32_t* id2coord /*3*nrows or NULL*/);
/* padded-ELL dump of level's system matrix: entryCol[nrows*colsize], entryVal[nrows*colsize*9]
 * (row-major slots, 3x3 column-major) — SquareMatrix.h:27-34.  Slot order inside a coarse row is
 * implementation-defined (the reference's is std::unordered_map iteration order, SquareMatrix.h:560-564). */
int hot_get_matrix(hot_ctx*, int32_t level, int32_t* entryCol, void* entryVal);
/* structurally non-zero 3x3 blocks of the level's system matrix (used for the roofline's algorithmic bytes) */
int hot_get_level_nnzb(hot_ctx*, int32_t level, int64_t* nnzb);
/* of those, the off-diagonal blocks whose column lies in the row's own 4^3 colour block (the coloured-GS kernels split a row into its
 * off-block part, streamed one wavefront per row, and its in-block part, the block's triangular solve): the roofline's algorithmic bytes
 * of the two kernels.  Valid after hot_build_mg on levels that were coloured; HIP product only. */
int hot_get_level_inblock_nnzb(hot_ctx*, int32_t level, int64_t* nnzb);
int hot_get_prolongation(hot_ctx*, int32_t level, int32_t* entryCol /*8*nrows(level)*/, void* weight /*8*nrows(level)*/);
/* measurement aid, HIP product only: the rate (GB/s, bytes read + bytes written) of a device-to-device copy KERNEL (one non-temporal 16-byte piece per thread) over `bytes` bytes, `reps` launches between two events on the context's stream — the attainable streaming rate of the box the
 * roofline fractions are quoted beside ("a device-to-device copy kernel"); allocates and frees two buffers of `bytes` bytes */
int hot_copy_bandwidth(hot_ctx*, int64_t bytes, int32_t reps, double* gbytes_per_s);

/* ---- operators */
int hot_spmv(hot_ctx*, int32_t level, const void* x, void* y); /* SquareMatrix::multiply :477-487 */
int hot_restrict(hot_ctx*, int32_t level, const void* fine, void* coarse); /* SparseMPMMatrix::transposeMultiply */
int hot_prolong(hot_ctx*, int32_t level, const void* coarse, void* fine); /* SparseMPMMatrix::multiply on promats */
/* smoother plug point (MultigridPreconditioner.h:67-79): kind uses the -smoother numbering */
int hot_smooth(hot_ctx*, int32_t level, int32_t kind, int32_t iterations, double tolerance, void* u, void* r,
    const void* initial_residual /* cg_smooth's reference residual, NULL = r */);
int hot_vcycle(hot_ctx*, const void* in, void* out); /* MultigridOperator::operator() :362-421 */

/* ---- nonlinear solve: LBFGS::solve (Lib/Ziran/Math/Nonlinear/LBFGS.h:300-437) or
 *      ExtendedNewtonsMethod::solve (ExtendedNewtonsMethod.h:39-66) + computeStep (ImplicitSolver.h:355-432) */
int hot_solve(hot_ctx*, hot_stats* stats);

/* ---- restoreStrain + constructNewVelocityFromNewtonResult (MpmSimulationBase.cpp:891-901) + gridToParticles
 *      (:903-1042) + evolveStrain + applyPlasticity (:1044-1064).  flags: bit0 = some particle moved > dx
 *      (faster_than_grid_cell), bit1 = > cfl*dx/2 */
int hot_g2p(hot_ctx*, double dt, int32_t* flags);

/* ---- MultigridSimulation::advanceOneTimeStep (MultigridSimulation.h:235-297) with device-side BCs */
int hot_advance(hot_ctx*, double dt, hot_stats* stats);

/* ---- MpmSimulationBase::calculateDt (Lib/MPM/MpmSimulationBase.cpp:789-814) with evalMaxParticleSpeed (:1186-1218):
 *      dt = cfl * dx / max_p |v_p| (step.max_dt when nothing moves); the particle bounding box is returned as well.
 *      The speed bound includes the collision objects' own evalMaxSpeed over the particle bounding box (hot_set_collision_objects: moving and
 *      rotating primitives, members of unions / differences riding along with their composite; static objects contribute 0). */
int hot_calculate_dt(hot_ctx*, double max_dt, double* dt, double* max_speed, double* min_corner /*3 or NULL*/, double* max_corner /*3 or NULL*/);
/* ---- SimulationBase::advanceOneFrame (Lib/Ziran/Sim/SimulationBase.h:291-327) with TimeStepping::nextDt / advance
 *      (Lib/Ziran/Sim/TimeStepping.h:45-76): substeps of hot_advance with dt = nextDt(calculateDt()) until frame_dt is
 *      consumed.  stats = the last substep's; iterations_total sums the nonlinear iterations of all substeps. */
int hot_advance_frame(hot_ctx*, double frame_dt, double min_dt, double max_dt, int32_t* substeps, int32_t* iterations_total, hot_stats* stats);

/* ---- one connected body over several ranks (one context per rank = per GPU).  The reference is a single process (SURVEY.md
 *      §8e); this is the MI355X-native extension of the path.  Decomposition ("halo mode", the default):
 *        particles   every rank holds one of `size` runs of the global SPGrid page order, re-cut by particle-count-weighted splitters at
 *                    every hot_sort (particles migrate with their global ids); per-particle sums (energy, max speed) are scalar all-reduces;
 *        index       block list, node numbering and coordinates, colouring, coarse numbering, transfer tables and row ownership are
 *                    REPLICATED integers: every rank derives them with deterministic kernels from small all-gathered inputs once per step,
 *                    so every exchange list is computable locally and every numbering equals the single-rank one;
 *        node tiles  (P2G, force, CN tolerance, matrix-free product) are summed PAIRWISE between the ranks whose particle groups cover a
 *                    block, in ascending rank order (every sharer ends with the same bits; nobody else receives anything);
 *        rows        a 4^3 colour block — its Hessian / Galerkin rows, its Gauss-Seidel updates — is owned by the rank whose particles first
 *                    touch it; rows that also receive contributions of another rank's particles / fine rows are completed by a personalised
 *                    exchange of partial rows per build;
 *        vectors     DOF vectors are valid on the rows a rank owns plus the halo it reads (125-stencil, particle tiles, transfer windows),
 *                    refreshed by one halo gather per operator; vector algebra runs on owned rows, every batch of inner products is one
 *                    small all-reduce, whose identical results let all ranks take the same line-search / termination decisions;
 *        smoothing   colour-synchronous Gauss-Seidel (the reference's update order across ranks, MultigridPreconditioner.h:266-318,
 *                    sixteen halo exchanges per symmetric sweep) or, hot_config.shard_gs = 1, rank-local sweeps with one exchange;
 *        coarse      levels below `partition_min_rows` rows are replicated (one all-reduce of the level's matrix per build).
 *      hot_config.shard_replicated = 1 selects the first-generation decomposition instead (replicated DOF vectors, one all-reduce of the
 *      node tiles per scatter, one all-gather per operator).
 *      The library performs no communication itself: it calls the three collectives below at those points, with DEVICE
 *      pointers (a host-memory implementation of this ABI: host pointers), after synchronising its stream; the callee must have completed the
 *      operation when it returns.  hot_amd/dist.py implements them over torch.distributed (RCCL on GPUs, gloo in the CPU
 *      tests); a C++ host would pass ncclAllReduce / ncclAllGather / grouped ncclSend+ncclRecv on its own stream + sync.
 *      All ranks must make the same sequence of API calls.  Return 0 on success. */
enum hot_comm_dtype { HOT_COMM_F32 = 0, HOT_COMM_F64 = 1, HOT_COMM_I32 = 2, HOT_COMM_I64 = 3 };
enum hot_comm_op { HOT_COMM_SUM = 0, HOT_COMM_MAX = 1 };
typedef struct hot_comm {
    int32_t rank, size;
    void* user;
    /* in-place element-wise reduction over the ranks of n elements at buf; on_device: buf is device memory (else host) */
    int32_t (*allreduce)(void* user, void* buf, int64_t n, int32_t dtype, int32_t op, int32_t on_device);
    /* recv[r * bytes .. (r + 1) * bytes) = rank r's `bytes` bytes at send (the same count on every rank) */
    int32_t (*allgather)(void* user, const void* send, void* recv, int64_t bytes, int32_t on_device);
    /* personalised exchange: send_bytes[r] bytes at send + send_off[r] go to rank r, recv_bytes[r] bytes from rank r land
     * at recv + recv_off[r]; both sides know all counts (host arrays of `size` entries); a rank sends nothing to itself */
    int32_t (*alltoallv)(void* user, const void* send, const int64_t* send_off, const int64_t* send_bytes, void* recv, const int64_t* recv_off, const int64_t* recv_bytes,
        int32_t on_device);
    int32_t partition_min_rows; /* coarse levels with fewer rows are replicated instead of partitioned; 0 = default (4096: a replicated level costs one all-reduce of its whole matrix per build, 9 KB per row, and of a vector per restriction; a partitioned one only halo exchanges) */
    int32_t stream_ordered; /* 1: the callbacks enqueue device-payload collectives on the context's own HIP stream (hot_get_stream) and
                               return without waiting, so the library does not synchronise its stream around them; 0: host-synchronous */
    int32_t reserved[2];
} hot_comm;
/* Install (size > 1) or remove (NULL or size == 1) the communicator; call before hot_set_particles.  The shard given to
 * hot_set_particles must be a contiguous range of the global particle list in sort-key order of their SPGrid pages
 * (hot_amd/dist.py: shard_by_page_order does this split) for node numbering identical to the single-rank run. */
int hot_set_comm(hot_ctx*, const hot_comm* comm);
/* Global particle ids of a sharded run (default: 0..Np-1 in the order of hot_set_particles; pass the particles' indices in the whole body).
 * They break ties of the sort key inside a cell (the reference uses the particle index, MpmSimulationBase.cpp:1084) and travel with a particle
 * when hot_sort hands it to another rank: at every hot_sort of a sharded context the particles are redistributed so that rank r holds
 * the r-th of `size` nearly equal runs of the SPGrid page order (the shard hot_set_particles was given is only the starting point), so
 * Np changes (hot_get_counts) and hot_get_particles / hot_get_particle_ids return the rank's current particles in ascending id order. */
int hot_set_particle_ids(hot_ctx*, const int32_t* ids /*Np*/);
int hot_get_particle_ids(hot_ctx*, int32_t* ids /*Np*/);
int hot_get_stream(hot_ctx*, void** hip_stream); /* the context's hipStream_t, for stream-ordered communicators */
/* Native RCCL communicator (hot_amd/csrc/rccl_comm.hip): ncclAllReduce / ncclAllGather / grouped ncclSend + ncclRecv on the
 * context's stream, stream-ordered.  Rank 0 calls hot_rccl_unique_id, the host hands the 128 bytes to every rank, each rank
 * calls hot_rccl_attach before hot_set_particles.  Both fail with HOT_ERR_DEVICE when RCCL cannot be loaded. */
int hot_rccl_unique_id(void* out128);
int hot_rccl_attach(hot_ctx*, const void* unique_id128, int32_t rank, int32_t size, int32_t partition_min_rows);
int hot_rccl_selftest(hot_ctx*); /* runs every collective of the attached communicator once on known data (any number of ranks) */

/* ---- the constitutive model and the plastic return mappings for caller-supplied deformation gradients (arrays of `real`,
 *      3x3 column-major, per-sample mu / lambda): CorotatedIsotropic<T,3>::updateScratch + psi + firstPiola +
 *      firstPiolaDerivative (Lib/Ziran/Physics/ConstitutiveModel/CorotatedIsotropic.h:110-230; dPdF is the 9x9 derivative, column-major
 *      over the column-major vectorisations of P and F, PSD-projected when project is non-zero; project == 2 exactly: psi alone, evaluated as the line
 *      search's energy-only trials evaluate it — from the invariants of F^T F without an SVD where det F > 0.1 —, P and dPdF untouched), and
 *      VonMisesFixedCorotated / SnowPlasticity::projectStrain (Lib/Ziran/Physics/PlasticityApplier.cpp:96-131, :18-50) with
 *      cfg.yield_stress / cfg.snow, in place.  NULL outputs are skipped. */
int hot_constitutive_eval(hot_ctx*, int32_t n, const void* F /*9n*/, const void* mu /*n*/, const void* lambda /*n*/, int32_t project,
    void* psi /*n*/, void* P /*9n*/, void* dPdF /*81n*/);
int hot_plasticity_eval(hot_ctx*, int32_t kind /*1 von Mises, 2 snow*/, int32_t n, void* F /*9n in/out*/, void* mu /*n in/out*/, void* lambda /*n in/out*/,
    void* Jp /*n in/out, snow only*/);

/* ---- frame output (SimulationBase::write -> MpmSimulationBase::writeState, Lib/Ziran/Sim/SimulationBase.h:152-190,
 *      Lib/MPM/MpmSimulationBase.cpp:754-785): hot_write_partio = writePartio's .bgeo of the particle positions (PartioIO.h:142-180);
 *      hot_write_restart / hot_read_restart = the particle DataManager in the container layout of DataManager::writeData
 *      (DataManager.h:263-294) with this library's columns (see hot_amd/csrc/io.hip for what is and is not interchangeable with the
 *      reference's restart_<frame>.dat).  Particles in the caller's order; hot_read_restart replaces the particle set. */
int hot_write_partio(hot_ctx*, const char* path);
int hot_write_restart(hot_ctx*, const char* path);
int hot_read_restart(hot_ctx*, const char* path);

/* ---- per-kernel timings gathered with HIP events on the launch stream when cfg.profile = 1 */
int hot_profile_reset(hot_ctx*);
int hot_profile_count(hot_ctx*, int32_t* n);
int hot_profile_get(hot_ctx*, int32_t i, char* name /*>=64 bytes*/, int64_t* calls, double* total_ms);

const char* hot_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HOT_MI355X_H */
