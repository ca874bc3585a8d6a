/*
 * mgsolve.h -- host-side C API of the MI355X multigrid V-cycle (the product's own driver).
 *
 * Mirrors the call sequence of the reference driver for `-cycle 0`
 * (src/poisson.c:27-138: SetUpMesh -> SetUpIndices/mapping -> SetUpOperator -> SetUpSolver ->
 *  Assemble -> Solve -> Postprocessing) without the O(N) host index maps and the ~5N scalar
 * MatSetValue calls that make the reference driver infeasible beyond ~4097^2:
 * operators are matrix-free constant stencils, maps are the implicit lexicographic formula.
 * Host code is C99; every device operation goes through the kernel ABI of include/mgk.h.
 *
 * The same objects serve 2-D (the reference's DIMENSION 2) and the 3-D extension.
 * There is no CPU fallback: creating a solver without a HIP device fails with MGK_ENOGPU.
 */
#ifndef MGSOLVE_H
#define MGSOLVE_H
#include "mgk.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MG_KSP_RICHARDSON = 0, MG_KSP_CHEBYSHEV = 1 } mg_ksp_type;
typedef enum { MG_PREC_FP64 = 0, MG_PREC_MIXED = 1 } mg_precision;

/* options of the reference driver (src/poisson.c:51-59, poisson.in) + the PETSc options that
 * KSPSetFromOptions (src/solver.c:1476,1492,1509) would pick up for the smoother */
typedef struct mg_config {
    int dim;            /* 2 (reference) or 3 (extension) */
    int npts;           /* -npts : points per side INCLUDING the boundary; unknowns per side = npts-2 */
    int levels;         /* -levels (== -grids: one grid per level, the V-cycle case) */
    int v[2];           /* -v v0,v1 : sweeps on levels 0..L-2 / on the coarsest level */
    int maxiter;        /* -iter */
    int ksp_type;       /* -ksp_type richardson|chebyshev */
    double scale;       /* -ksp_richardson_scale (PETSc default 1.0) */
    double emin, emax;  /* -ksp_chebyshev_eigenvalues emin,emax */
    double rtol;        /* stopping factor of src/solver.c:1530; <=0 selects the reference's 1.e-7 */
    int device;         /* HIP device ordinal */
    int precision;      /* mg_precision */
    int rank, nranks;   /* z-slab decomposition over `nranks` GPUs (1: whole grid) */
    int dist_min_n;     /* levels with n >= dist_min_n stay distributed, coarser ones are replicated; <=0: default 255
                         * (below that a slab sweep is shorter than the latency of its halo exchange) */
    int fuse;           /* bit 0: final residual fused with its norm (no rv write); bit 1: prolongation fused into the
                         * first post-smoothing sweep; bit 2: pre-restriction residual fused with the restriction;
                         * bit 3: the residual norm that closes a cycle is evaluated by the kernel that also makes the first
                         * pre-smoothing sweep of the next cycle (adopted only if a next cycle runs); bit 4 (mixed precision): the
                         * fp64 correction u += e and the fp64 residual -> fp32 in one pass; bit 5: pairs of sweeps in one pass
                         * (temporal blocking, levels >= pair_min_n, both precisions); bit 7 (testing): bit 2 also below 255^3, where two short
                         * kernels are quicker; bit 8: the fused residual+restriction also writes the coarse level's first (zero-guess)
                         * sweep; bit 9: the levels that fit in LDS (n <= 15 in 3-D, <= 63 in 2-D) run as ONE kernel per cycle (mgk_tail_cycle_*);
                         * bit 10 (fp64, 3-D, full-row shapes n = 127 .. 1023, whole grids and z-slabs): the pass of bit 3 makes the first TWO
                         * sweeps of the next cycle (mgk_jacobi2_sumsq_f64 / _slab_f64) and the last pre-smoothing sweep runs inside the
                         * restriction's pass (mgk_sweep_residual_restrict_f64 / _slab_f64): the fine level moves 99 instead of 115 B per
                         * unknown and cycle;
                         * bit 11 (3-D whole levels that sweep in pairs: fp32 up to 1023^3, fp64 up to 511^3): a pre-smoothing of >= 3 sweeps
                         * from the zero guess starts with ONE pass that makes three of them and reads b alone (mgk_jacobi2_zero_*);
                         * bit 12 (fp64, 3-D, level 0 of 511- / 1023-wide whole grids, v0 = 3): post-smoothing is ONE pass for the prolongation and two
                         * sweeps (mgk_prolong_jacobi2_f64); the third sweep is the first stage of the two-sweep pass that evaluates the
                         * norm (mgk_jacobi2_sumsq_mid_f64): 91 B per fine unknown and cycle.  The iterate the norm belongs to is not
                         * stored; when the iteration stops one more sweep materialises it;
                         * bit 13 (2-D, fp64, Richardson, uniform and stretched meshes): THREE sweeps per pass (mgk_jacobi3_2d_*): pre-smoothing
                         * from the zero guess in one pass over b, post-smoothing in one pass with the prolongation, and the norm pass of
                         * bit 3 makes all three pre-smoothing sweeps of the next cycle: three passes over a level per V(3,3) cycle;
                         * bit 14 (round 3; nranks > 1): bit 12's three passes on z-slabs -- mgk_prolong_jacobi2_slab_f64 (the neighbours' boundary and
                         * second planes of u and of the coarse u arrive in two grouped exchanges hidden behind the interior planes),
                         * mgk_jacobi2_sumsq_mid_slab_f64, the plain fused residual + restriction: every rank moves 91 instead of 99 B per fine
                         * unknown and cycle;
                         * default (-1): bits 0-5 and 8-14 on */
    int overlap;        /* nranks > 1: halo of sweep k on the comm stream while sweep k's interior runs; default on (-1) */
    int graph;          /* replay the launch-bound coarse levels as one captured HIP graph; default on (-1) */
    int pair_min_n;     /* levels with n >= pair_min_n run their sweeps two per pass (fuse bit 5); <=0: default 255 (3-D), 2047 (2-D) */
    int slab_chunk;     /* nranks > 1: planes per workgroup of the marching kernels (short blocks let the exchange kernels in beside the
                         * interior launches); <0: default = a quarter of the rank's fine planes, at least 32 (MG_SLAB_CHUNK overrides); 0: the long
                         * streams of a single GPU */
    int mesh;           /* -mesh: 0 uniform; 1 / 2: the reference's meshes stretched in y (src/mesh.c:45-107,165-169), 2-D, one GPU,
                         * Richardson + Jacobi: the operator rows then depend on the grid row (per-row coefficient tables); the same
                         * fused cycle on the row-table forms of its kernels (mgk_*_rowcoef_f64) */
} mg_config;

void mg_config_default(mg_config *cfg);     /* poisson.in defaults + -pc_type jacobi -ksp_richardson_scale 1 */

typedef struct mg_solver mg_solver;

/* communication hooks for nranks > 1 (see mg_comm.h); a solver with nranks == 1 needs none */
struct mg_comm;

int  mg_solver_create(mg_solver **s, const mg_config *cfg, struct mg_comm *comm);
void mg_solver_destroy(mg_solver *s);
const char *mg_last_error(void);

/* levelvecb (src/solver.c:558-620) with Ffunc (src/problem.c:24-28) on the uniform mesh (src/mesh.c:130-195) */
int  mg_solver_set_rhs_problem(mg_solver *s);
/* arbitrary right-hand side: compact lexicographic array of the LOCAL slab (nz_local*ny*nx doubles) */
int  mg_solver_set_rhs_host(mg_solver *s, const double *b_compact);
/* u0 = 0, iteration counter and KSP guess flags back to their initial state (src/solver.c:1512-1523) */
int  mg_solver_reset(mg_solver *s);
/* Solve(): MultigridVcycle (src/solver.c:1414-1575) until ||r|| <= rtol*||b||, divergence, or maxiter */
int  mg_solver_solve(mg_solver *s);
/* exactly `ncycles` more V-cycles from the current state (no stopping test); used by bench.py */
int  mg_solver_cycles(mg_solver *s, int ncycles);
/* block until every stream of this solver's device is idle */
int  mg_solver_sync(mg_solver *s);

int    mg_solver_iterations(const mg_solver *s);          /* solver->numIter (src/solver.c:1558) */
double mg_solver_bnorm(const mg_solver *s);
/* absolute residual norms rnorm[0..iterations] (src/solver.c:1520,1549) BEFORE the division by rnorm[0] */
const double *mg_solver_rnorm(const mg_solver *s);
double mg_solver_solve_seconds(const mg_solver *s);       /* "Solver walltime" window (src/solver.c:1526,1553) */

int  mg_solver_num_levels(const mg_solver *s);
int  mg_solver_level_n(const mg_solver *s, int level);              /* unknowns per side */
int  mg_solver_level_local_planes(const mg_solver *s, int level, int *z0);
long mg_solver_local_unknowns(const mg_solver *s);
/* smoother point updates of one V-cycle summed over levels (all ranks): sum_l sweeps_l * N_l */
double mg_solver_dof_updates_per_cycle(const mg_solver *s);

/* fine-level solution of the local slab, compact lexicographic (GetSol, src/solver.c:1239-1315) */
int  mg_solver_get_solution(mg_solver *s, double *u_compact);
/* GetError (src/solver.c:1211-1237): {max|e|, sum|e|, sqrt(sum e^2)} against sin(pi x)sin(pi y)[sin(pi z)] */
int  mg_solver_error_norms(mg_solver *s, double err[3]);

/* per-kernel event timing of the fine-level smoother sweeps (bench.py roofline leg) */
int  mg_solver_profile(mg_solver *s, int enable);
int  mg_solver_profile_read(mg_solver *s, double *total_ms, int *launches);
/* the same for one kind of launch: 0 = plain fine-level sweeps, 1 = two-sweeps-in-one-pass launches */
int  mg_solver_profile_read_kind(mg_solver *s, int kind, double *total_ms, int *launches);

/* integer half of the reference (src/matbuild.c), implicit form */
void mg_get_ranges(int totaln, int procs, int *ranges);             /* matbuild.c:120-144 */
int  mg_grid_n(int npts, int grid);                                 /* matbuild.c:62-66 */
/* one-grid-per-level maps: grid (i,j[,k]) <-> global lexicographic index (matbuild.c:280-309) */
long mg_grid_to_global(int dim, int n, int k, int i, int j);
void mg_global_to_grid(int dim, int n, long idx, int *k, int *i, int *j);
/* plane-aligned slab split used by the multi-GPU path: [z0,z1) of level `level` for `rank` */
int  mg_slab_range(int npts, int levels_dist, int level, int rank, int nranks, int *z0, int *z1);

#ifdef __cplusplus
}
#endif
#endif
