/*
 * mg_solver.c -- C99 host side of the MI355X multigrid V-cycle: level hierarchy, matrix-free
 * operators, and the cycle driver.  Mirrors the reference's driver for `-cycle 0`:
 *
 *   reference (paths relative to /root/reference)              here
 *   SetUpMesh/Coords          src/mesh.c:130-249                coords_uniform()
 *   SetUpIndices/mapping      src/matbuild.c:85-323             mg_grid_n(), implicit lexicographic maps
 *   OpA + fillJacobians       src/problem.c:3-22, solver.c:185  level_stencil(): 5/7 constants per level
 *   Res / Pro                 src/solver.c:1035-1154            matrix-free kernels (mgk_restrict/prolong)
 *   levelvecb                 src/solver.c:558-620              mg_solver_set_rhs_problem()
 *   MultigridVcycle           src/solver.c:1414-1575            vcycle_once(), mg_solver_solve()
 *   GetError                  src/solver.c:1211-1237            mg_solver_error_norms()
 *
 * All device work goes through include/mgk.h; this file contains no HIP.
 */
#include "mgsolve.h"
#include "mg_comm.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MG_PI 3.14159265358979323846   /* include/problem.h:13 */
#define MG_MAX_LEVELS 32
#define MG_MAX_TIMERS 4096

static __thread char g_mgerr[512] = "ok";
const char *mg_last_error(void) { return g_mgerr; }
static int mgfail(int code, const char *what) {
    snprintf(g_mgerr, sizeof(g_mgerr), "%s (code %d; kernel layer: %s)", what, code, mgk_last_error());
    return code;
}
#define CHK(call) do { int rc_ = (call); if (rc_) return mgfail(rc_, #call); } while (0)

/* the fields of one level in one precision (index 0: fp64, 1: fp32) */
typedef struct mg_fset {
    mgk_geom g;             /* local geometry in elements of that precision */
    void *u, *b, *rv, *tmp;
    int guess_nonzero;      /* KSPSetInitialGuessNonzero state of ksp[l] (src/solver.c:1532,1537,1543) */
    int u_ghost_ok;         /* z ghost planes of `u` hold the neighbours' current boundary planes */
    int u_ghost_pending;    /* ... but the exchange is still in flight on the comm stream */
    int jz_ready;           /* tmp already holds the first sweep from a zero guess (written by the fused residual+restriction) */
    int last_sweep_pending; /* pre-smoothing stopped one sweep short: the restriction that follows makes it (mgk_sweep_residual_restrict_f64) */
    int b_ghost_ok;         /* z ghost planes of `b` hold the neighbours' boundary planes (two-sweep passes on slabs) */
    void *far;              /* distributed levels: field of geometry gfar = (nx, ny, 2) for the neighbours' SECOND planes of u */
    void *far2, *bfar;      /* fp64, fuse bit 10: same geometry; hi ghost = the rank above's THIRD plane of u / SECOND plane of b (sweep fused
                             * with residual + restriction on a slab: mgk_sweep_residual_restrict_slab_f64) */
    int bfar_ok;            /* bfar's hi ghost plane is valid (b of a level changes only when the restriction above rewrites it) */
    mgk_geom gfar;
} mg_fset;

typedef struct mg_level {
    int n;                  /* unknowns per side of the whole grid */
    int z0, nzl;            /* owned planes [z0, z0+nzl) (3-D); whole grid when replicated / 2-D */
    int nz_min;             /* fewest planes any rank owns on this level: every choice between code paths that differ in their
                             * exchanges is made on it, never on the own slab size, so that all ranks take the same path */
    int distributed;
    double coef[7], dinv, h;
    double *ctab, *dtab;    /* -mesh 1/2 (2-D): device tables, 5 coefficients {(i-1), W, C, E, (i+1)} and 1/diag per grid row */
    mg_fset f[2];
    double *p2;             /* Chebyshev: third recurrence vector (fp64) */
} mg_level;

/* the kernel ABI of one precision behind untyped pointers: the cycle code below is written once */
typedef struct mg_ops {
    int esz;
    int (*jacobi_range)(mgk_ctx *, const mgk_geom *, const double *, double, double, const void *, const void *, void *, int, int, void *);
    int (*jacobi_zero)(mgk_ctx *, const mgk_geom *, double, double, const void *, void *, void *);
    int (*residual)(mgk_ctx *, const mgk_geom *, const double *, const void *, const void *, void *, void *);
    int (*restrict_fw)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const void *, void *, void *);
    int (*prolong_add)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const void *, void *, void *);
    int (*prolong_jacobi)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const double *, double, double, const void *, const void *, const void *, void *, void *);
    int (*residual_restrict)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const double *, const void *, const void *, void *, void *);   /* NULL: not built */
    int (*residual_range)(mgk_ctx *, const mgk_geom *, const double *, const void *, const void *, void *, int, int, void *);
    int (*restrict_finish)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const void *, void *, void *);
    int (*residual_restrict_jz)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const double *, const void *, const void *, void *, void *,
                                double, double, void *);
    int (*jacobi2)(mgk_ctx *, const mgk_geom *, const double *, double, double, const void *, const void *, void *, void *);
    int (*jacobi2_slab)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const double *, double, double, const void *, const void *, void *,
                        const void *, int, int, int, int, void *);
    int (*prolong_jacobi_range)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const double *, double, double, const void *, const void *, const void *,
                                void *, int, int, void *);
    int (*residual_restrict_range)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const double *, const void *, const void *, void *, int, int, void *);
    int (*residual_restrict_slab)(mgk_ctx *, const mgk_geom *, const mgk_geom *, const mgk_geom *, const double *, const void *, const void *,
                                  const void *, int, void *, int, int, void *);
    int (*tail_cycle)(mgk_ctx *, const mgk_geom *, int, const int *, const double *, const double *, double, int, int, const void *, void *, void *);
} mg_ops;

#define W64(name) static int name##_64
#define W32(name) static int name##_32
W64(jr)(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double sc, const void *b, const void *u, void *o, int z0, int z1, void *st) { return mgk_jacobi_range_f64(c, g, k, d, sc, (const double *)b, (const double *)u, (double *)o, z0, z1, st); }
W32(jr)(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double sc, const void *b, const void *u, void *o, int z0, int z1, void *st) { return mgk_jacobi_range_f32(c, g, k, d, sc, (const float *)b, (const float *)u, (float *)o, z0, z1, st); }
W64(jz)(mgk_ctx *c, const mgk_geom *g, double d, double sc, const void *b, void *o, void *st) { return mgk_jacobi_zero_f64(c, g, d, sc, (const double *)b, (double *)o, st); }
W32(jz)(mgk_ctx *c, const mgk_geom *g, double d, double sc, const void *b, void *o, void *st) { return mgk_jacobi_zero_f32(c, g, d, sc, (const float *)b, (float *)o, st); }
W64(rs)(mgk_ctx *c, const mgk_geom *g, const double *k, const void *b, const void *u, void *r, void *st) { return mgk_residual_f64(c, g, k, (const double *)b, (const double *)u, (double *)r, st); }
W32(rs)(mgk_ctx *c, const mgk_geom *g, const double *k, const void *b, const void *u, void *r, void *st) { return mgk_residual_f32(c, g, k, (const float *)b, (const float *)u, (float *)r, st); }
W64(rf)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const void *r, void *b, void *st) { return mgk_restrict_fw_f64(c, gf, gc, (const double *)r, (double *)b, st); }
W32(rf)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const void *r, void *b, void *st) { return mgk_restrict_fw_f32(c, gf, gc, (const float *)r, (float *)b, st); }
W64(pa)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const void *uc, void *uf, void *st) { return mgk_prolong_add_f64(c, gf, gc, (const double *)uc, (double *)uf, st); }
W32(pa)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const void *uc, void *uf, void *st) { return mgk_prolong_add_f32(c, gf, gc, (const float *)uc, (float *)uf, st); }
W64(pj)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, double d, double sc, const void *b, const void *uc, const void *u, void *o, void *st) { return mgk_prolong_jacobi_f64(c, gf, gc, k, d, sc, (const double *)b, (const double *)uc, (const double *)u, (double *)o, st); }
W32(pj)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, double d, double sc, const void *b, const void *uc, const void *u, void *o, void *st) { return mgk_prolong_jacobi_f32(c, gf, gc, k, d, sc, (const float *)b, (const float *)uc, (const float *)u, (float *)o, st); }
W64(rr)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const void *b, const void *u, void *bc, void *st) { return mgk_residual_restrict_f64(c, gf, gc, k, (const double *)b, (const double *)u, (double *)bc, st); }
W32(rr)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const void *b, const void *u, void *bc, void *st) { return mgk_residual_restrict_f32(c, gf, gc, k, (const float *)b, (const float *)u, (float *)bc, st); }
W64(rg)(mgk_ctx *c, const mgk_geom *g, const double *k, const void *b, const void *u, void *r, int z0, int z1, void *st) { return mgk_residual_range_f64(c, g, k, (const double *)b, (const double *)u, (double *)r, z0, z1, st); }
W32(rg)(mgk_ctx *c, const mgk_geom *g, const double *k, const void *b, const void *u, void *r, int z0, int z1, void *st) { return mgk_residual_range_f32(c, g, k, (const float *)b, (const float *)u, (float *)r, z0, z1, st); }
W64(fin)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const void *r, void *bc, void *st) { return mgk_restrict_finish_f64(c, gf, gc, (const double *)r, (double *)bc, st); }
W32(fin)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const void *r, void *bc, void *st) { return mgk_restrict_finish_f32(c, gf, gc, (const float *)r, (float *)bc, st); }
W64(j2)(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double sc, const void *b, const void *u, void *o, void *st) { return mgk_jacobi2_f64(c, g, k, d, sc, (const double *)b, (const double *)u, (double *)o, st); }
W32(j2)(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double sc, const void *b, const void *u, void *o, void *st) { return mgk_jacobi2_f32(c, g, k, d, sc, (const float *)b, (const float *)u, (float *)o, st); }
W64(j2s)(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gf, const double *k, double d, double sc, const void *b, const void *u, void *o, const void *far, int lo, int hi, int z0, int z1, void *st) { return mgk_jacobi2_slab_f64(c, g, gf, k, d, sc, (const double *)b, (const double *)u, (double *)o, (const double *)far, lo, hi, z0, z1, st); }
W32(j2s)(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gf, const double *k, double d, double sc, const void *b, const void *u, void *o, const void *far, int lo, int hi, int z0, int z1, void *st) { return mgk_jacobi2_slab_f32(c, g, gf, k, d, sc, (const float *)b, (const float *)u, (float *)o, (const float *)far, lo, hi, z0, z1, st); }
W64(rrz)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const void *b, const void *u, void *bc, void *uc0, double d, double sc, void *st) { return mgk_residual_restrict_jz_f64(c, gf, gc, k, (const double *)b, (const double *)u, (double *)bc, (double *)uc0, d, sc, st); }
W32(rrz)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const void *b, const void *u, void *bc, void *uc0, double d, double sc, void *st) { return mgk_residual_restrict_jz_f32(c, gf, gc, k, (const float *)b, (const float *)u, (float *)bc, (float *)uc0, d, sc, st); }
W64(pjr)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, double d, double sc, const void *b, const void *uc, const void *u, void *o, int z0, int z1, void *st) { return mgk_prolong_jacobi_range_f64(c, gf, gc, k, d, sc, (const double *)b, (const double *)uc, (const double *)u, (double *)o, z0, z1, st); }
W32(pjr)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, double d, double sc, const void *b, const void *uc, const void *u, void *o, int z0, int z1, void *st) { return mgk_prolong_jacobi_range_f32(c, gf, gc, k, d, sc, (const float *)b, (const float *)uc, (const float *)u, (float *)o, z0, z1, st); }
W64(rrr)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const void *b, const void *u, void *bc, int k0, int k1, void *st) { return mgk_residual_restrict_range_f64(c, gf, gc, k, (const double *)b, (const double *)u, (double *)bc, k0, k1, st); }
W32(rrr)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const void *b, const void *u, void *bc, int k0, int k1, void *st) { return mgk_residual_restrict_range_f32(c, gf, gc, k, (const float *)b, (const float *)u, (float *)bc, k0, k1, st); }
W64(rrs)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *k, const void *b, const void *u, const void *far, int hi, void *bc, int k0, int k1, void *st) { return mgk_residual_restrict_slab_f64(c, gf, gc, gfar, k, (const double *)b, (const double *)u, (const double *)far, hi, (double *)bc, k0, k1, st); }
W32(rrs)(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *k, const void *b, const void *u, const void *far, int hi, void *bc, int k0, int k1, void *st) { return mgk_residual_restrict_slab_f32(c, gf, gc, gfar, k, (const float *)b, (const float *)u, (const float *)far, hi, (float *)bc, k0, k1, st); }
W64(tc)(mgk_ctx *c, const mgk_geom *g0, int nl, const int *n, const double *k7, const double *di, double sc, int v0, int v1, const void *b, void *u, void *st) { return mgk_tail_cycle_f64(c, g0, nl, n, k7, di, sc, v0, v1, (const double *)b, (double *)u, st); }
W32(tc)(mgk_ctx *c, const mgk_geom *g0, int nl, const int *n, const double *k7, const double *di, double sc, int v0, int v1, const void *b, void *u, void *st) { return mgk_tail_cycle_f32(c, g0, nl, n, k7, di, sc, v0, v1, (const float *)b, (float *)u, st); }
static const mg_ops OPS[2] = {
    {8, jr_64, jz_64, rs_64, rf_64, pa_64, pj_64, rr_64, rg_64, fin_64, rrz_64, j2_64, j2s_64, pjr_64, rrr_64, rrs_64, tc_64},
    {4, jr_32, jz_32, rs_32, rf_32, pa_32, pj_32, rr_32, rg_32, fin_32, rrz_32, j2_32, j2s_32, pjr_32, rrr_32, rrs_32, tc_32},
};

struct mg_solver {
    mg_config cfg;
    mgk_ctx *ctx;
    mg_comm *comm;
    int levels, ldist;      /* ldist: number of distributed (finest) levels; 0 when nranks == 1 */
    mg_level L[MG_MAX_LEVELS];
    int *zstart;            /* plane starts of the first replicated level's producers (nranks+1) */
    double *rnorm;          /* maxiter+1 */
    int rnorm_cap;
    int iter;
    double bnorm, rchk;
    int started;
    int deferring;          /* mg_solver_cycles: norms are deposited on the device and read once at the end */
    double *d_norms; int d_norms_cap;
    double *pin; int pin_cap; /* pinned host landing area of the reduced norms */
    int spec_valid;         /* > 0: level-0 tmp holds that many sweeps of u, made by the sweep(s)+norm kernel that closed the last cycle */
    int sweep_owed;         /* fuse bit 12: the post-smoothing of level 0 stopped one sweep short (prolongation + two sweeps in one pass); the
                             * pass that evaluates the norm makes that sweep first */
    int last_cycle;         /* the caller knows (fixed cycle count) or expects (contraction so far) that this cycle is the last one: no sweep is
                             * owed and no speculative sweep is made -- the norm comes from the store-free residual + norm pass */
    int iterate_behind;     /* ... and after that pass u is still ONE sweep behind the iterate the norm belongs to (it was never stored:
                             * tmp holds the sweep after it); finalize_iterate() makes the sweep if the iteration stops here */
    double solve_seconds;
    int lgraph;             /* levels >= lgraph form the launch-bound coarse part replayed as one HIP graph (0: off) */
    int ltail;              /* levels >= ltail (n <= 15 in 3-D, <= 63 in 2-D) run as ONE kernel with their fields in LDS (0: off) */
    void *coarse_graph[2];  /* one recording per precision */
    void *graph_u[2], *graph_tmp[2];   /* u / tmp of the level that feeds the recording, as the recorded kernels know them */
    int graph_rerecorded;   /* recordings thrown away because those pointers had changed (0 in every default configuration) */
    /* profiling */
    int prof_on, prof_n;
    void *timers[MG_MAX_TIMERS];
    unsigned char timer_kind[MG_MAX_TIMERS];
    int prof_kind;          /* kind of the next timer: 0 plain sweep, 1 two sweeps in one pass */
    int ntimers_created;
};

/* ------------------------------------------------------------------ */
/* integer half                                                        */
/* ------------------------------------------------------------------ */
void mg_get_ranges(int totaln, int procs, int *ranges) {
    /* src/matbuild.c:120-144 */
    int q = totaln / procs, rem = totaln % procs;
    ranges[0] = 0;
    for (int p = 0; p < procs; p++) ranges[p + 1] = ranges[p] + q + (p < rem ? 1 : 0);
}

int mg_grid_n(int npts, int grid) {
    /* src/matbuild.c:62-66: n = (npts-1)/factor^g - 1 with factor 2 (src/poisson.c:91) */
    int f = 1;
    for (int q = 0; q < grid; q++) f *= 2;
    return (npts - 1) / f - 1;
}

long mg_grid_to_global(int dim, int n, int k, int i, int j) {
    /* src/matbuild.c:292-300 with one grid per level: count runs over i (rows) then j */
    return dim == 3 ? ((long)k * n + i) * n + j : (long)i * n + j;
}
void mg_global_to_grid(int dim, int n, long idx, int *k, int *i, int *j) {
    *j = (int)(idx % n);
    *i = (int)((idx / n) % n);
    *k = dim == 3 ? (int)(idx / ((long)n * n)) : 0;
}

/* Plane-aligned, nested slab split (multi-GPU; DESIGN.md "decomposition").  The planes of the
 * first NON-distributed level (index levels_dist) are cut like GetRanges (src/matbuild.c:120-144);
 * every finer level doubles the bounds and the last rank takes the one extra plane
 * (n_f = 2 n_c + 1), so slab starts are even and coarse plane c of a rank is centred on its own
 * fine plane 2c+1. */
int mg_slab_range(int npts, int levels_dist, int level, int rank, int nranks, int *z0, int *z1) {
    if (level < 0 || level > levels_dist || rank < 0 || rank >= nranks) return MGK_EINVAL;
    int nc = mg_grid_n(npts, levels_dist);
    if (nc < nranks) return MGK_EINVAL;
    int q = nc / nranks, rem = nc % nranks;
    int a = rank * q + (rank < rem ? rank : rem);
    int b = a + q + (rank < rem ? 1 : 0);
    for (int l = levels_dist - 1; l >= level; l--) {
        a = 2 * a;
        b = (rank == nranks - 1) ? mg_grid_n(npts, l) : 2 * b;
    }
    *z0 = a; *z1 = b;
    return 0;
}

/* ------------------------------------------------------------------ */
/* mesh / problem                                                      */
/* ------------------------------------------------------------------ */
static void coords_uniform(int npts, double *c) {
    /* src/mesh.c:140-171: end points set, interior by repeated addition of the spacing */
    c[0] = 0.0;
    c[npts - 1] = 1.0;
    double d = (c[npts - 1] - c[0]) / (npts - 1);
    for (int j = 1; j < npts - 1; j++) c[j] = c[j - 1] + d;
}

/* y coordinates of the stretched meshes (x stays uniform): src/mesh.c:154-176 */
static void coords_mesh_y(int npts, int mesh, double *c) {
    if (mesh == 0) { coords_uniform(npts, c); return; }
    c[0] = 0.0; c[npts - 1] = 1.0;
    const double length = c[npts - 1] - c[0];
    for (int j = 1; j < npts - 1; j++) {
        if (mesh == 1) c[j] = 1.0 - length * (cos(MG_PI * 0.5 * (j / (double)(npts - 1))));          /* :165-166 */
        else { double eta = (j / (double)(npts - 1)); c[j] = 0.0 + length * ((exp(2 * eta) - 1) / (exp(2) - 1)); }   /* :168-169 */
    }
}
/* metric coefficients at height y on [0,1]^2: MetricsNonUniform1 / 2, src/mesh.c:45-75 / 77-107 */
static void metrics_mesh(int mesh, double y, double *m) {
    const double b0 = 0.0, b1 = 1.0, b2 = 0.0, b3 = 1.0;
    if (mesh == 1) {
        double temp = ((b3 - b2) * (b3 - b2) - (b3 - y) * (b3 - y));
        m[0] = 1.0;
        m[1] = 4.0 / (MG_PI * MG_PI * temp);
        m[2] = 0.0;
        m[3] = (-2.0 * (b3 - y)) / (MG_PI * sqrt(temp * temp * temp));
        m[4] = 0.0;
        return;
    }
    double temp = ((exp(2) - 1) * (exp(2) - 1)) / (((y - b2) * (exp(2) - 1) + (b3 - b2)) * ((y - b2) * (exp(2) - 1) + (b3 - b2)));
    m[0] = 1.0 / ((b1 - b0) * (b1 - b0));
    m[1] = 0.25 * temp;
    m[2] = 0.0;
    m[3] = (-0.5) * temp;
    m[4] = 0.0;
}
/* rows of level l on a stretched mesh: metrics at the fine-grid point of grid row i (src/solver.c:227-232), OpA with the level's
 * computational spacing (src/problem.c:3-22).  ctab: n x 5, dtab: n (host) */
static void level_row_tables(int npts, int mesh, int l, int n, double *ctab, double *dtab) {
    double *cy = (double *)malloc(sizeof(double) * (size_t)npts);
    coords_mesh_y(npts, mesh, cy);
    const double h[2] = {1.0 / (n + 1), 1.0 / (n + 1)};
    const double hx2 = h[0] * h[0], hy2 = h[1] * h[1];
    const int f = 1 << l;
    for (int i = 0; i < n; i++) {
        double m[5];
        const int ifine = f * (i + 1) - 1;
        metrics_mesh(mesh, cy[ifine + 1], m);
        double *As = ctab + 5 * (size_t)i;
        As[0] = (m[1] / hy2) - (m[3] / (2 * h[1]));
        As[1] = (m[0] / hx2) - (m[2] / (2 * h[0]));
        As[2] = -2.0 * ((m[0] / hx2) + (m[1] / hy2));
        As[3] = (m[0] / hx2) + (m[2] / (2 * h[0]));
        As[4] = (m[1] / hy2) + (m[3] / (2 * h[1]));
        dtab[i] = 1.0 / As[2];
    }
    free(cy);
}

static void level_stencil(int dim, int n, double *As, double *h_out) {
    /* h: src/matbuild.c:99-104; OpA: src/problem.c:3-22 with MetricsUniform (src/mesh.c:29-43) */
    double h[3] = {1.0 / (n + 1), 1.0 / (n + 1), 1.0 / (n + 1)};
    double m[5] = {1.0, 1.0, 0.0, 0.0, 0.0};
    double hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];
    *h_out = h[0];
    if (dim == 2) {
        As[0] = (m[1] / hy2) - (m[3] / (2 * h[1]));
        As[1] = (m[0] / hx2) - (m[2] / (2 * h[0]));
        As[2] = -2.0 * ((m[0] / hx2) + (m[1] / hy2));
        As[3] = (m[0] / hx2) + (m[2] / (2 * h[0]));
        As[4] = (m[1] / hy2) + (m[3] / (2 * h[1]));
        return;
    }
    double mz = 1.0, mzz = 0.0;   /* 3-D extension */
    As[0] = (mz / hz2) - (mzz / (2 * h[2]));
    As[1] = (m[1] / hy2) - (m[3] / (2 * h[1]));
    As[2] = (m[0] / hx2) - (m[2] / (2 * h[0]));
    As[3] = -2.0 * (((m[0] / hx2) + (m[1] / hy2)) + (mz / hz2));
    As[4] = (m[0] / hx2) + (m[2] / (2 * h[0]));
    As[5] = (m[1] / hy2) + (m[3] / (2 * h[1]));
    As[6] = (mz / hz2) + (mzz / (2 * h[2]));
}

/* ------------------------------------------------------------------ */
/* create / destroy                                                    */
/* ------------------------------------------------------------------ */
void mg_config_default(mg_config *c) {
    memset(c, 0, sizeof(*c));
    c->dim = 2; c->npts = 17; c->levels = 2;     /* poisson.in:2,8-9 */
    c->v[0] = 3; c->v[1] = 3;                    /* poisson.in:12 */
    c->maxiter = 100000;                         /* poisson.in:6 */
    c->ksp_type = MG_KSP_RICHARDSON;             /* src/solver.c:1465 */
    c->scale = 1.0;                              /* PETSc default Richardson scale */
    c->emin = 0.0; c->emax = 0.0;
    c->rtol = 1.e-7;                             /* src/solver.c:1530 */
    c->device = 0; c->precision = MG_PREC_FP64;
    c->rank = 0; c->nranks = 1; c->dist_min_n = 0;
    c->fuse = -1;
    c->overlap = -1;
    c->graph = -1;
    c->pair_min_n = 0;
    c->slab_chunk = -1;
}

static int alloc_fset(mg_solver *s, mg_fset *F, int esz, int all4) {
    void **f[4] = {&F->u, &F->b, &F->rv, &F->tmp};
    for (int k = 0; k < (all4 ? 4 : 2); k++) {
        void *q = NULL;
        CHK(mgk_malloc(s->ctx, &q, (size_t)esz * (size_t)F->g.total));
        *f[k] = q;
    }
    return 0;
}

static int upload(mg_solver *s, const double *h, size_t n, double **d);

/* distributed level: the (nx, ny, 2) field through which the neighbours' second planes of u travel (two-sweep passes) */
static int alloc_far(mg_solver *s, mg_level *L, int P) {
    mg_fset *F = &L->f[P];
    if (!L->distributed || !(s->cfg.fuse & 32) || s->cfg.dim != 3) return 0;
    int rc = (P == 0) ? mgk_geom_init(&F->gfar, 3, L->n, L->n, 2) : mgk_geom_init_f32(&F->gfar, 3, L->n, L->n, 2);
    if (rc) return rc;
    void *q = NULL;
    const size_t bytes = (size_t)(P == 0 ? 8 : 4) * (size_t)F->gfar.total;
    CHK(mgk_malloc(s->ctx, &q, bytes));
    CHK(mgk_memset0(s->ctx, q, bytes, NULL));
    F->far = q;
    if (P == 0 && (s->cfg.fuse & 1024)) {
        void *q2 = NULL, *q3 = NULL;
        CHK(mgk_malloc(s->ctx, &q2, bytes));
        F->far2 = q2;
        CHK(mgk_malloc(s->ctx, &q3, bytes));
        F->bfar = q3;
        CHK(mgk_memset0(s->ctx, q2, bytes, NULL));
        CHK(mgk_memset0(s->ctx, q3, bytes, NULL));
    }
    return 0;
}

int mg_solver_create(mg_solver **out, const mg_config *cfg, mg_comm *comm) {
    if (!out || !cfg) return mgfail(MGK_EINVAL, "mg_solver_create: null argument");
    if (cfg->dim != 2 && cfg->dim != 3) return mgfail(MGK_EINVAL, "mg_solver_create: dim must be 2 or 3");
    if (cfg->levels < 1 || cfg->levels > MG_MAX_LEVELS) return mgfail(MGK_EINVAL, "mg_solver_create: bad level count");
    if (cfg->npts < 3) return mgfail(MGK_EINVAL, "mg_solver_create: npts < 3");
    if (cfg->precision == MG_PREC_MIXED && (cfg->dim != 3 || cfg->ksp_type != MG_KSP_RICHARDSON))
        return mgfail(MGK_EINVAL, "mg_solver_create: mixed precision is built for 3-D, Richardson+Jacobi");
    /* npts-1 must be divisible by 2^(levels-1) and the coarsest grid must keep >= 1 unknown */
    for (int l = 0; l < cfg->levels; l++) {
        int n = mg_grid_n(cfg->npts, l);
        int f = 1 << l;
        if (n < 1 || (cfg->npts - 1) % f != 0 || (n & 1) == 0)
            return mgfail(MGK_EINVAL, "mg_solver_create: npts-1 must be 2^m with m >= levels (vertex-centred coarsening, src/matbuild.c:62-66)");
    }
    if (cfg->mesh != 0 && (cfg->mesh < 0 || cfg->mesh > 2 || cfg->dim != 2 || cfg->precision != MG_PREC_FP64 || cfg->nranks > 1))
        return mgfail(MGK_EINVAL, "mg_solver_create: -mesh 1/2 is built for 2-D, fp64, one GPU");
    if (cfg->nranks > 1 && (!comm || cfg->dim != 3))
        return mgfail(MGK_EINVAL, "mg_solver_create: nranks > 1 needs a communicator and dim == 3");
    if (cfg->ksp_type == MG_KSP_CHEBYSHEV && !(cfg->emax > cfg->emin && cfg->emin > 0.0))
        return mgfail(MGK_EINVAL, "mg_solver_create: chebyshev needs 0 < emin < emax (-ksp_chebyshev_eigenvalues)");

    mg_solver *s = (mg_solver *)calloc(1, sizeof(mg_solver));
    s->cfg = *cfg;
    if (s->cfg.rtol <= 0) s->cfg.rtol = 1.e-7;
    if (s->cfg.dist_min_n <= 0) {               /* default; a fine grid smaller than that is still cut into slabs */
        const int n0 = mg_grid_n(cfg->npts, 0);
        s->cfg.dist_min_n = n0 < 255 ? n0 : 255;
    }
    if (s->cfg.fuse < 0) s->cfg.fuse = 63 | 256 | 512 | 1024 | 2048 | 4096 | 8192 | 16384;
    if (s->cfg.pair_min_n <= 0) s->cfg.pair_min_n = (cfg->dim == 3) ? 255 : 2047;   /* where a two-sweep pass beats two sweeps
                                                                                      * (255^3: 0.107 ms against 2 x 0.063) */
    if (s->cfg.mesh) s->cfg.fuse &= ~(16 | 128);   /* row-dependent coefficients (2-D, fp64): the same fused cycle on the row-table forms of the kernels */
    if (s->cfg.overlap < 0) s->cfg.overlap = 1;
    if (s->cfg.graph < 0) s->cfg.graph = 1;
    if (s->cfg.nranks < 1) s->cfg.nranks = 1;
    s->comm = comm;
    s->levels = cfg->levels;
    int rc = mgk_ctx_create(&s->ctx, cfg->device);
    if (rc) { free(s); return mgfail(rc, "mg_solver_create: mgk_ctx_create"); }
    if (s->cfg.slab_chunk < 0) {                /* default 32 planes; MG_SLAB_CHUNK overrides (0: the long streams of a single GPU) */
        const char *e = getenv("MG_SLAB_CHUNK");
        /* (round 3, second session) a quarter of the rank's fine planes, at least 32: the interior of a pass then frees a CU four times -- enough for an
         * exchange kernel that needs one (8 ranks at 1023^3: 32 planes as before; 2 ranks: 127 -- one rank's share 12.16 -> 11.60 ms, long streams 11.45) */
        const int quarter = (mg_grid_n(cfg->npts, 0) / (s->cfg.nranks > 0 ? s->cfg.nranks : 1)) / 4;
        s->cfg.slab_chunk = (e && *e && atoi(e) >= 0) ? atoi(e) : (quarter > 32 ? quarter : 32);
    }
    if (s->cfg.nranks > 1 && s->cfg.dim == 3) mgk_ctx_set_chunk_planes(s->ctx, s->cfg.slab_chunk);

    /* which levels are distributed */
    s->ldist = 0;
    if (s->cfg.nranks > 1) {
        for (int l = 0; l < s->levels; l++) {
            int n = mg_grid_n(cfg->npts, l);
            if (n >= s->cfg.dist_min_n && n >= 2 * s->cfg.nranks) s->ldist = l + 1; else break;
        }
        if (s->ldist == 0) { mg_solver_destroy(s); return mgfail(MGK_EINVAL, "mg_solver_create: grid too small to distribute"); }
        s->zstart = (int *)calloc(s->cfg.nranks + 1, sizeof(int));
        for (int r = 0; r < s->cfg.nranks; r++) {
            int a, b;
            if (mg_slab_range(cfg->npts, s->ldist, s->ldist, r, s->cfg.nranks, &a, &b)) {
                mg_solver_destroy(s);
                return mgfail(MGK_EINVAL, "mg_solver_create: too many ranks for this grid");
            }
            s->zstart[r] = a; s->zstart[r + 1] = b;   /* planes of level ldist produced by rank r */
        }
    }

    const int mixed = (cfg->precision == MG_PREC_MIXED);
    for (int l = 0; l < s->levels; l++) {
        mg_level *L = &s->L[l];
        L->n = mg_grid_n(cfg->npts, l);
        L->distributed = (l < s->ldist);
        L->z0 = 0; L->nzl = (cfg->dim == 3) ? L->n : 1;
        if (L->distributed) {
            int a, b;
            mg_slab_range(cfg->npts, s->ldist, l, s->cfg.rank, s->cfg.nranks, &a, &b);
            L->z0 = a; L->nzl = b - a;
        }
        L->nz_min = L->nzl;
        for (int r = 0; L->distributed && r < s->cfg.nranks; r++) {
            int a, b;
            mg_slab_range(cfg->npts, s->ldist, l, r, s->cfg.nranks, &a, &b);
            if (b - a < L->nz_min) L->nz_min = b - a;
        }
        rc = mgk_geom_init(&L->f[0].g, cfg->dim, L->n, L->n, L->nzl);
        if (rc) { mg_solver_destroy(s); return mgfail(rc, "mg_solver_create: geometry"); }
        level_stencil(cfg->dim, L->n, L->coef, &L->h);
        L->dinv = 1.0 / L->coef[cfg->dim == 3 ? 3 : 2];      /* PCJACOBI: 1/diag(A) */
        if (cfg->mesh) {
            double *hc = (double *)malloc(sizeof(double) * 5 * (size_t)L->n), *hd = (double *)malloc(sizeof(double) * (size_t)L->n);
            level_row_tables(cfg->npts, cfg->mesh, l, L->n, hc, hd);
            rc = upload(s, hc, 5 * (size_t)L->n, &L->ctab);
            if (!rc) rc = upload(s, hd, (size_t)L->n, &L->dtab);
            free(hc); free(hd);
            if (rc) { mg_solver_destroy(s); return mgfail(rc, "mg_solver_create: coefficient tables"); }
        }
        if (mixed) {
            /* fp64 only where the outer defect correction lives (level 0: u, b); fp32 on every level */
            if (l == 0 && (rc = alloc_fset(s, &L->f[0], 8, 0))) { mg_solver_destroy(s); return rc; }
            if (l == 0 && (s->cfg.fuse & 16) && !L->distributed) {          /* spare fp64 field of the fused correction + residual pass */
                void *q = NULL;
                if ((rc = mgk_malloc(s->ctx, &q, sizeof(double) * (size_t)L->f[0].g.total))) { mg_solver_destroy(s); return mgfail(rc, "mg_solver_create: field"); }
                L->f[0].tmp = q;
            }
            if ((rc = mgk_geom_init_f32(&L->f[1].g, 3, L->n, L->n, L->nzl))) { mg_solver_destroy(s); return mgfail(rc, "mg_solver_create: fp32 geometry"); }
            if ((rc = alloc_fset(s, &L->f[1], 4, 1))) { mg_solver_destroy(s); return rc; }
            if ((rc = alloc_far(s, L, 1))) { mg_solver_destroy(s); return rc; }
            continue;
        }
        if ((rc = alloc_fset(s, &L->f[0], 8, 1))) { mg_solver_destroy(s); return rc; }
        if ((rc = alloc_far(s, L, 0))) { mg_solver_destroy(s); return rc; }
        if (cfg->ksp_type == MG_KSP_CHEBYSHEV) {
            void *q = NULL;
            if ((rc = mgk_malloc(s->ctx, &q, sizeof(double) * (size_t)L->f[0].g.total))) { mg_solver_destroy(s); return mgfail(rc, "mg_solver_create: field"); }
            L->p2 = (double *)q;
        }
    }
    /* coarse part for the HIP graph: the first level that is neither distributed nor fed by a distributed one and
     * has at most 2^21 unknowns (3-D n <= 127, 2-D n <= 1023): below that a kernel is shorter than its launch */
    s->lgraph = 0;
    if (s->cfg.graph && s->cfg.ksp_type == MG_KSP_RICHARDSON) {
        for (int l = (s->ldist > 0 ? s->ldist + 1 : 1); l < s->levels; l++) {
            double N = pow((double)s->L[l].n, (double)cfg->dim);
            if (N <= 2097152.0) { s->lgraph = l; break; }
        }
        if (s->lgraph && s->levels - s->lgraph < 2) s->lgraph = 0;      /* not worth a graph */
    }
    /* the tail: the first level l >= 1 that is whole on this rank, fed by a whole level... any level qualifies as long as it
     * and everything below it fit in LDS (n <= mgk_tail_max_n) -- and at least two levels are left, else a tail is no gain */
    s->ltail = 0;
    if ((s->cfg.fuse & 512) && s->cfg.ksp_type == MG_KSP_RICHARDSON && s->cfg.v[0] >= 1) {
        for (int l = (s->ldist > 0 ? s->ldist : 1); l < s->levels; l++)
            if (s->L[l].n <= mgk_tail_max_n(cfg->dim)) { s->ltail = l; break; }
        if (s->ltail && (s->levels - s->ltail < 2 || s->levels - s->ltail > 8)) s->ltail = 0;
    }
    s->rnorm_cap = (cfg->maxiter > 0 ? cfg->maxiter : 0) + 1;
    s->rnorm = (double *)calloc((size_t)s->rnorm_cap, sizeof(double));
    *out = s;
    return 0;
}

void mg_solver_destroy(mg_solver *s) {
    if (!s) return;
    if (s->ctx) {
        mgk_sync(s->ctx, NULL);
        for (int q = 0; q < s->ntimers_created; q++) mgk_timer_destroy(s->ctx, s->timers[q]);
        if (s->d_norms) mgk_free(s->ctx, s->d_norms);
        if (s->pin) mgk_host_free(s->ctx, s->pin);
        for (int p = 0; p < 2; p++) if (s->coarse_graph[p]) mgk_graph_destroy(s->ctx, s->coarse_graph[p]);
        for (int l = 0; l < s->levels; l++) {
            mg_level *L = &s->L[l];
            for (int p = 0; p < 2; p++) {
                void *f[7] = {L->f[p].u, L->f[p].b, L->f[p].rv, L->f[p].tmp, L->f[p].far, L->f[p].far2, L->f[p].bfar};
                for (int k = 0; k < 7; k++) if (f[k]) mgk_free(s->ctx, f[k]);
            }
            if (L->p2) mgk_free(s->ctx, L->p2);
            if (L->ctab) mgk_free(s->ctx, L->ctab);
            if (L->dtab) mgk_free(s->ctx, L->dtab);
        }
        mgk_ctx_destroy(s->ctx);
    }
    free(s->zstart);
    free(s->rnorm);
    free(s);
}

/* ------------------------------------------------------------------ */
/* right-hand side, solution, error                                    */
/* ------------------------------------------------------------------ */
static int upload(mg_solver *s, const double *h, size_t n, double **d) {
    void *q = NULL;
    CHK(mgk_malloc(s->ctx, &q, sizeof(double) * n));
    CHK(mgk_h2d(s->ctx, q, h, sizeof(double) * n));
    *d = (double *)q;
    return 0;
}

/* sin(pi x) tables of the interior nodes; cx additionally carries the constant of Ffunc:
 * -2*PI*PI*sin(PI*x)*sin(PI*y) evaluates left to right as ((-2*PI)*PI)*sin(PI*x) then *sin(PI*y)
 * (src/problem.c:27), 3-D extension -3*PI*PI*sin*sin*sin */
static int sin_tables(mg_solver *s, double **cx, double **sx, double **sy, double **sz) {
    const mg_level *L = &s->L[0];
    int npts = s->cfg.npts, n = L->n;
    double *c = (double *)malloc(sizeof(double) * npts);
    double *t = (double *)malloc(sizeof(double) * n), *tc = (double *)malloc(sizeof(double) * n);
    if (!c || !t || !tc) { free(c); free(t); free(tc); return mgfail(MGK_EINVAL, "sin_tables: out of host memory"); }
    coords_uniform(npts, c);
    for (int j = 0; j < n; j++) {
        t[j] = sin(MG_PI * c[j + 1]);
        tc[j] = (s->cfg.dim == 2 ? -2 * MG_PI * MG_PI : -3 * MG_PI * MG_PI) * t[j];
    }
    int rc = 0;
    if (cx) rc = upload(s, tc, n, cx);
    if (!rc && sx) rc = upload(s, t, n, sx);
    if (s->cfg.mesh) {                                   /* stretched in y: sin(pi y_i) at the mesh's own coordinates */
        coords_mesh_y(npts, s->cfg.mesh, c);
        for (int j = 0; j < n; j++) t[j] = sin(MG_PI * c[j + 1]);
    }
    if (!rc) rc = upload(s, t, n, sy);
    if (!rc && s->cfg.dim == 3) rc = upload(s, t + L->z0, L->nzl, sz); else if (!rc) *sz = NULL;
    free(c); free(t); free(tc);
    return rc;
}

int mg_solver_set_rhs_problem(mg_solver *s) {
    double *cx = NULL, *sy = NULL, *sz = NULL;
    mg_fset *F = &s->L[0].f[0];
    CHK(sin_tables(s, &cx, NULL, &sy, &sz));
    CHK(mgk_fill_separable_f64(s->ctx, &F->g, cx, sy, sz, (double *)F->b, NULL));
    CHK(mgk_sync(s->ctx, NULL));
    mgk_free(s->ctx, cx); mgk_free(s->ctx, sy); if (sz) mgk_free(s->ctx, sz);
    return mg_solver_reset(s);
}

int mg_solver_set_rhs_host(mg_solver *s, const double *b_compact) {
    mg_fset *F = &s->L[0].f[0];
    size_t n = (size_t)F->g.nx * F->g.ny * F->g.nz;
    double *d = NULL;
    CHK(upload(s, b_compact, n, &d));
    CHK(mgk_pack_f64(s->ctx, &F->g, d, (double *)F->b, NULL));
    CHK(mgk_sync(s->ctx, NULL));
    mgk_free(s->ctx, d);
    return mg_solver_reset(s);
}

static int finalize_iterate(mg_solver *s);
int mg_solver_get_solution(mg_solver *s, double *u_compact) {
    CHK(finalize_iterate(s));
    mg_fset *F = &s->L[0].f[0];
    size_t n = (size_t)F->g.nx * F->g.ny * F->g.nz;
    void *d = NULL;
    CHK(mgk_malloc(s->ctx, &d, sizeof(double) * n));
    CHK(mgk_unpack_f64(s->ctx, &F->g, (const double *)F->u, (double *)d, NULL));
    CHK(mgk_d2h(s->ctx, u_compact, d, sizeof(double) * n));
    mgk_free(s->ctx, d);
    return 0;
}

int mg_solver_error_norms(mg_solver *s, double err[3]) {
    CHK(finalize_iterate(s));
    double *sx = NULL, *sy = NULL, *sz = NULL;
    mg_fset *F = &s->L[0].f[0];
    CHK(sin_tables(s, NULL, &sx, &sy, &sz));
    double e[3];
    CHK(mgk_error_sums_f64(s->ctx, &F->g, (const double *)F->u, sx, sy, sz, e, NULL));
    mgk_free(s->ctx, sx); mgk_free(s->ctx, sy); if (sz) mgk_free(s->ctx, sz);
    if (s->cfg.nranks > 1) {
        /* sums through the all-reduce; the max by all-reducing a one-hot vector (ranks are few) */
        double v[2] = {e[1], e[2]};
        CHK(s->comm->allreduce_sum(s->comm, s->ctx, v, 2, NULL));
        e[1] = v[0]; e[2] = v[1];
        double *m = (double *)calloc((size_t)s->cfg.nranks, sizeof(double));
        m[s->cfg.rank] = e[0];
        int rc = s->comm->allreduce_sum(s->comm, s->ctx, m, s->cfg.nranks, NULL);
        if (rc) { free(m); return mgfail(rc, "error_norms: allreduce"); }
        for (int r = 0; r < s->cfg.nranks; r++) e[0] = fmax(e[0], m[r]);
        free(m);
    }
    err[0] = e[0]; err[1] = e[1]; err[2] = sqrt(e[2]);
    return 0;
}

/* ------------------------------------------------------------------ */
/* the cycle (written once for both precisions: P = 0 fp64, 1 fp32)    */
/* ------------------------------------------------------------------ */
/* Every RCCL operation of a solver is issued on ITS comm stream (one communicator, one stream: a
 * single total order on every rank); cross-stream events tie it to the compute stream.
 * Blocking form: the exchange sees everything queued on the compute stream so far, and everything
 * queued on the compute stream afterwards sees the ghosts. */
static int halo(mg_solver *s, int P, mg_level *L, void *field) {
    if (!L->distributed) return 0;
    void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
    CHK(mgk_stream_wait(s->ctx, ms, cs));
    CHK(s->comm->halo(s->comm, s->ctx, field, &L->f[P].g, OPS[P].esz, ms));
    CHK(mgk_stream_wait(s->ctx, cs, ms));
    return 0;
}

/* start the exchange of u's ghost planes on the comm stream unless they are valid or already travelling: the caller
 * queues work that needs no ghost plane on the compute stream, then calls ensure_u_ghosts */
static int begin_u_ghosts(mg_solver *s, int P, mg_level *L) {
    mg_fset *F = &L->f[P];
    if (!L->distributed || F->u_ghost_ok || F->u_ghost_pending) return 0;
    void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
    CHK(mgk_stream_wait(s->ctx, ms, cs));
    CHK(s->comm->halo(s->comm, s->ctx, F->u, &F->g, OPS[P].esz, ms));
    F->u_ghost_pending = 1;
    return 0;
}
/* make the ghost planes of u valid and visible to the compute stream */
static int ensure_u_ghosts(mg_solver *s, int P, mg_level *L) {
    mg_fset *F = &L->f[P];
    if (!L->distributed) return 0;
    CHK(begin_u_ghosts(s, P, L));
    if (F->u_ghost_pending) {
        CHK(mgk_stream_wait(s->ctx, mgk_stream_compute(s->ctx), mgk_stream_comm(s->ctx)));
        F->u_ghost_pending = 0;
        F->u_ghost_ok = 1;
    }
    return 0;
}

static void *prof_begin(mg_solver *s, int level) {
    if (!s->prof_on || level != 0 || s->prof_n >= MG_MAX_TIMERS) return NULL;
    if (s->prof_n >= s->ntimers_created) {
        if (mgk_timer_create(s->ctx, &s->timers[s->ntimers_created])) return NULL;
        s->ntimers_created++;
    }
    s->timer_kind[s->prof_n] = (unsigned char)s->prof_kind;
    void *t = s->timers[s->prof_n++];
    mgk_timer_start(s->ctx, t, NULL);
    return t;
}
static void prof_end(mg_solver *s, void *t) { if (t) mgk_timer_stop(s->ctx, t, NULL); }

int mg_solver_profile(mg_solver *s, int enable) { s->prof_on = enable; s->prof_n = 0; return 0; }
int mg_solver_profile_read_kind(mg_solver *s, int kind, double *total_ms, int *launches) {
    double tot = 0.0, ms;
    int n = 0;
    for (int q = 0; q < s->prof_n; q++) {
        if (s->timer_kind[q] != kind) continue;
        CHK(mgk_timer_elapsed_ms(s->ctx, s->timers[q], &ms));
        tot += ms; n++;
    }
    *total_ms = tot; *launches = n;
    return 0;
}
int mg_solver_profile_read(mg_solver *s, double *total_ms, int *launches) {
    return mg_solver_profile_read_kind(s, 0, total_ms, launches);
}
static void swap_ptr(void **a, void **b) { void *t = *a; *a = *b; *b = t; }

/* ONE grouped exchange of a pass on a z-slab (one latency): u's ghost planes unless valid / travelling, b's ghost planes once per right-hand
 * side, the neighbours' SECOND planes of u (far), and -- for the sweep inside the restriction -- the upper neighbour's THIRD plane of u (far2)
 * and SECOND plane of b (bfar, once per right-hand side).  With a transport that can send any plane (mg_comm.exchange: rccl, peer, loopback,
 * phantom) the planes 1, nz-2, 2 of u and 1 of b leave from where they are; otherwise (host-staged) they are first copied into the interior
 * planes of the far fields, whose ordinary halo then carries them (round 2: 20 plane copies per cycle on the compute stream).
 * Queued on the comm stream behind everything the compute stream holds so far. */
static int group_exchange(mg_solver *s, int P, mg_fset *F, int with_far2) {
    const mg_ops *O = &OPS[P];
    const size_t pb = (size_t)O->esz * (size_t)F->g.plane;
    const int nz = F->g.nz;
    void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
    const int want_u = !F->u_ghost_pending && !F->u_ghost_ok, want_b = !F->b_ghost_ok, want_bfar = with_far2 && !F->bfar_ok;
    char *u = (char *)F->u, *b = (char *)F->b, *far = (char *)F->far, *far2 = (char *)F->far2, *bfar = (char *)F->bfar;
    if (s->comm->exchange) {
        const void *slo[5], *shi[5];
        void *rlo[5], *rhi[5];
        size_t nb[5];
        int n = 0;
#define SLOT(a, b_, c, d) do { slo[n] = (a); shi[n] = (b_); rlo[n] = (c); rhi[n] = (d); nb[n++] = pb; } while (0)
        if (want_u) SLOT(u + pb, u + (size_t)nz * pb, u, u + (size_t)(nz + 1) * pb);
        if (want_b) SLOT(b + pb, b + (size_t)nz * pb, b, b + (size_t)(nz + 1) * pb);
        SLOT(u + 2 * pb, u + (size_t)(nz - 1) * pb, far, far + 3 * pb);                 /* my planes 1 / nz-2 -> their far hi / lo ghost */
        if (with_far2) SLOT(u + 3 * pb, NULL, NULL, far2 + 3 * pb);                     /* my plane 2 -> the lower neighbour's far2 hi ghost */
        if (want_bfar) SLOT(b + 2 * pb, NULL, NULL, bfar + 3 * pb);                     /* my plane 1 of b */
#undef SLOT
        CHK(mgk_stream_wait(s->ctx, ms, cs));
        CHK(s->comm->exchange(s->comm, s->ctx, n, slo, shi, rlo, rhi, nb, ms));
        return 0;
    }
    CHK(mgk_d2d(s->ctx, far + pb, u + 2 * pb, pb, cs));                                 /* my plane 1 */
    CHK(mgk_d2d(s->ctx, far + 2 * pb, u + (size_t)(nz - 1) * pb, pb, cs));              /* my plane nz-2 */
    if (with_far2) CHK(mgk_d2d(s->ctx, far2 + pb, u + 3 * pb, pb, cs));                 /* my plane 2 */
    if (want_bfar) CHK(mgk_d2d(s->ctx, bfar + pb, b + 2 * pb, pb, cs));                 /* my plane 1 of b */
    CHK(mgk_stream_wait(s->ctx, ms, cs));
    void *ff[5];
    const mgk_geom *gg[5];
    int nf = 0;
    if (want_u) { ff[nf] = F->u; gg[nf++] = &F->g; }
    if (want_b) { ff[nf] = F->b; gg[nf++] = &F->g; }
    ff[nf] = F->far; gg[nf++] = &F->gfar;
    if (with_far2) { ff[nf] = F->far2; gg[nf++] = &F->gfar; }
    if (want_bfar) { ff[nf] = F->bfar; gg[nf++] = &F->gfar; }
    CHK(mg_comm_halo_n(s->comm, s->ctx, nf, ff, gg, O->esz, ms));
    return 0;
}

/* KSPSolve(ksp[l], b[l], u[l]) with KSPCHEBYSHEV (fp64 only), classic three-term recurrence (oracle/mgo.c) */
static int smooth_chebyshev(mg_solver *s, int l, int maxit) {
    mg_level *L = &s->L[l];
    mg_fset *F = &L->f[0];
    const size_t bytes = sizeof(double) * (size_t)F->g.total;
    double scale = 2.0 / (s->cfg.emax + s->cfg.emin), alpha = 1.0 - scale * s->cfg.emin, Gamma = 1.0;
    double mu = 1.0 / alpha, omegaprod = 2.0 / alpha, ckm1 = 1.0, ck = mu, ckp1;
    double *pkm1 = (double *)F->u, *pk = (double *)F->tmp, *pkp1 = L->p2;
    const int mesh = s->cfg.mesh != 0;                              /* -mesh 1/2: the same steps on the level's row tables (2-D, one rank) */
    /* the recurrence takes its first step BEFORE its loop (PETSc's cheby.c; oracle/mgo.c: smooth, mgo_chebyshev_csr): with max_it = 0 the
     * solve still makes that one step (round 3: the product returned the guess untouched -- found by the random draws over the mock) */
    if (!F->guess_nonzero) {
        CHK(mgk_memset0(s->ctx, pkm1, bytes, NULL));
        F->u_ghost_ok = 0; F->u_ghost_pending = 0;
        if (mesh) CHK(mgk_jacobi_zero_rowcoef_f64(s->ctx, &F->g, L->dtab, scale, (const double *)F->b, pk, NULL));
        else CHK(mgk_jacobi_zero_f64(s->ctx, &F->g, L->dinv, scale, (const double *)F->b, pk, NULL));
    } else {
        CHK(ensure_u_ghosts(s, 0, L));
        if (mesh) CHK(mgk_rowcoef_f64(s->ctx, &F->g, 0, L->ctab, L->dtab, scale, (const double *)F->b, pkm1, pk, NULL));
        else CHK(mgk_jacobi_f64(s->ctx, &F->g, L->coef, L->dinv, scale, (const double *)F->b, pkm1, pk, NULL));
    }
    for (int it = 1; it < maxit; it++) {
        ckp1 = 2.0 * mu * ck - ckm1;
        double omega = omegaprod * ck / ckp1;
        CHK(halo(s, 0, L, pk));
        if (mesh) CHK(mgk_cheby_rowcoef_f64(s->ctx, &F->g, L->ctab, L->dtab, 1.0 - omega, omega, omega * Gamma * scale,
                                            (const double *)F->b, pk, pkm1, pkp1, NULL));
        else CHK(mgk_cheby_f64(s->ctx, &F->g, L->coef, L->dinv, 1.0 - omega, omega, omega * Gamma * scale,
                          (const double *)F->b, pk, pkm1, pkp1, NULL));
        double *t = pkm1; pkm1 = pk; pk = pkp1; pkp1 = t;
        ckm1 = ck; ck = ckp1;
    }
    F->u = pk; F->tmp = pkm1; L->p2 = pkp1;
    F->u_ghost_ok = 0; F->u_ghost_pending = 0;
    return 0;
}

/* KSPSolve(ksp[l], b[l], u[l]) with KSP_NORM_NONE and max_it = maxit (src/solver.c:1465-1509) */
/* the last pre-smoothing sweep of level l can be left to the restriction that follows (one pass for sweep + residual + full weighting,
 * fuse bit 10): fp64, whole 3-D grids of full-row shape.  Not when that restriction is the first kernel of the coarse-level graph while
 * this level's sweeps run outside it (the replayed kernel would keep the buffers of the recording cycle; the swap is made on the host) */
static int srr_ok(const mg_solver *s, int P, int l) {
    if (!(s->cfg.fuse & 1024) || P != 0 || s->cfg.ksp_type != MG_KSP_RICHARDSON) return 0;
    if (l + 1 >= s->levels || (s->lgraph && l + 1 == s->lgraph)) return 0;
    /* 2-D (mgk_sweep_residual_restrict_2d_f64, any grid): from 2047^2 on; on the smaller levels of the 4097^2 cycle its marching waves
     * take 15-20 us where a short sweep and the short fused restriction take 5 + 10 (MG_SRR2D_MIN_N overrides, for tests) */
    if (s->cfg.dim == 2) {
        const char *e = getenv("MG_SRR2D_MIN_N");
        return s->L[l].n >= ((e && *e) ? atoi(e) : 2047);
    }
    if (s->L[l].distributed) {
        /* z-slab: the neighbours' planes arrive in ONE grouped exchange (u and b ghosts, far, far2, bfar) */
        const mg_fset *F = &s->L[l].f[0];
        mgk_geom gc = s->L[l + 1].f[0].g;
        if (!s->L[l + 1].distributed) gc.nz = s->zstart[s->cfg.rank + 1] - s->zstart[s->cfg.rank];
        return F->far && F->far2 && F->bfar && s->L[l].nz_min >= 6 && mgk_sweep_residual_restrict_slab_ok_f64(&F->g, &gc);
    }
    return mgk_sweep_residual_restrict_ok_f64(&s->L[l].f[0].g, &s->L[l + 1].f[0].g);
}

/* fuse bit 11: a pre-smoothing KSPSolve of >= 3 sweeps from the zero guess starts with ONE pass that makes three of them and reads b alone
 * (mgk_jacobi2_zero_*): whole 3-D levels that sweep in pairs, fp32 up to 1023^3, fp64 up to 511^3.  The producers of such a level's
 * right-hand side then do not write its zero-guess sweep (fuse bit 8). */
/* fuse bit 13 (2-D, fp64, Richardson; uniform and stretched meshes): THREE sweeps per pass on every level the cycle launches kernels for
 * (mgk_jacobi3_2d_*): a KSPSolve of >= 3 sweeps from the zero guess starts with one pass over b (16 B), the post-smoothing is ONE pass with
 * the prolongation (25 B), and the norm that closes a cycle makes all three pre-smoothing sweeps of the next one (24 B): with the fused
 * residual + restriction (18 B) a V(3,3) cycle makes three passes over a level, 67 B on the fine level and 59 B below, instead of four. */
static int j3_2d_ok(const mg_solver *s, int P, int l, int maxit) {
    return (s->cfg.fuse & 8192) && s->cfg.dim == 2 && P == 0 && s->cfg.ksp_type == MG_KSP_RICHARDSON && maxit >= 3 && !s->L[l].distributed;
}
static int triple_ok(const mg_solver *s, int P, int l, int maxit) {
    const mg_level *L = &s->L[l];
    if (j3_2d_ok(s, P, l, maxit)) return 1;
    if (!(s->cfg.fuse & 2048) || !(s->cfg.fuse & 32) || s->cfg.ksp_type != MG_KSP_RICHARDSON || s->cfg.dim != 3 || s->cfg.mesh) return 0;
    if (maxit < 3 || L->distributed || L->n < s->cfg.pair_min_n || L->n + 1 > 1024) return 0;
    return P == 0 ? mgk_jacobi2_zero_ok_f64(&L->f[0].g) : mgk_jacobi2_zero_ok_f32(&L->f[1].g);
}

/* pre: pre-smoothing, a restriction from this level follows (src/solver.c:1531 / :1536 before :1534 of the next level) */
static int smooth(mg_solver *s, int P, int l, int maxit, int pre) {
    if (s->cfg.ksp_type == MG_KSP_CHEBYSHEV) return smooth_chebyshev(s, l, maxit);
    mg_level *L = &s->L[l];
    mg_fset *F = &L->f[P];
    const mg_ops *O = &OPS[P];
    if (maxit == 0 && !F->guess_nonzero) {                          /* KSPSolve zero-fills */
        CHK(mgk_memset0(s->ctx, F->u, (size_t)O->esz * (size_t)F->g.total, NULL));
        F->u_ghost_ok = 0; F->u_ghost_pending = 0;
    }
    int it0 = 0;
    F->last_sweep_pending = 0;
    if (l == 0 && P == 0 && s->spec_valid && maxit >= s->spec_valid && F->guess_nonzero) {
        /* the first sweep(s) were already made by the kernel that evaluated the previous cycle's residual norm */
        swap_ptr(&F->u, &F->tmp);
        F->u_ghost_ok = 0; F->u_ghost_pending = 0;
        it0 = s->spec_valid;
        s->spec_valid = 0;
        s->iterate_behind = 0;              /* (u is now the sweep AFTER the iterate the last norm belongs to) */
    }
    const int defer_last = pre && srr_ok(s, P, l);
    /* two sweeps per pass (temporal blocking) where it pays (3-D from 255^3, 2-D from 2047^2).  Also on the level whose buffers
     * the coarse-level HIP graph refers to: pre- and post-smoothing group their v0 sweeps in the same way (one launch for the
     * first sweep -- zero guess / fused prolongation / adopted speculative sweep -- then pairs), so a level swaps u/tmp the
     * same number of times on the way down and on the way up: an even count per cycle, the pointers the graph recorded stay valid */
    const int pair_ok = (s->cfg.fuse & 32) && L->n >= s->cfg.pair_min_n &&
                        ((s->cfg.dim == 3 && L->n + 1 <= 1024 && (!L->distributed || (F->far && L->nz_min >= 4))) ||
                         (s->cfg.dim == 2 && P == 0));
    if (maxit < 1 || F->guess_nonzero) F->jz_ready = 0;
    const int mesh = s->cfg.mesh != 0;                   /* -mesh 1/2 (2-D, fp64, one GPU): the row-table forms of the same kernels */
    for (int it = it0; it < maxit; it++) {
        if (it == 0 && !F->jz_ready && j3_2d_ok(s, P, l, maxit)) {
            /* 2-D: sweeps 1-3 in one pass -- over b alone from the zero guess, over u and b otherwise (level 0 of a cycle that has no
             * speculative sweeps to adopt).  The result goes to tmp and is swapped in: ONE swap in every case (zero guess, plain, adopted
             * norm pass above), like the one pass of the post-smoothing (prolong_smooth) -- the same even count in every cycle, so the
             * pointers the coarse-level graph recorded stay valid */
            if (!F->guess_nonzero)
                CHK(mgk_jacobi3_2d_zero_f64(s->ctx, &F->g, mesh ? NULL : L->coef, mesh ? 1.0 : L->dinv, s->cfg.scale, mesh ? L->ctab : NULL,
                                            mesh ? L->dtab : NULL, (const double *)F->b, (double *)F->tmp, NULL));
            else
                CHK(mgk_jacobi3_2d_f64(s->ctx, &F->g, mesh ? NULL : L->coef, mesh ? 1.0 : L->dinv, s->cfg.scale, mesh ? L->ctab : NULL,
                                       mesh ? L->dtab : NULL, (const double *)F->b, (const double *)F->u, (double *)F->tmp, NULL));
            swap_ptr(&F->u, &F->tmp);
            F->u_ghost_ok = 0; F->u_ghost_pending = 0;
            it += 2;
            continue;
        }
        if (it == 0 && !F->guess_nonzero && !F->jz_ready && triple_ok(s, P, l, maxit)) {
            /* sweeps 1-3 from the zero guess in one pass over b; u is not an input, so the result lands in u itself: no swap (the
             * two swaps of the zero-guess sweep and the pair it replaces cancel, the pointers the coarse-level graph holds stay valid) */
            if (P == 0) CHK(mgk_jacobi2_zero_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (double *)F->u, NULL));
            else CHK(mgk_jacobi2_zero_f32(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const float *)F->b, (float *)F->u, NULL));
            F->u_ghost_ok = 0; F->u_ghost_pending = 0;
            it += 2;
            continue;
        }
        if (it == 0 && !F->guess_nonzero) {
            /* r = b, x = 0 + scale*(B b): u is not read (already in tmp when the fused residual+restriction wrote it) */
            if (!F->jz_ready) {
                if (mesh) CHK(mgk_jacobi_zero_rowcoef_f64(s->ctx, &F->g, L->dtab, s->cfg.scale, (const double *)F->b, (double *)F->tmp, NULL));
                else CHK(O->jacobi_zero(s->ctx, &F->g, L->dinv, s->cfg.scale, F->b, F->tmp, NULL));
            }
            F->jz_ready = 0;
        } else if (defer_last && it == maxit - 1) {
            F->last_sweep_pending = 1;                      /* made by the restriction's kernel (descend_restrict) */
            break;
        } else if (pair_ok && maxit - it >= 2) {
            if (L->distributed) {
                /* slab: the second sweep of my first / last plane needs the first sweep of the neighbour's last / first plane,
                 * i.e. TWO of its planes of u (one is the regular ghost plane) and its b on that plane */
                const int nz = F->g.nz, lo = s->cfg.rank > 0, hi = s->cfg.rank < s->cfg.nranks - 1;
                void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
                /* all exchanges of this pass on the comm stream: ONE group (the far planes, u's ghosts unless valid / travelling, b's once) ... */
                CHK(group_exchange(s, P, F, 0));
                /* ... while the planes 2 .. nz-3, which need no ghost data, are already being swept */
                if (s->cfg.overlap && L->nz_min >= 6) {
                    s->prof_kind = 1;                       /* timed: the interior planes 2 .. nz-3 of the slab */
                    void *t = prof_begin(s, l);
                    s->prof_kind = 0;
                    CHK(O->jacobi2_slab(s->ctx, &F->g, &F->gfar, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, F->far, lo, hi, 2, nz - 2, cs));
                    prof_end(s, t);
                }
                CHK(mgk_stream_wait(s->ctx, cs, ms));
                F->u_ghost_pending = 0; F->u_ghost_ok = 1; F->b_ghost_ok = 1;
                if (s->cfg.overlap && L->nz_min >= 6) {
                    CHK(O->jacobi2_slab(s->ctx, &F->g, &F->gfar, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, F->far, lo, hi, 0, 2, cs));
                    CHK(O->jacobi2_slab(s->ctx, &F->g, &F->gfar, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, F->far, lo, hi, nz - 2, nz, cs));
                } else {
                    CHK(O->jacobi2_slab(s->ctx, &F->g, &F->gfar, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, F->far, lo, hi, 0, nz, cs));
                }
            } else if (s->cfg.dim == 2) {
                s->prof_kind = 1;
                void *t = prof_begin(s, l);
                s->prof_kind = 0;
                if (mesh) CHK(mgk_jacobi2_2d_rowcoef_f64(s->ctx, &F->g, L->ctab, L->dtab, s->cfg.scale, (const double *)F->b,
                                                         (const double *)F->u, (double *)F->tmp, NULL));
                else CHK(mgk_jacobi2_2d_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                            (double *)F->tmp, NULL));
                prof_end(s, t);
            } else {
                s->prof_kind = 1;
                void *t = prof_begin(s, l);
                s->prof_kind = 0;
                CHK(O->jacobi2(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, NULL));
                prof_end(s, t);
            }
            it++;                                   /* this pass made sweeps it and it + 1 */
        } else if (L->distributed && s->cfg.overlap && L->nz_min >= 3) {
            /* boundary planes first, ship them on the comm stream, sweep the interior meanwhile */
            void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
            const int nz = F->g.nz;
            CHK(ensure_u_ghosts(s, P, L));
            CHK(O->jacobi_range(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, 0, 1, cs));
            CHK(O->jacobi_range(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, nz - 1, nz, cs));
            CHK(mgk_stream_wait(s->ctx, ms, cs));
            void *t = prof_begin(s, l);
            CHK(O->jacobi_range(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, 1, nz - 1, cs));
            prof_end(s, t);
            CHK(s->comm->halo(s->comm, s->ctx, F->tmp, &F->g, O->esz, ms));
            swap_ptr(&F->u, &F->tmp);
            F->u_ghost_pending = 1; F->u_ghost_ok = 0;
            continue;
        } else {
            CHK(ensure_u_ghosts(s, P, L));
            void *t = prof_begin(s, l);
            if (mesh) CHK(mgk_rowcoef_f64(s->ctx, &F->g, 0, L->ctab, L->dtab, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                          (double *)F->tmp, NULL));
            else CHK(O->jacobi_range(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, F->b, F->u, F->tmp, 0,
                                     F->g.dim == 3 ? F->g.nz : F->g.ny, NULL));
            prof_end(s, t);
        }
        F->u_ghost_ok = 0; F->u_ghost_pending = 0;
        swap_ptr(&F->u, &F->tmp);
    }
    return 0;
}

/* KSPBuildResidual(ksp[l],NULL,rv[l],&r) : rv = b - A u (src/solver.c:1534,1545) */
static int residual(mg_solver *s, int P, int l) {
    mg_level *L = &s->L[l];
    mg_fset *F = &L->f[P];
    CHK(ensure_u_ghosts(s, P, L));
    if (s->cfg.mesh) return mgk_rowcoef_f64(s->ctx, &F->g, 1, L->ctab, L->dtab, 1.0, (const double *)F->b, (const double *)F->u, (double *)F->rv, NULL);
    CHK(OPS[P].residual(s->ctx, &F->g, L->coef, F->b, F->u, F->rv, NULL));
    return 0;
}

static int norm_from_sumsq(mg_solver *s, double ss, double *out) {
    if (s->cfg.nranks > 1) CHK(s->comm->allreduce_sum(s->comm, s->ctx, &ss, 1, NULL));
    *out = sqrt(ss);
    return 0;
}

/* Sum over ranks of n device doubles that kernels on the compute stream deposited (mgk_defer_result), delivered to the host.
 * With a device all-reduce hook nothing blocks the compute stream: the comm stream waits for the producers, reduces in
 * place (RCCL: ncclAllReduce, in the communicator's one total order behind any halo still travelling), copies to pinned
 * memory, and the host waits for the comm stream alone. */
static int reduce_slots(mg_solver *s, double *dslots, int n, double *host) {
    if (n < 1) return 0;
    if (s->cfg.nranks > 1 && s->comm->allreduce_sum_dev) {
        void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
        if (n > s->pin_cap) {
            if (s->pin) mgk_host_free(s->ctx, s->pin);
            void *q = NULL;
            s->pin = NULL; s->pin_cap = 0;
            CHK(mgk_host_alloc(s->ctx, &q, sizeof(double) * (size_t)n));
            s->pin = (double *)q; s->pin_cap = n;
        }
        CHK(mgk_stream_wait(s->ctx, ms, cs));
        CHK(s->comm->allreduce_sum_dev(s->comm, s->ctx, dslots, n, ms));
        CHK(mgk_d2h_async(s->ctx, s->pin, dslots, sizeof(double) * (size_t)n, ms));
        CHK(mgk_sync(s->ctx, ms));
        memcpy(host, s->pin, sizeof(double) * (size_t)n);
        if (s->comm->check) { int rcc = s->comm->check(s->comm); if (rcc) return mgfail(rcc, mg_comm_last_error()); }   /* did every exchange arrive? */
        return 0;
    }
    CHK(mgk_d2h(s->ctx, host, dslots, sizeof(double) * (size_t)n));                    /* synchronises the compute stream */
    for (int q = 0; q < n && s->cfg.nranks > 1; q += 64)
        CHK(s->comm->allreduce_sum(s->comm, s->ctx, host + q, n - q < 64 ? n - q : 64, NULL));
    return 0;
}
static int need_slots(mg_solver *s, int n) {
    if (n <= s->d_norms_cap) return 0;
    if (s->d_norms) mgk_free(s->ctx, s->d_norms);
    void *q = NULL;
    s->d_norms = NULL; s->d_norms_cap = 0;
    CHK(mgk_malloc(s->ctx, &q, sizeof(double) * (size_t)n));
    s->d_norms = (double *)q; s->d_norms_cap = n;
    return 0;
}

/* MatMult(res[l-1], r[l-1], b[l]) (src/solver.c:1535) */
static int restrict_to(mg_solver *s, int P, int l) {
    mg_level *Lf = &s->L[l - 1], *Lc = &s->L[l];
    mg_fset *F = &Lf->f[P], *Cq = &Lc->f[P];
    const mg_ops *O = &OPS[P];
    Cq->b_ghost_ok = 0; Cq->bfar_ok = 0;
    CHK(halo(s, P, Lf, F->rv));
    if (Lf->distributed && !Lc->distributed) {
        /* slab -> replicated: produce my coarse planes in place, then all-gather them */
        mgk_geom gc = Cq->g;
        int c0 = s->zstart[s->cfg.rank], c1 = s->zstart[s->cfg.rank + 1];
        gc.nz = c1 - c0;
        CHK(O->restrict_fw(s->ctx, &F->g, &gc, F->rv, (char *)Cq->b + (size_t)O->esz * (size_t)c0 * (size_t)Cq->g.plane, NULL));
        void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
        CHK(mgk_stream_wait(s->ctx, ms, cs));
        CHK(s->comm->allgather_planes(s->comm, s->ctx, Cq->b, &Cq->g, s->zstart, O->esz, ms));
        CHK(mgk_stream_wait(s->ctx, cs, ms));
        return 0;
    }
    CHK(O->restrict_fw(s->ctx, &F->g, &Cq->g, F->rv, Cq->b, NULL));
    return 0;
}

/* MatMult(pro[l],u[l+1],rv[l]); VecAXPY(u[l],1.0,rv[l]) (src/solver.c:1540-1541) */
static int prolong_from(mg_solver *s, int P, int l) {
    mg_level *Lf = &s->L[l], *Lc = &s->L[l + 1];
    mg_fset *F = &Lf->f[P], *Cq = &Lc->f[P];
    const mg_ops *O = &OPS[P];
    if (F->u_ghost_pending) CHK(ensure_u_ghosts(s, P, Lf));
    if (Lf->distributed && !Lc->distributed) {
        mgk_geom gc = Cq->g;
        int c0 = s->zstart[s->cfg.rank], c1 = s->zstart[s->cfg.rank + 1];
        gc.nz = c1 - c0;
        CHK(O->prolong_add(s->ctx, &F->g, &gc, (char *)Cq->u + (size_t)O->esz * (size_t)c0 * (size_t)Cq->g.plane, F->u, NULL));
    } else {
        CHK(ensure_u_ghosts(s, P, Lc));
        CHK(O->prolong_add(s->ctx, &F->g, &Cq->g, Cq->u, F->u, NULL));
    }
    F->u_ghost_ok = 0;
    return 0;
}

/* fuse bit 12 (fp64, 3-D, level 0 of 1023-wide whole grids, v0 = 3): post-smoothing = ONE pass for the prolongation and two sweeps; the third
 * sweep is the first stage of the two-sweep pass that evaluates the norm (mgk_jacobi2_sumsq_mid_f64), whose second stage is the first
 * pre-smoothing sweep of the next cycle: 25 + 24 + 24 (two more pre-sweeps) + 18 (restriction) = 91 B per fine unknown and cycle.  The
 * iterate the norm belongs to is never stored: if the iteration stops, finalize_iterate() makes that one sweep. */
static int pjp_ok(const mg_solver *s, int P) {
    const mg_level *L = &s->L[0];
    const int need = 4096 | 1024 | 32 | 8 | 2 | 1;
    if ((s->cfg.fuse & need) != need || P != 0 || s->cfg.dim != 3 || s->cfg.mesh || s->cfg.ksp_type != MG_KSP_RICHARDSON) return 0;
    if (s->cfg.v[0] != 3 || s->levels < 2 || s->lgraph == 1 || L->n < s->cfg.pair_min_n || s->last_cycle) return 0;
    if (L->distributed) {
        /* (round 3, second session; fuse bit 14) the same three passes on z-slabs: the prolongation pass needs the neighbours' boundary AND second
         * planes of u (before the correction) and of the coarse u -- the far fields of the two-sweep passes of both levels carry them */
        const mg_level *Lc = &s->L[1];
        const int hi = s->cfg.rank < s->cfg.nranks - 1;
        if (!(s->cfg.fuse & 16384) || !Lc->distributed || !L->f[0].far || !Lc->f[0].far || L->nz_min < 8 || Lc->nz_min < 4) return 0;
        return mgk_prolong_jacobi2_slab_ok_f64(&L->f[0].g, &Lc->f[0].g, hi) && mgk_jacobi2_sumsq_ok_f64(&L->f[0].g);
    }
    return mgk_prolong_jacobi2_ok_f64(&L->f[0].g, &s->L[1].f[0].g) && mgk_jacobi2_sumsq_ok_f64(&L->f[0].g);
}
static int finalize_iterate(mg_solver *s) {
    if (!s->iterate_behind) return 0;
    mg_level *L = &s->L[0];
    mg_fset *F = &L->f[0];
    if (L->distributed) {
        CHK(ensure_u_ghosts(s, 0, L));
        CHK(mgk_jacobi_range_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u, (double *)F->tmp, 0, F->g.nz, NULL));
    } else
    CHK(mgk_jacobi_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u, (double *)F->tmp, NULL));
    swap_ptr(&F->u, &F->tmp);
    F->u_ghost_ok = 0; F->u_ghost_pending = 0;
    s->iterate_behind = 0; s->spec_valid = 0;
    return 0;
}

/* prolongation fused into the first post-smoothing sweep: u <- Jacobi(u + P u_c)  (src/solver.c:1540-1542) */
static int prolong_smooth(mg_solver *s, int P, int l) {
    mg_level *Lf = &s->L[l], *Lc = &s->L[l + 1];
    mg_fset *F = &Lf->f[P], *Cq = &Lc->f[P];
    const mg_ops *O = &OPS[P];
    const int v0 = s->cfg.v[0];
    if (!(s->cfg.fuse & 2) || s->cfg.ksp_type != MG_KSP_RICHARDSON || v0 < 1) {
        CHK(prolong_from(s, P, l));
        return smooth(s, P, l, v0, 0);
    }
    if (j3_2d_ok(s, P, l, v0)) {
        /* 2-D: the prolongation, the correction and the first THREE post-smoothing sweeps in one pass */
        const int mesh = s->cfg.mesh != 0;
        CHK(mgk_prolong_jacobi3_2d_f64(s->ctx, &F->g, &Cq->g, mesh ? NULL : Lf->coef, mesh ? 1.0 : Lf->dinv, s->cfg.scale, mesh ? Lf->ctab : NULL,
                                       mesh ? Lf->dtab : NULL, (const double *)F->b, (const double *)Cq->u, (const double *)F->u, (double *)F->tmp, NULL));
        swap_ptr(&F->u, &F->tmp);
        F->u_ghost_ok = 0; F->u_ghost_pending = 0;
        return v0 > 3 ? smooth(s, P, l, v0 - 3, 0) : 0;
    }
    if (l == 0 && pjp_ok(s, P) && Lf->distributed) {
        /* ... on a z-slab: TWO grouped exchanges on the comm stream (the coarse u's ghost planes + the neighbours' second coarse planes; u's ghost planes,
         * the neighbours' second planes of u, b's ghost planes once per cycle) travel while the output planes that read none of it are produced:
         * plane z reads u on z-2 .. z+2 and their parents -- with a rank below from z = 4 on (plane 2's parents are the own coarse planes 0, 1),
         * with a rank above up to nz-3 (plane nz-1's parent is the own coarse plane nzc-1) */
        const int nz = F->g.nz, lo = s->cfg.rank > 0, hi = s->cfg.rank < s->cfg.nranks - 1;
        void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
        {   /* (the coarse b is done with for this cycle: its ghost planes stay as they are -- two planes less on the wire) */
            const int cb = Cq->b_ghost_ok;
            Cq->b_ghost_ok = 1;
            int rcx = group_exchange(s, P, Cq, 0);
            Cq->b_ghost_ok = cb;
            CHK(rcx);
        }
        CHK(group_exchange(s, P, F, 0));
        const int zi0 = lo ? 4 : 0, zi1 = hi ? nz - 2 : nz;
        const int split = s->cfg.overlap && zi1 - zi0 >= 2;
#define PJ2S(z0, z1) mgk_prolong_jacobi2_slab_f64(s->ctx, &F->g, &Cq->g, &F->gfar, &Cq->gfar, Lf->coef, Lf->dinv, s->cfg.scale, (const double *)F->b, \
                        (const double *)Cq->u, (const double *)F->u, (double *)F->tmp, (const double *)F->far, (const double *)Cq->far, lo, hi, z0, z1, cs)
        if (split) CHK(PJ2S(zi0, zi1));
        CHK(mgk_stream_wait(s->ctx, cs, ms));
        Cq->u_ghost_pending = 0; Cq->u_ghost_ok = 1;
        F->u_ghost_pending = 0; F->u_ghost_ok = 1; F->b_ghost_ok = 1;
        if (split) {
            if (lo) CHK(PJ2S(0, 4));
            if (hi) CHK(PJ2S(nz - 2, nz));
        } else CHK(PJ2S(0, nz));
#undef PJ2S
        swap_ptr(&F->u, &F->tmp);
        F->u_ghost_ok = 0; F->u_ghost_pending = 0;
        s->sweep_owed = 1;
        return 0;
    }
    if (l == 0 && pjp_ok(s, P)) {
        /* the prolongation and the first TWO post-smoothing sweeps in one pass; the third one is made by the pass that evaluates the
         * norm (vcycle_once), which reads this field anyway */
        CHK(mgk_prolong_jacobi2_f64(s->ctx, &F->g, &Cq->g, Lf->coef, Lf->dinv, s->cfg.scale, (const double *)F->b, (const double *)Cq->u,
                                    (const double *)F->u, (double *)F->tmp, NULL));
        swap_ptr(&F->u, &F->tmp);
        F->u_ghost_ok = 0; F->u_ghost_pending = 0;
        s->sweep_owed = 1;
        return 0;
    }
    mgk_geom gc = Cq->g;
    const void *ucoarse = Cq->u;
    if (Lf->distributed && !Lc->distributed) {
        int c0 = s->zstart[s->cfg.rank], c1 = s->zstart[s->cfg.rank + 1];
        gc.nz = c1 - c0;
        ucoarse = (const char *)Cq->u + (size_t)O->esz * (size_t)c0 * (size_t)Cq->g.plane;
    } else if (Lc->distributed) {
        CHK(begin_u_ghosts(s, P, Lc));       /* the coarse u's ghost planes start travelling now */
    }
    /* not counted by the profile: that one times the plain sweep kernel (bench.py roofline leg) */
    if (Lf->distributed && s->cfg.overlap && Lf->nz_min >= 4 && s->cfg.dim == 3) {
        /* slab: the output planes 2 .. nz-2 read neither a ghost plane of u (the neighbours' boundary planes BEFORE the
         * correction) nor one of the coarse u (parents of the fine planes -1, 0 and nz): they are swept while those travel */
        const int nz = F->g.nz;
        CHK(begin_u_ghosts(s, P, Lf));
        CHK(O->prolong_jacobi_range(s->ctx, &F->g, &gc, Lf->coef, Lf->dinv, s->cfg.scale, F->b, ucoarse, F->u, F->tmp, 2, nz - 1, NULL));
        if (Lc->distributed) CHK(ensure_u_ghosts(s, P, Lc));
        CHK(ensure_u_ghosts(s, P, Lf));
        CHK(O->prolong_jacobi_range(s->ctx, &F->g, &gc, Lf->coef, Lf->dinv, s->cfg.scale, F->b, ucoarse, F->u, F->tmp, 0, 2, NULL));
        CHK(O->prolong_jacobi_range(s->ctx, &F->g, &gc, Lf->coef, Lf->dinv, s->cfg.scale, F->b, ucoarse, F->u, F->tmp, nz - 1, nz, NULL));
    } else {
        if (Lf->distributed && Lc->distributed) CHK(ensure_u_ghosts(s, P, Lc));
        CHK(ensure_u_ghosts(s, P, Lf));          /* the neighbours' boundary planes BEFORE the correction */
        if (s->cfg.mesh) CHK(mgk_prolong_jacobi_rowcoef_f64(s->ctx, &F->g, &gc, Lf->ctab, Lf->dtab, s->cfg.scale, (const double *)F->b,
                                                            (const double *)ucoarse, (const double *)F->u, (double *)F->tmp, NULL));
        else CHK(O->prolong_jacobi(s->ctx, &F->g, &gc, Lf->coef, Lf->dinv, s->cfg.scale, F->b, ucoarse, F->u, F->tmp, NULL));
    }
    swap_ptr(&F->u, &F->tmp);
    F->u_ghost_ok = 0; F->u_ghost_pending = 0;
    return smooth(s, P, l, v0 - 1, 0);
}

static int descend_restrict(mg_solver *s, int P, int l, int no_jz);
/* the levels ltail .. L-1 of one cycle in ONE kernel (their fields live in LDS): b of level ltail is there, u of level ltail
 * comes back post-smoothed (src/solver.c:1536-1543 for those levels) */
static int tail(mg_solver *s, int P) {
    const int lt = s->ltail, nl = s->levels - lt;
    int n[8];
    double k7[8 * 7], di[8];
    for (int q = 0; q < nl; q++) {
        const mg_level *L = &s->L[lt + q];
        n[q] = L->n; di[q] = L->dinv;
        for (int e = 0; e < 7; e++) k7[7 * q + e] = L->coef[e];
    }
    mg_fset *F = &s->L[lt].f[P];
    if (s->cfg.mesh) {
        const double *ct[8], *dt[8];
        for (int q = 0; q < nl; q++) { ct[q] = s->L[lt + q].ctab; dt[q] = s->L[lt + q].dtab; }
        CHK(mgk_tail_cycle_rowcoef_f64(s->ctx, &F->g, nl, n, ct, dt, s->cfg.scale, s->cfg.v[0], s->cfg.v[1], (const double *)F->b,
                                       (double *)F->u, NULL));
    } else
    CHK(OPS[P].tail_cycle(s->ctx, &F->g, nl, n, k7, di, s->cfg.scale, s->cfg.v[0], s->cfg.v[1], F->b, F->u, NULL));
    F->jz_ready = 0;
    return 0;
}

/* one step of the descent: b_l = R(b_{l-1} - A u_{l-1}); smooth level l from a zero guess (src/solver.c:1534-1537) */
static int descend(mg_solver *s, int P, int l) {
    const int levels = s->levels, *v = s->cfg.v;
    if (s->ltail && l == s->ltail) {              /* the restriction feeds the tail kernel, which smooths this level and all below */
        CHK(descend_restrict(s, P, l, 1));
        return tail(s, P);
    }
    CHK(descend_restrict(s, P, l, 0));
    CHK(smooth(s, P, l, l == levels - 1 ? v[1] : v[0], l != levels - 1));   /* :1536 */
    if (l != levels - 1) s->L[l].f[P].guess_nonzero = 1;                /* :1537 */
    return 0;
}

/* b_l = R(b_{l-1} - A u_{l-1})  (src/solver.c:1534-1535) */
static int descend_restrict(mg_solver *s, int P, int l, int no_jz) {
    const int levels = s->levels, *v = s->cfg.v;
    mg_level *Lf = &s->L[l - 1];
    const mg_ops *O = &OPS[P];
    if (Lf->f[P].last_sweep_pending && Lf->distributed) {
        /* the same pass on a z-slab.  ONE grouped exchange brings everything the slab needs to sweep one plane below and two planes
         * above itself and to evaluate the residual of the plane above: u's and b's ghost planes, the neighbours' second planes of u
         * (far), the upper neighbour's third plane of u (far2) and second plane of b (bfar; b of a level is rewritten only by the
         * restriction above it).  It travels while the coarse planes that read none of it are produced. */
        mg_level *Lc = &s->L[l];
        mg_fset *F = &Lf->f[P], *Cq = &Lc->f[P];
        void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
        const int me = s->cfg.rank, lo = me > 0, hi = me < s->cfg.nranks - 1;
        mgk_geom gc = Cq->g;
        void *bc = Cq->b;
        Cq->b_ghost_ok = 0; Cq->bfar_ok = 0;
        if (!Lc->distributed) {                 /* slab -> replicated: my coarse planes land in place, then all-gather */
            const int c0 = s->zstart[me], c1 = s->zstart[me + 1];
            gc.nz = c1 - c0;
            bc = (char *)Cq->b + (size_t)O->esz * (size_t)c0 * (size_t)Cq->g.plane;
        }
        const int nzc = gc.nz;
        CHK(group_exchange(s, P, F, 1));
        /* coarse planes 1 .. nzc-3 read the fine planes 0 .. nz-2 of u only */
        const int split = s->cfg.overlap && nzc >= 5 && Lf->nz_min >= 10;
#define SRRS(k0, k1) mgk_sweep_residual_restrict_slab_f64(s->ctx, &F->g, &gc, &F->gfar, Lf->coef, Lf->dinv, s->cfg.scale, (const double *)F->b, \
                        (const double *)F->u, (double *)F->tmp, (const double *)F->far, (const double *)F->far2, (const double *)F->bfar, lo, hi, (double *)bc, k0, k1, cs)
        if (split) CHK(SRRS(1, nzc - 2));
        CHK(mgk_stream_wait(s->ctx, cs, ms));
        if (split) { CHK(SRRS(0, 1)); CHK(SRRS(nzc - 2, nzc)); }
        else CHK(SRRS(0, nzc));
#undef SRRS
        swap_ptr(&F->u, &F->tmp);
        F->u_ghost_ok = 0; F->u_ghost_pending = 0; F->b_ghost_ok = 1; F->bfar_ok = 1; F->last_sweep_pending = 0;
        if (!Lc->distributed) {
            CHK(mgk_stream_wait(s->ctx, ms, cs));
            CHK(s->comm->allgather_planes(s->comm, s->ctx, Cq->b, &Cq->g, s->zstart, O->esz, ms));
            CHK(mgk_stream_wait(s->ctx, cs, ms));
        }
        return 0;
    }
    if (Lf->f[P].last_sweep_pending) {
        /* the last pre-smoothing sweep, the residual and its restriction in one pass (:1531 / :1536 last iteration, :1534-1535) */
        mg_fset *F = &Lf->f[P], *Cq = &s->L[l].f[P];
        const int sweeps = (l == levels - 1) ? v[1] : v[0];
        const int jz = (s->cfg.fuse & 256) && !no_jz && sweeps >= 1 && !Cq->guess_nonzero && !triple_ok(s, P, l, sweeps);
        if (s->cfg.mesh)
            CHK(mgk_sweep_residual_restrict_2d_rowcoef_f64(s->ctx, &F->g, &Cq->g, Lf->ctab, Lf->dtab, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                                           (double *)F->tmp, (double *)Cq->b, jz ? (double *)Cq->tmp : NULL, s->L[l].dtab, s->cfg.scale, NULL));
        else if (s->cfg.dim == 2)
            CHK(mgk_sweep_residual_restrict_2d_f64(s->ctx, &F->g, &Cq->g, Lf->coef, Lf->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                                   (double *)F->tmp, (double *)Cq->b, jz ? (double *)Cq->tmp : NULL, s->L[l].dinv, s->cfg.scale, NULL));
        else
        CHK(mgk_sweep_residual_restrict_f64(s->ctx, &F->g, &Cq->g, Lf->coef, Lf->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                            (double *)F->tmp, (double *)Cq->b, jz ? (double *)Cq->tmp : NULL, s->L[l].dinv, s->cfg.scale, NULL));
        swap_ptr(&F->u, &F->tmp);
        F->u_ghost_ok = 0; F->u_ghost_pending = 0; F->last_sweep_pending = 0;
        Cq->b_ghost_ok = 0; Cq->bfar_ok = 0;
        if (jz) Cq->jz_ready = 1;
        return 0;
    }
    /* (below 255^3 the marching fused kernel is latency bound: residual + restriction as two short kernels are quicker) */
    if ((s->cfg.fuse & 4) && O->residual_restrict && s->cfg.dim == 3 && !Lf->distributed && Lf->n + 1 <= 1024 &&
        (Lf->n >= 255 || (s->cfg.fuse & 128))) {
        /* :1534-1535 in one pass: b_l = R (b - A u), the fine residual is never written */
        mg_fset *Cq = &s->L[l].f[P];
        const int sweeps = (l == levels - 1) ? v[1] : v[0];
        if ((s->cfg.fuse & 256) && !no_jz && s->cfg.ksp_type == MG_KSP_RICHARDSON && sweeps >= 1 && !Cq->guess_nonzero && !triple_ok(s, P, l, sweeps)) {
            /* ... and the coarse level's first sweep from its zero guess comes out of the same kernel (saves re-reading b_l) */
            CHK(O->residual_restrict_jz(s->ctx, &Lf->f[P].g, &Cq->g, Lf->coef, Lf->f[P].b, Lf->f[P].u, Cq->b, Cq->tmp, s->L[l].dinv,
                                        s->cfg.scale, NULL));
            Cq->jz_ready = 1;
        } else {
            CHK(O->residual_restrict(s->ctx, &Lf->f[P].g, &Cq->g, Lf->coef, Lf->f[P].b, Lf->f[P].u, Cq->b, NULL));
        }
    } else if ((s->cfg.fuse & 4) && O->residual_restrict && s->cfg.dim == 3 && Lf->distributed && Lf->n + 1 <= 1024 &&
               Lf->nz_min >= 2) {
        /* the same on a z-slab.  The last coarse plane of every rank but the last needs the residual of the NEXT rank's
         * first plane: each rank evaluates that one plane first and ships it (comm stream) while the fused kernel
         * runs over the slab and leaves its last coarse plane partial; a small kernel then appends the missing terms in
         * the order of the row of res. */
        mg_level *Lc = &s->L[l];
        mg_fset *F = &Lf->f[P], *Cq = &Lc->f[P];
        void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
        const int me = s->cfg.rank, last = (me == s->cfg.nranks - 1);
        Cq->b_ghost_ok = 0; Cq->bfar_ok = 0;
        mgk_geom gc = Cq->g;
        void *bc = Cq->b;
        if (!Lc->distributed) {                 /* slab -> replicated: my coarse planes land in place, then all-gather */
            const int c0 = s->zstart[me], c1 = s->zstart[me + 1];
            gc.nz = c1 - c0;
            bc = (char *)Cq->b + (size_t)O->esz * (size_t)c0 * (size_t)Cq->g.plane;
        }
        if (F->far && Lf->nz_min >= 4) {
            /* ONE exchange: with u's and b's ghost planes and the neighbours' second planes of u (the far field of the two-sweep
             * passes) every rank evaluates the residual of the plane above its slab itself and completes its last coarse plane.
             * The exchange travels while the coarse planes that read no ghost plane are restricted. */
            const int nzc = gc.nz, hi = !last;
            CHK(group_exchange(s, P, F, 0));
            const int split = s->cfg.overlap && nzc >= 3 && Lf->nz_min >= 6;
            if (split) CHK(O->residual_restrict_slab(s->ctx, &F->g, &gc, &F->gfar, Lf->coef, F->b, F->u, F->far, hi, bc, 1, nzc - 1, cs));
            CHK(mgk_stream_wait(s->ctx, cs, ms));
            F->u_ghost_pending = 0; F->u_ghost_ok = 1; F->b_ghost_ok = 1;
            if (split) {
                CHK(O->residual_restrict_slab(s->ctx, &F->g, &gc, &F->gfar, Lf->coef, F->b, F->u, F->far, hi, bc, 0, 1, cs));
                CHK(O->residual_restrict_slab(s->ctx, &F->g, &gc, &F->gfar, Lf->coef, F->b, F->u, F->far, hi, bc, nzc - 1, nzc, cs));
            } else {
                CHK(O->residual_restrict_slab(s->ctx, &F->g, &gc, &F->gfar, Lf->coef, F->b, F->u, F->far, hi, bc, 0, nzc, cs));
            }
            goto gathered;
        } else if (s->cfg.overlap && Lf->nz_min >= 8) {
            /* (without the far field: fuse bit 5 off) two dependent exchanges (u's ghost planes, then the residual of plane 0,
             * which needs them) each hidden behind half of the coarse planes that read no ghost plane */
            const int nzc = gc.nz, kmid = nzc / 2;
            CHK(begin_u_ghosts(s, P, Lf));
            CHK(O->residual_restrict_range(s->ctx, &F->g, &gc, Lf->coef, F->b, F->u, bc, 1, kmid, cs));
            CHK(ensure_u_ghosts(s, P, Lf));
            CHK(O->residual_range(s->ctx, &F->g, Lf->coef, F->b, F->u, F->rv, 0, 1, cs));
            CHK(mgk_stream_wait(s->ctx, ms, cs));
            CHK(s->comm->halo(s->comm, s->ctx, F->rv, &F->g, O->esz, ms));
            CHK(O->residual_restrict_range(s->ctx, &F->g, &gc, Lf->coef, F->b, F->u, bc, kmid, nzc - 1, cs));
            CHK(O->residual_restrict_range(s->ctx, &F->g, &gc, Lf->coef, F->b, F->u, bc, 0, 1, cs));
            CHK(O->residual_restrict_range(s->ctx, &F->g, &gc, Lf->coef, F->b, F->u, bc, nzc - 1, nzc, cs));
        } else {
            CHK(ensure_u_ghosts(s, P, Lf));
            CHK(O->residual_range(s->ctx, &F->g, Lf->coef, F->b, F->u, F->rv, 0, 1, cs));
            CHK(mgk_stream_wait(s->ctx, ms, cs));
            CHK(s->comm->halo(s->comm, s->ctx, F->rv, &F->g, O->esz, ms));
            CHK(O->residual_restrict(s->ctx, &F->g, &gc, Lf->coef, F->b, F->u, bc, cs));
        }
        CHK(mgk_stream_wait(s->ctx, cs, ms));
        if (!last) CHK(O->restrict_finish(s->ctx, &F->g, &gc, F->rv, bc, cs));
gathered:
        if (!Lc->distributed) {
            CHK(mgk_stream_wait(s->ctx, ms, cs));
            CHK(s->comm->allgather_planes(s->comm, s->ctx, Cq->b, &Cq->g, s->zstart, O->esz, ms));
            CHK(mgk_stream_wait(s->ctx, cs, ms));
        }
    } else if ((s->cfg.fuse & 4) && s->cfg.dim == 2 && P == 0 && Lf->n >= 127) {
        /* 2-D: the same fusion, one independent wave per tile (mgk_residual_restrict_2d_f64), with the coarse level's first
         * zero-guess sweep when that level is smoothed by its own launches */
        mg_fset *Cq = &s->L[l].f[P];
        const int sweeps = (l == levels - 1) ? v[1] : v[0];
        const int jz = (s->cfg.fuse & 256) && !no_jz && s->cfg.ksp_type == MG_KSP_RICHARDSON && sweeps >= 1 && !Cq->guess_nonzero && !triple_ok(s, P, l, sweeps);
        if (s->cfg.mesh) CHK(mgk_residual_restrict_2d_rowcoef_f64(s->ctx, &Lf->f[P].g, &Cq->g, Lf->ctab, (const double *)Lf->f[P].b,
                                                                  (const double *)Lf->f[P].u, (double *)Cq->b, jz ? (double *)Cq->tmp : NULL,
                                                                  s->L[l].dtab, s->cfg.scale, NULL));
        else CHK(mgk_residual_restrict_2d_f64(s->ctx, &Lf->f[P].g, &Cq->g, Lf->coef, (const double *)Lf->f[P].b, (const double *)Lf->f[P].u,
                                              (double *)Cq->b, jz ? (double *)Cq->tmp : NULL, s->L[l].dinv, s->cfg.scale, NULL));
        if (jz) Cq->jz_ready = 1;
    } else {
        CHK(residual(s, P, l - 1));                                     /* :1534 */
        CHK(restrict_to(s, P, l));                                      /* :1535 */
    }
    return 0;
}

/* levels lg..L-1: down from lg-1 and back up to lg.  Every buffer pointer is the same at entry of every cycle
 * (each level swaps u/tmp an even number of times per cycle; the coarsest is copied back when v1 is odd), so the
 * recorded kernels stay valid. */
static int coarse_part(mg_solver *s, int P, int lg) {
    const int levels = s->levels, lend = s->ltail ? s->ltail : levels - 1;     /* last level the loops below handle themselves */
    for (int l = lg; l <= lend; l++) CHK(descend(s, P, l));
    if (!s->ltail && (s->cfg.v[1] & 1)) {                               /* restore the coarsest level's buffer identity */
        mg_fset *Cz = &s->L[levels - 1].f[P];
        CHK(mgk_d2d(s->ctx, Cz->tmp, Cz->u, (size_t)OPS[P].esz * (size_t)Cz->g.total, NULL));
        swap_ptr(&Cz->u, &Cz->tmp);
    }
    for (int l = lend - 1; l >= lg; l--) {
        CHK(prolong_smooth(s, P, l));                                   /* :1540-1542 */
        s->L[l].f[P].guess_nonzero = 0;                                 /* :1543 (l != 0 here) */
    }
    return 0;
}

/* the multigrid part of one iteration of the while loop (src/solver.c:1531-1544) in precision P.
 * first: the level-0 KSP has not been switched to a non-zero initial guess yet (:1532) */
static int cycle_body(mg_solver *s, int P, int first) {
    const int levels = s->levels, *v = s->cfg.v;
    const int lg = s->lgraph ? s->lgraph : levels;                      /* levels >= lg run as one HIP graph */
    CHK(smooth(s, P, 0, v[0], levels > 1));                             /* :1531 */
    if (first) s->L[0].f[P].guess_nonzero = 1;                          /* :1532 */
    const int lend = s->ltail ? s->ltail : levels - 1;
    for (int l = 1; l < lg && l <= lend; l++) CHK(descend(s, P, l));
    if (lg < levels && lg <= lend) {
        mg_fset *Fe = &s->L[lg - 1].f[P];
        if (s->coarse_graph[P] && (Fe->u != s->graph_u[P] || Fe->tmp != s->graph_tmp[P])) {
            /* the level that feeds the recording has swapped u / tmp an odd number of times since it was made (sweep groupings that differ
             * on the way down and on the way up, e.g. pairs without the fused prolongation and an even v0): the recorded restriction would
             * read the stale buffer.  Never in a default configuration; correctness first -- record again on the pointers of this cycle */
            CHK(mgk_sync(s->ctx, NULL));                   /* (the last replay may still be running: nothing is destroyed under it) */
            mgk_graph_destroy(s->ctx, s->coarse_graph[P]);
            s->coarse_graph[P] = NULL;
            s->graph_rerecorded++;
        }
        if (!s->coarse_graph[P]) {                                      /* record once ... */
            s->graph_u[P] = Fe->u; s->graph_tmp[P] = Fe->tmp;
            CHK(mgk_capture_begin(s->ctx));
            int rc = coarse_part(s, P, lg);
            void *ge = NULL;
            int rc2 = mgk_capture_end(s->ctx, &ge);
            if (rc || rc2) return mgfail(rc ? rc : rc2, "HIP graph capture of the coarse levels");
            s->coarse_graph[P] = ge;
        }
        CHK(mgk_graph_launch(s->ctx, s->coarse_graph[P]));              /* ... replay every cycle */
    }
    for (int l = ((lg < levels && lg <= lend) ? lg - 1 : lend - 1); l >= 0; l--) {
        CHK(prolong_smooth(s, P, l));                                   /* :1540-1542 */
        if (l != 0) s->L[l].f[P].guess_nonzero = 0;                     /* :1543 */
    }
    return 0;
}

/* body of the while loop, src/solver.c:1531-1549 */
static int vcycle_once(mg_solver *s) {
    mg_level *L = &s->L[0];
    mg_fset *F = &L->f[0];
    double ss;
    if (s->cfg.precision == MG_PREC_MIXED) {
        /* BASELINE config 5: e = Vcycle32((float) r) from e = 0;  u += (double) e;  r = b - A u in fp64, ||r|| */
        mg_fset *E = &L->f[1];
        E->guess_nonzero = 0;
        CHK(cycle_body(s, 1, 1));
        /* the fp32 cycle that follows starts from e = 0: its first sweep, scale * (r32 * dinv), comes out of the same pass that
         * produces r32 (fuse bit 8): one launch and one read of r32 less per outer step */
        const int jz = (s->cfg.fuse & 256) && s->cfg.v[0] >= 1 && s->levels > 1 && !triple_ok(s, 1, 0, s->cfg.v[0]);
        if ((s->cfg.fuse & 16) && F->tmp && !L->distributed) {
            /* u += (double) e and r32 = (float)(b - A u) in one pass (32 B/unknown instead of 20 + 20) */
            if (jz) CHK(mgk_correct_residual_f64_f32_jz(s->ctx, &F->g, &E->g, L->coef, (const double *)F->b, (const double *)F->u,
                                                        (const float *)E->u, (double *)F->tmp, (float *)E->b, (float *)E->tmp, L->dinv,
                                                        s->cfg.scale, &ss, NULL));
            else CHK(mgk_correct_residual_f64_f32(s->ctx, &F->g, &E->g, L->coef, (const double *)F->b, (const double *)F->u,
                                                  (const float *)E->u, (double *)F->tmp, (float *)E->b, &ss, NULL));
            swap_ptr(&F->u, &F->tmp);
            F->u_ghost_ok = 0; F->u_ghost_pending = 0;
        } else {
            CHK(mgk_correct_f64_from_f32(s->ctx, &F->g, &E->g, (const float *)E->u, (double *)F->u, NULL));
            F->u_ghost_ok = 0; F->u_ghost_pending = 0;
            CHK(ensure_u_ghosts(s, 0, L));
            if (jz) CHK(mgk_residual_f64_to_f32_jz(s->ctx, &F->g, &E->g, L->coef, (const double *)F->b, (const double *)F->u, (float *)E->b,
                                                   (float *)E->tmp, L->dinv, s->cfg.scale, &ss, NULL));
            else CHK(mgk_residual_f64_to_f32(s->ctx, &F->g, &E->g, L->coef, (const double *)F->b, (const double *)F->u, (float *)E->b, &ss, NULL));
        }
        E->jz_ready = jz;
        E->b_ghost_ok = 0;
    } else {
        CHK(cycle_body(s, 0, s->iter == 0));
        /* :1545-1546  r0 = b0 - A0 u0 ; ||r0|| */
        const int jnorm = (s->cfg.fuse & 8) && (s->cfg.fuse & 1) && s->cfg.ksp_type == MG_KSP_RICHARDSON && s->cfg.v[0] >= 1 && !s->last_cycle;
        if (s->sweep_owed && L->distributed) {
            /* ... on a z-slab: the planes 2 .. nz-3 while the grouped exchange (u's ghosts, the far planes; b's are valid) travels, one reduction
             * over the block partials of the three launches */
            const int nz = F->g.nz, lo = s->cfg.rank > 0, hi = s->cfg.rank < s->cfg.nranks - 1;
            void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
            int n1 = 0, n2 = 0, n3 = 0;
            CHK(group_exchange(s, 0, F, 0));
#define J2M(z0, z1, off, np) mgk_jacobi2_sumsq_mid_slab_f64(s->ctx, &F->g, &F->gfar, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u, \
                                (double *)F->tmp, (const double *)F->far, lo, hi, z0, z1, off, np, cs)
            const int split = s->cfg.overlap && nz >= 6;
            if (split) {
                s->prof_kind = 1;
                void *t = prof_begin(s, 0);
                s->prof_kind = 0;
                int rc2 = J2M(2, nz - 2, 0, &n1);
                prof_end(s, t);
                CHK(rc2);
            }
            CHK(mgk_stream_wait(s->ctx, cs, ms));
            F->u_ghost_pending = 0; F->u_ghost_ok = 1; F->b_ghost_ok = 1;
            if (split) { CHK(J2M(0, 2, n1, &n2)); CHK(J2M(nz - 2, nz, n1 + n2, &n3)); }
            else CHK(J2M(0, nz, 0, &n1));
#undef J2M
            CHK(mgk_partials_finish(s->ctx, n1 + n2 + n3, &ss, NULL));
            s->sweep_owed = 0; s->iterate_behind = 1; s->spec_valid = 1;
            goto norm_done;
        }
        if (s->sweep_owed) {
            /* the third post-smoothing sweep, the norm of ITS result and the first pre-smoothing sweep of the next cycle in one pass;
             * u stays one sweep behind the iterate, tmp is one sweep ahead of it */
            s->prof_kind = 1;
            void *t = prof_begin(s, 0);
            s->prof_kind = 0;
            int rc2 = mgk_jacobi2_sumsq_mid_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                                (double *)F->tmp, &ss, NULL);
            prof_end(s, t);
            CHK(rc2);
            s->sweep_owed = 0; s->iterate_behind = 1; s->spec_valid = 1;
            goto norm_done;
        }
        if (jnorm && L->distributed && s->cfg.dim == 3 && (s->cfg.fuse & 1024) && (s->cfg.fuse & 32) && F->far && L->nz_min >= 6 &&
            s->cfg.v[0] >= 2 && L->n >= s->cfg.pair_min_n && L->n + 1 <= 1024 && s->lgraph != 1 && mgk_jacobi2_sumsq_ok_f64(&F->g)) {
            /* slab: the norm and the first TWO sweeps of the next cycle in one pass (the two-sweep slab pass with the norm of its
             * input's residual): the planes 2 .. nz-3 while the grouped exchange (u's ghosts, b's once, the far planes) travels */
            const int nz = F->g.nz, lo = s->cfg.rank > 0, hi = s->cfg.rank < s->cfg.nranks - 1;
            void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
            int n1 = 0, n2 = 0, n3 = 0;
            CHK(group_exchange(s, 0, F, 0));
#define J2N(z0, z1, off, np) mgk_jacobi2_sumsq_slab_f64(s->ctx, &F->g, &F->gfar, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u, \
                                (double *)F->tmp, (const double *)F->far, lo, hi, z0, z1, off, np, cs)
            const int split = s->cfg.overlap && nz >= 6;
            if (split) CHK(J2N(2, nz - 2, 0, &n1));
            CHK(mgk_stream_wait(s->ctx, cs, ms));
            F->u_ghost_pending = 0; F->u_ghost_ok = 1; F->b_ghost_ok = 1;
            if (split) { CHK(J2N(0, 2, n1, &n2)); CHK(J2N(nz - 2, nz, n1 + n2, &n3)); }
            else CHK(J2N(0, nz, 0, &n1));
#undef J2N
            CHK(mgk_partials_finish(s->ctx, n1 + n2 + n3, &ss, NULL));
            s->spec_valid = 2;
            goto norm_done;
        }
        if (jnorm && L->distributed && s->cfg.overlap && L->nz_min >= 3 && s->cfg.dim == 3) {
            /* the same on a slab with the exchange of u's ghost planes hidden: inner planes first, the two boundary planes
             * once the ghosts have arrived; one reduction over the block partials of the three launches */
            const int nz = F->g.nz;
            int n1 = 0, n2 = 0, n3 = 0;
            CHK(begin_u_ghosts(s, 0, L));
            CHK(mgk_jacobi_sumsq_range_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                           (double *)F->tmp, 1, nz - 1, 0, &n1, NULL));
            CHK(ensure_u_ghosts(s, 0, L));
            CHK(mgk_jacobi_sumsq_range_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                           (double *)F->tmp, 0, 1, n1, &n2, NULL));
            CHK(mgk_jacobi_sumsq_range_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                           (double *)F->tmp, nz - 1, nz, n1 + n2, &n3, NULL));
            CHK(mgk_partials_finish(s->ctx, n1 + n2 + n3, &ss, NULL));
            s->spec_valid = 1;
            goto norm_done;
        }
        CHK(ensure_u_ghosts(s, 0, L));
        if (jnorm) {
            /* ||b - A u|| and, speculatively, the first pre-smoothing sweep of the next cycle in one pass over u and b
             * (both form the same residual).  The sweep lands in tmp and is adopted by smooth() only if another
             * cycle follows; u itself is untouched, so stopping here leaves the solution as the reference has it. */
            /* (not when level 0 feeds the coarse-level graph: adopting two sweeps at once changes how often level 0 swaps u / tmp in
             * a cycle from the recording cycle's count, and the recorded restriction reads level 0's buffers) */
            const int two = (s->cfg.fuse & 1024) && (s->cfg.fuse & 32) && !L->distributed &&
                            s->cfg.v[0] >= 2 && L->n >= s->cfg.pair_min_n && s->lgraph != 1 && (s->cfg.dim == 2 || mgk_jacobi2_sumsq_ok_f64(&F->g));
            if (j3_2d_ok(s, 0, 0, s->cfg.v[0])) {
                /* 2-D: ... and ALL THREE pre-smoothing sweeps of the next cycle */
                const int mesh = s->cfg.mesh != 0;
                s->prof_kind = 1;
                void *t = prof_begin(s, 0);
                s->prof_kind = 0;
                int rc2 = mgk_jacobi3_2d_sumsq_f64(s->ctx, &F->g, mesh ? NULL : L->coef, mesh ? 1.0 : L->dinv, s->cfg.scale, mesh ? L->ctab : NULL,
                                                   mesh ? L->dtab : NULL, (const double *)F->b, (const double *)F->u, (double *)F->tmp, &ss, NULL);
                prof_end(s, t);
                CHK(rc2);
                s->spec_valid = 3;
                goto norm_done;
            }
            if (s->cfg.mesh && two) CHK(mgk_jacobi2_2d_sumsq_rowcoef_f64(s->ctx, &F->g, L->ctab, L->dtab, s->cfg.scale, (const double *)F->b,
                                                                         (const double *)F->u, (double *)F->tmp, &ss, NULL));
            else if (s->cfg.mesh) CHK(mgk_jacobi_sumsq_rowcoef_f64(s->ctx, &F->g, L->ctab, L->dtab, s->cfg.scale, (const double *)F->b,
                                                                   (const double *)F->u, (double *)F->tmp, &ss, NULL));
            else if (two && s->cfg.dim == 2) CHK(mgk_jacobi2_2d_sumsq_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b,
                                                                          (const double *)F->u, (double *)F->tmp, &ss, NULL));
            else if (two) {                                                             /* ... and the second one: two sweeps in the pass */
                s->prof_kind = 1;                   /* timed with the two-sweep launches (same kernel + the per-block partial sums) */
                void *t = prof_begin(s, 0);
                s->prof_kind = 0;
                int rc2 = mgk_jacobi2_sumsq_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                                (double *)F->tmp, &ss, NULL);
                prof_end(s, t);
                CHK(rc2);
            }
            else CHK(mgk_jacobi_sumsq_f64(s->ctx, &F->g, L->coef, L->dinv, s->cfg.scale, (const double *)F->b, (const double *)F->u,
                                          (double *)F->tmp, &ss, NULL));
            s->spec_valid = two ? 2 : 1;
        } else if ((s->cfg.fuse & 1) && s->cfg.mesh)
            CHK(mgk_residual_sumsq_rowcoef_f64(s->ctx, &F->g, L->ctab, (const double *)F->b, (const double *)F->u, &ss, NULL));
        else if (s->cfg.fuse & 1) CHK(mgk_residual_sumsq_f64(s->ctx, &F->g, L->coef, (const double *)F->b, (const double *)F->u, &ss, NULL));
        else {
            CHK(residual(s, 0, 0));
            CHK(mgk_sumsq_f64(s->ctx, &F->g, (const double *)F->rv, &ss, NULL));
        }
    }
norm_done:
    if (!s->deferring) CHK(norm_from_sumsq(s, ss, &s->rchk));
    s->iter++;
    if (s->iter < s->rnorm_cap) s->rnorm[s->iter] = s->rchk;            /* :1549 */
    return 0;
}

/* src/solver.c:1512-1523 */
static int start(mg_solver *s) {
    mg_level *L = &s->L[0];
    mg_fset *F = &L->f[0];
    double ss;
    CHK(mgk_sumsq_f64(s->ctx, &F->g, (const double *)F->b, &ss, NULL)); /* VecNorm(b[0]) :1512 */
    CHK(norm_from_sumsq(s, ss, &s->bnorm));
    for (int l = 0; l < s->levels; l++)
        for (int p = 0; p < 2; p++) { s->L[l].f[p].guess_nonzero = 0; s->L[l].f[p].u_ghost_ok = 0; s->L[l].f[p].u_ghost_pending = 0; s->L[l].f[p].b_ghost_ok = 0; s->L[l].f[p].jz_ready = 0; s->L[l].f[p].last_sweep_pending = 0; s->L[l].f[p].bfar_ok = 0; }
    CHK(mgk_memset0(s->ctx, F->u, sizeof(double) * (size_t)F->g.total, NULL));   /* VecSet(u[0],0) :1514 */
    /* rv = A u - b with u = 0 (:1516-1517); ||A u - b|| = ||b - A u||, evaluated by the same residual kernel */
    if (s->cfg.precision == MG_PREC_MIXED)
        CHK(mgk_residual_f64_to_f32(s->ctx, &F->g, &L->f[1].g, L->coef, (const double *)F->b, (const double *)F->u, (float *)L->f[1].b, &ss, NULL));
    else CHK(mgk_residual_sumsq_f64(s->ctx, &F->g, L->coef, (const double *)F->b, (const double *)F->u, &ss, NULL));
    CHK(norm_from_sumsq(s, ss, &s->rchk));
    s->spec_valid = 0; s->sweep_owed = 0; s->iterate_behind = 0;
    s->iter = 0;
    s->rnorm[0] = s->rchk;                                              /* :1520 */
    s->started = 1;
    return 0;
}

int mg_solver_reset(mg_solver *s) { s->started = 0; s->iter = 0; return 0; }

static double wall(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

int mg_solver_solve(mg_solver *s) {
    CHK(start(s));
    CHK(mgk_sync(s->ctx, NULL));
    double t0 = wall();                                                  /* MPI_Wtime :1526 */
    const int devnorm = s->cfg.nranks > 1 && s->comm->allreduce_sum_dev != NULL;
    if (devnorm) CHK(need_slots(s, 1));
    while (s->iter < s->cfg.maxiter && 100000000 * s->bnorm > s->rchk && s->rchk > s->cfg.rtol * s->bnorm) {   /* :1530 */
        /* expected to be the last cycle (the contraction of the previous one would reach the tolerance, or the count runs out): skip the
         * speculative work; a wrong guess costs nothing but that saving */
        s->last_cycle = (s->iter + 1 >= s->cfg.maxiter) ||
                        (s->iter >= 1 && s->rnorm[s->iter - 1] > 0.0 && s->rchk * (s->rchk / s->rnorm[s->iter - 1]) <= s->cfg.rtol * s->bnorm);
        if (!devnorm) { int rc1 = vcycle_once(s); s->last_cycle = 0; CHK(rc1); continue; }
        /* N ranks: the cycle leaves its sum of squares in a device slot; all-reduce + read-back on the comm stream, one
         * synchronisation (of that stream) per cycle -- the convergence test needs the norm on the host */
        double ss = 0.0;
        int rc = mgk_defer_result(s->ctx, s->d_norms);
        s->deferring = 1;
        if (!rc) rc = vcycle_once(s);
        s->deferring = 0;
        s->last_cycle = 0;
        mgk_defer_result(s->ctx, NULL);
        if (rc) return rc;
        CHK(reduce_slots(s, s->d_norms, 1, &ss));
        s->rchk = sqrt(ss);
        if (s->iter < s->rnorm_cap) s->rnorm[s->iter] = s->rchk;
    }
    CHK(finalize_iterate(s));
    CHK(mgk_sync(s->ctx, NULL));
    s->solve_seconds = wall() - t0;                                      /* :1553 */
    return 0;
}

int mg_solver_cycles(mg_solver *s, int ncycles) {
    if (!s->started) CHK(start(s));
    if (s->iter + ncycles >= s->rnorm_cap) {
        int cap = s->iter + ncycles + 1;
        double *r = (double *)realloc(s->rnorm, sizeof(double) * (size_t)cap);
        if (!r) return mgfail(MGK_EINVAL, "mg_solver_cycles: out of host memory");
        s->rnorm = r; s->rnorm_cap = cap;
    }
    /* a fixed number of cycles needs no norm on the host in between: every cycle deposits its sum of squares in a device
     * slot (no synchronisation, the host runs ahead), one copy and -- on N ranks -- one all-reduce per 64 cycles at the end */
    CHK(need_slots(s, ncycles));
    const int it0 = s->iter;
    s->deferring = 1;
    int rc = 0;
    for (int q = 0; q < ncycles && !rc; q++) {
        rc = mgk_defer_result(s->ctx, s->d_norms + q);
        s->last_cycle = (q == ncycles - 1);          /* nothing follows: no speculative sweep, no owed sweep */
        if (!rc) rc = vcycle_once(s);
    }
    s->last_cycle = 0;
    mgk_defer_result(s->ctx, NULL);
    s->deferring = 0;
    if (!rc) rc = finalize_iterate(s);
    if (rc) return rc;
    double *ss = (double *)malloc(sizeof(double) * (size_t)(ncycles > 0 ? ncycles : 1));
    if (!ss) return mgfail(MGK_EINVAL, "mg_solver_cycles: out of host memory");
    rc = reduce_slots(s, s->d_norms, ncycles, ss);
    if (rc) { free(ss); return mgfail(rc, "mg_solver_cycles: reading the deferred norms"); }
    for (int q = 0; q < ncycles; q++) s->rnorm[it0 + 1 + q] = sqrt(ss[q]);
    if (ncycles > 0) s->rchk = s->rnorm[it0 + ncycles];
    free(ss);
    return 0;
}

int mg_solver_sync(mg_solver *s) {
    CHK(mgk_sync(s->ctx, NULL));
    if (s->comm && s->comm->check) { int rcc = s->comm->check(s->comm); if (rcc) return mgfail(rcc, mg_comm_last_error()); }
    return 0;
}
int mg_solver_iterations(const mg_solver *s) { return s->iter; }
double mg_solver_bnorm(const mg_solver *s) { return s->bnorm; }
const double *mg_solver_rnorm(const mg_solver *s) { return s->rnorm; }
double mg_solver_solve_seconds(const mg_solver *s) { return s->solve_seconds; }
int mg_solver_num_levels(const mg_solver *s) { return s->levels; }
int mg_solver_level_n(const mg_solver *s, int l) { return (l >= 0 && l < s->levels) ? s->L[l].n : -1; }
int mg_solver_level_local_planes(const mg_solver *s, int l, int *z0) {
    if (l < 0 || l >= s->levels) return -1;
    if (z0) *z0 = s->L[l].z0;
    return s->L[l].nzl;
}
long mg_solver_local_unknowns(const mg_solver *s) {
    const mgk_geom *g = &s->L[0].f[0].g;
    return (long)g->nx * g->ny * g->nz;
}
double mg_solver_dof_updates_per_cycle(const mg_solver *s) {
    double tot = 0.0;
    for (int l = 0; l < s->levels; l++) {
        double N = pow((double)s->L[l].n, (double)s->cfg.dim);
        int sweeps = (l == s->levels - 1 && s->levels > 1) ? s->cfg.v[1] : 2 * s->cfg.v[0];
        if (s->levels == 1) sweeps = s->cfg.v[0];
        tot += sweeps * N;
    }
    return tot;
}
