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

typedef struct mg_level {
    int n;                  /* unknowns per side of the whole grid */
    int z0, nzl;            /* owned planes [z0, z0+nzl) (3-D); whole grid when replicated / 2-D */
    int distributed;
    mgk_geom g;             /* local geometry */
    double coef[7], dinv, h;
    double *u, *b, *rv, *tmp, *p2;
    mgk_geom g32;           /* mixed precision: geometry and fields of the fp32 correction cycle */
    float *u32, *b32, *rv32, *tmp32;
    int guess_nonzero;      /* KSPSetInitialGuessNonzero state of ksp[l] (src/solver.c:1532,1537,1543) */
    int u_ghost_ok;         /* z ghost planes of `u` hold the neighbours' current boundary planes */
    int u_ghost_pending;    /* ... but the exchange is still in flight on the comm stream */
} mg_level;

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
    double solve_seconds;
    /* profiling */
    int lgraph;             /* levels >= lgraph form the launch-bound coarse part replayed as one HIP graph (0: off) */
    void *coarse_graph;
    int prof_on, prof_n;
    void *timers[MG_MAX_TIMERS];
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
    c->rank = 0; c->nranks = 1; c->dist_min_n = 127;
    c->fuse = -1;
    c->overlap = -1;
    c->graph = -1;
}

static int alloc_field(mg_solver *s, const mgk_geom *g, double **p) {
    void *q = NULL;
    CHK(mgk_malloc(s->ctx, &q, sizeof(double) * (size_t)g->total));
    *p = (double *)q;
    return 0;
}

int mg_solver_create(mg_solver **out, const mg_config *cfg, mg_comm *comm) {
    if (!out || !cfg) return mgfail(MGK_EINVAL, "mg_solver_create: null argument");
    if (cfg->dim != 2 && cfg->dim != 3) return mgfail(MGK_EINVAL, "mg_solver_create: dim must be 2 or 3");
    if (cfg->levels < 1 || cfg->levels > MG_MAX_LEVELS) return mgfail(MGK_EINVAL, "mg_solver_create: bad level count");
    if (cfg->npts < 3) return mgfail(MGK_EINVAL, "mg_solver_create: npts < 3");
    if (cfg->precision == MG_PREC_MIXED && (cfg->dim != 3 || cfg->nranks > 1 || cfg->ksp_type != MG_KSP_RICHARDSON))
        return mgfail(MGK_EINVAL, "mg_solver_create: mixed precision is built for 3-D, one GPU, Richardson+Jacobi");
    /* npts-1 must be divisible by 2^(levels-1) and the coarsest grid must keep >= 1 unknown */
    for (int l = 0; l < cfg->levels; l++) {
        int n = mg_grid_n(cfg->npts, l);
        int f = 1 << l;
        if (n < 1 || (cfg->npts - 1) % f != 0 || (n & 1) == 0)
            return mgfail(MGK_EINVAL, "mg_solver_create: npts-1 must be 2^m with m >= levels (vertex-centred coarsening, src/matbuild.c:62-66)");
    }
    if (cfg->nranks > 1 && (!comm || cfg->dim != 3))
        return mgfail(MGK_EINVAL, "mg_solver_create: nranks > 1 needs a communicator and dim == 3");
    if (cfg->ksp_type == MG_KSP_CHEBYSHEV && !(cfg->emax > cfg->emin && cfg->emin > 0.0))
        return mgfail(MGK_EINVAL, "mg_solver_create: chebyshev needs 0 < emin < emax (-ksp_chebyshev_eigenvalues)");

    mg_solver *s = (mg_solver *)calloc(1, sizeof(mg_solver));
    s->cfg = *cfg;
    if (s->cfg.rtol <= 0) s->cfg.rtol = 1.e-7;
    if (s->cfg.dist_min_n <= 0) s->cfg.dist_min_n = 127;
    if (s->cfg.fuse < 0) s->cfg.fuse = 7;
    if (s->cfg.overlap < 0) s->cfg.overlap = 1;
    if (s->cfg.graph < 0) s->cfg.graph = 1;
    if (s->cfg.nranks < 1) s->cfg.nranks = 1;
    s->comm = comm;
    s->levels = cfg->levels;
    int rc = mgk_ctx_create(&s->ctx, cfg->device);
    if (rc) { free(s); return mgfail(rc, "mg_solver_create: mgk_ctx_create"); }

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
        rc = mgk_geom_init(&L->g, cfg->dim, L->n, L->n, L->nzl);
        if (rc) { mg_solver_destroy(s); return mgfail(rc, "mg_solver_create: geometry"); }
        level_stencil(cfg->dim, L->n, L->coef, &L->h);
        L->dinv = 1.0 / L->coef[cfg->dim == 3 ? 3 : 2];      /* PCJACOBI: 1/diag(A) */
        if (cfg->precision == MG_PREC_MIXED) {
            /* fp64 only where the outer defect correction lives (level 0: u, b); fp32 everywhere else */
            void *q = NULL;
            if (l == 0 && ((rc = alloc_field(s, &L->g, &L->u)) || (rc = alloc_field(s, &L->g, &L->b)))) { mg_solver_destroy(s); return rc; }
            if ((rc = mgk_geom_init_f32(&L->g32, 3, L->n, L->n, L->n))) { mg_solver_destroy(s); return mgfail(rc, "mg_solver_create: fp32 geometry"); }
            float **f[4] = {&L->u32, &L->b32, &L->rv32, &L->tmp32};
            for (int k = 0; k < 4; k++) {
                if ((rc = mgk_malloc(s->ctx, &q, sizeof(float) * (size_t)L->g32.total))) { mg_solver_destroy(s); return mgfail(rc, "mg_solver_create: fp32 field"); }
                *f[k] = (float *)q;
            }
            continue;
        }
        if ((rc = alloc_field(s, &L->g, &L->u)) || (rc = alloc_field(s, &L->g, &L->b)) ||
            (rc = alloc_field(s, &L->g, &L->rv)) || (rc = alloc_field(s, &L->g, &L->tmp))) {
            mg_solver_destroy(s); return rc;
        }
        if (cfg->ksp_type == MG_KSP_CHEBYSHEV && (rc = alloc_field(s, &L->g, &L->p2))) { mg_solver_destroy(s); return rc; }
    }
    /* coarse part for the HIP graph: the first level that is neither distributed nor fed by a distributed one and
     * has at most 2^21 unknowns (3-D n <= 127, 2-D n <= 1023): below that a kernel is shorter than its launch */
    s->lgraph = 0;
    if (s->cfg.graph && s->cfg.ksp_type == MG_KSP_RICHARDSON && s->cfg.precision == MG_PREC_FP64) {
        for (int l = (s->ldist > 0 ? s->ldist + 1 : 1); l < s->levels; l++) {
            double N = pow((double)s->L[l].n, (double)cfg->dim);
            if (N <= 2097152.0) { s->lgraph = l; break; }
        }
        if (s->lgraph && s->levels - s->lgraph < 2) s->lgraph = 0;      /* not worth a graph */
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
        if (s->coarse_graph) mgk_graph_destroy(s->ctx, s->coarse_graph);
        for (int l = 0; l < s->levels; l++) {
            mg_level *L = &s->L[l];
            if (L->u) mgk_free(s->ctx, L->u);
            if (L->b) mgk_free(s->ctx, L->b);
            if (L->rv) mgk_free(s->ctx, L->rv);
            if (L->tmp) mgk_free(s->ctx, L->tmp);
            if (L->p2) mgk_free(s->ctx, L->p2);
            if (L->u32) mgk_free(s->ctx, L->u32);
            if (L->b32) mgk_free(s->ctx, L->b32);
            if (L->rv32) mgk_free(s->ctx, L->rv32);
            if (L->tmp32) mgk_free(s->ctx, L->tmp32);
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
    coords_uniform(npts, c);
    for (int j = 0; j < n; j++) {
        t[j] = sin(MG_PI * c[j + 1]);
        tc[j] = (s->cfg.dim == 2 ? -2 * MG_PI * MG_PI : -3 * MG_PI * MG_PI) * t[j];
    }
    int rc = 0;
    if (cx) rc = upload(s, tc, n, cx);
    if (!rc && sx) rc = upload(s, t, n, sx);
    if (!rc) rc = upload(s, t, n, sy);
    if (!rc && s->cfg.dim == 3) rc = upload(s, t + L->z0, L->nzl, sz); else if (!rc) *sz = NULL;
    free(c); free(t); free(tc);
    return rc;
}

int mg_solver_set_rhs_problem(mg_solver *s) {
    double *cx = NULL, *sy = NULL, *sz = NULL;
    CHK(sin_tables(s, &cx, NULL, &sy, &sz));
    CHK(mgk_fill_separable_f64(s->ctx, &s->L[0].g, cx, sy, sz, s->L[0].b, NULL));
    CHK(mgk_sync(s->ctx, NULL));
    mgk_free(s->ctx, cx); mgk_free(s->ctx, sy); if (sz) mgk_free(s->ctx, sz);
    return mg_solver_reset(s);
}

int mg_solver_set_rhs_host(mg_solver *s, const double *b_compact) {
    const mg_level *L = &s->L[0];
    size_t n = (size_t)L->g.nx * L->g.ny * L->g.nz;
    double *d = NULL;
    CHK(upload(s, b_compact, n, &d));
    CHK(mgk_pack_f64(s->ctx, &L->g, d, L->b, NULL));
    CHK(mgk_sync(s->ctx, NULL));
    mgk_free(s->ctx, d);
    return mg_solver_reset(s);
}

int mg_solver_get_solution(mg_solver *s, double *u_compact) {
    const mg_level *L = &s->L[0];
    size_t n = (size_t)L->g.nx * L->g.ny * L->g.nz;
    void *d = NULL;
    CHK(mgk_malloc(s->ctx, &d, sizeof(double) * n));
    CHK(mgk_unpack_f64(s->ctx, &L->g, L->u, (double *)d, NULL));
    CHK(mgk_d2h(s->ctx, u_compact, d, sizeof(double) * n));
    mgk_free(s->ctx, d);
    return 0;
}

int mg_solver_error_norms(mg_solver *s, double err[3]) {
    double *sx = NULL, *sy = NULL, *sz = NULL;
    CHK(sin_tables(s, NULL, &sx, &sy, &sz));
    double e[3];
    CHK(mgk_error_sums_f64(s->ctx, &s->L[0].g, s->L[0].u, sx, sy, sz, e, NULL));
    mgk_free(s->ctx, sx); mgk_free(s->ctx, sy); if (sz) mgk_free(s->ctx, sz);
    if (s->cfg.nranks > 1) {
        /* max over ranks via sums of one-hot is not available: exchange through allreduce of
         * [sum|e|, sum e^2] and a separate max emulated by gathering (ranks are few) */
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
/* the cycle                                                           */
/* ------------------------------------------------------------------ */
/* Every RCCL operation of a solver is issued on ITS comm stream (one communicator, one stream: a
 * single total order on every rank); cross-stream events tie it to the compute stream.
 * Blocking form: the exchange sees everything queued on the compute stream so far, and everything
 * queued on the compute stream afterwards sees the ghosts. */
static int halo(mg_solver *s, mg_level *L, double *field) {
    if (!L->distributed) return 0;
    void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
    CHK(mgk_stream_wait(s->ctx, ms, cs));
    CHK(s->comm->halo(s->comm, s->ctx, field, &L->g, ms));
    CHK(mgk_stream_wait(s->ctx, cs, ms));
    return 0;
}

/* make the ghost planes of L->u valid and visible to the compute stream */
static int ensure_u_ghosts(mg_solver *s, mg_level *L) {
    if (!L->distributed) return 0;
    if (L->u_ghost_pending) {
        CHK(mgk_stream_wait(s->ctx, mgk_stream_compute(s->ctx), mgk_stream_comm(s->ctx)));
        L->u_ghost_pending = 0;
        L->u_ghost_ok = 1;
    }
    if (!L->u_ghost_ok) { CHK(halo(s, L, L->u)); L->u_ghost_ok = 1; }
    return 0;
}

static void *prof_begin(mg_solver *s, int level) {
    if (!s->prof_on || level != 0 || s->prof_n >= MG_MAX_TIMERS) return NULL;
    if (s->prof_n >= s->ntimers_created) {
        if (mgk_timer_create(s->ctx, &s->timers[s->ntimers_created])) return NULL;
        s->ntimers_created++;
    }
    void *t = s->timers[s->prof_n++];
    mgk_timer_start(s->ctx, t, NULL);
    return t;
}
static void prof_end(mg_solver *s, void *t) { if (t) mgk_timer_stop(s->ctx, t, NULL); }

int mg_solver_profile(mg_solver *s, int enable) { s->prof_on = enable; s->prof_n = 0; return 0; }
int mg_solver_profile_read(mg_solver *s, double *total_ms, int *launches) {
    double tot = 0.0, ms;
    for (int q = 0; q < s->prof_n; q++) { CHK(mgk_timer_elapsed_ms(s->ctx, s->timers[q], &ms)); tot += ms; }
    *total_ms = tot; *launches = s->prof_n;
    s->prof_n = 0;
    return 0;
}

static void swap_ptr(double **a, double **b) { double *t = *a; *a = *b; *b = t; }

/* KSPSolve(ksp[l], b[l], u[l]) with KSP_NORM_NONE and max_it = maxit (src/solver.c:1465-1509) */
static int smooth(mg_solver *s, int l, int maxit) {
    mg_level *L = &s->L[l];
    const size_t bytes = sizeof(double) * (size_t)L->g.total;
    if (s->cfg.ksp_type == MG_KSP_RICHARDSON) {
        if (maxit == 0 && !L->guess_nonzero) { CHK(mgk_memset0(s->ctx, L->u, bytes, NULL)); L->u_ghost_ok = 0; L->u_ghost_pending = 0; }   /* KSPSolve zero-fills */
        for (int it = 0; it < maxit; it++) {
            if (it == 0 && !L->guess_nonzero) {
                /* r = b, x = 0 + scale*(B b): u is not read */
                CHK(mgk_jacobi_zero_f64(s->ctx, &L->g, L->dinv, s->cfg.scale, L->b, L->tmp, NULL));
            } else if (L->distributed && s->cfg.overlap && L->g.nz >= 3) {
                /* boundary planes first, ship them on the comm stream, sweep the interior meanwhile */
                void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
                const int nz = L->g.nz;
                CHK(ensure_u_ghosts(s, L));
                CHK(mgk_jacobi_range_f64(s->ctx, &L->g, L->coef, L->dinv, s->cfg.scale, L->b, L->u, L->tmp, 0, 1, cs));
                CHK(mgk_jacobi_range_f64(s->ctx, &L->g, L->coef, L->dinv, s->cfg.scale, L->b, L->u, L->tmp, nz - 1, nz, cs));
                CHK(mgk_stream_wait(s->ctx, ms, cs));
                void *t = prof_begin(s, l);
                CHK(mgk_jacobi_range_f64(s->ctx, &L->g, L->coef, L->dinv, s->cfg.scale, L->b, L->u, L->tmp, 1, nz - 1, cs));
                prof_end(s, t);
                CHK(s->comm->halo(s->comm, s->ctx, L->tmp, &L->g, ms));
                swap_ptr(&L->u, &L->tmp);
                L->u_ghost_pending = 1; L->u_ghost_ok = 0;
                continue;
            } else {
                CHK(ensure_u_ghosts(s, L));
                void *t = prof_begin(s, l);
                CHK(mgk_jacobi_f64(s->ctx, &L->g, L->coef, L->dinv, s->cfg.scale, L->b, L->u, L->tmp, NULL));
                prof_end(s, t);
            }
            L->u_ghost_ok = 0; L->u_ghost_pending = 0;
            swap_ptr(&L->u, &L->tmp);
        }
        return 0;
    }
    /* KSPCHEBYSHEV, classic three-term recurrence (see oracle/mgo.c mgo_chebyshev_csr) */
    double scale = 2.0 / (s->cfg.emax + s->cfg.emin), alpha = 1.0 - scale * s->cfg.emin, Gamma = 1.0;
    double mu = 1.0 / alpha, omegaprod = 2.0 / alpha, ckm1 = 1.0, ck = mu, ckp1;
    double *pkm1 = L->u, *pk = L->tmp, *pkp1 = L->p2;
    if (!L->guess_nonzero) {
        CHK(mgk_memset0(s->ctx, pkm1, bytes, NULL));
        L->u_ghost_ok = 0; L->u_ghost_pending = 0;
        if (maxit > 0) CHK(mgk_jacobi_zero_f64(s->ctx, &L->g, L->dinv, scale, L->b, pk, NULL));
    } else if (maxit > 0) {
        CHK(ensure_u_ghosts(s, L));
        CHK(mgk_jacobi_f64(s->ctx, &L->g, L->coef, L->dinv, scale, L->b, pkm1, pk, NULL));
    }
    if (maxit == 0) return 0;
    for (int it = 1; it < maxit; it++) {
        ckp1 = 2.0 * mu * ck - ckm1;
        double omega = omegaprod * ck / ckp1;
        CHK(halo(s, L, pk));
        CHK(mgk_cheby_f64(s->ctx, &L->g, L->coef, L->dinv, 1.0 - omega, omega, omega * Gamma * scale,
                          L->b, pk, pkm1, pkp1, NULL));
        double *t = pkm1; pkm1 = pk; pk = pkp1; pkp1 = t;
        ckm1 = ck; ck = ckp1;
    }
    L->u = pk; L->tmp = pkm1; L->p2 = pkp1;
    L->u_ghost_ok = 0; L->u_ghost_pending = 0;
    return 0;
}

/* KSPBuildResidual(ksp[l],NULL,rv[l],&r) : rv = b - A u (src/solver.c:1534,1545) */
static int residual(mg_solver *s, int l) {
    mg_level *L = &s->L[l];
    CHK(ensure_u_ghosts(s, L));
    CHK(mgk_residual_f64(s->ctx, &L->g, L->coef, L->b, L->u, L->rv, NULL));
    return 0;
}

static int norm_from_sumsq(mg_solver *s, double ss, double *out) {
    if (s->cfg.nranks > 1) CHK(s->comm->allreduce_sum(s->comm, s->ctx, &ss, 1, NULL));
    *out = sqrt(ss);
    return 0;
}

/* MatMult(res[l-1], r[l-1], b[l]) (src/solver.c:1535) */
static int restrict_to(mg_solver *s, int l) {
    mg_level *F = &s->L[l - 1], *Cq = &s->L[l];
    CHK(halo(s, F, F->rv));
    if (F->distributed && !Cq->distributed) {
        /* slab -> replicated: produce my coarse planes in place, then all-gather them */
        mgk_geom gc = Cq->g;
        int c0 = s->zstart[s->cfg.rank], c1 = s->zstart[s->cfg.rank + 1];
        gc.nz = c1 - c0;
        CHK(mgk_restrict_fw_f64(s->ctx, &F->g, &gc, F->rv, Cq->b + (long)c0 * Cq->g.plane, NULL));
        {
            void *cs = mgk_stream_compute(s->ctx), *ms = mgk_stream_comm(s->ctx);
            CHK(mgk_stream_wait(s->ctx, ms, cs));
            CHK(s->comm->allgather_planes(s->comm, s->ctx, Cq->b, &Cq->g, s->zstart, ms));
            CHK(mgk_stream_wait(s->ctx, cs, ms));
        }
        return 0;
    }
    CHK(mgk_restrict_fw_f64(s->ctx, &F->g, &Cq->g, F->rv, Cq->b, NULL));
    return 0;
}

/* MatMult(pro[l],u[l+1],rv[l]); VecAXPY(u[l],1.0,rv[l]) (src/solver.c:1540-1541) */
static int prolong_from(mg_solver *s, int l) {
    mg_level *F = &s->L[l], *Cq = &s->L[l + 1];
    if (F->distributed && !Cq->distributed) {
        mgk_geom gc = Cq->g;
        int c0 = s->zstart[s->cfg.rank], c1 = s->zstart[s->cfg.rank + 1];
        gc.nz = c1 - c0;
        if (F->u_ghost_pending) CHK(ensure_u_ghosts(s, F));
        CHK(mgk_prolong_add_f64(s->ctx, &F->g, &gc, Cq->u + (long)c0 * Cq->g.plane, F->u, NULL));
        F->u_ghost_ok = 0;
        return 0;
    }
    CHK(ensure_u_ghosts(s, Cq));
    if (F->u_ghost_pending) CHK(ensure_u_ghosts(s, F));
    CHK(mgk_prolong_add_f64(s->ctx, &F->g, &Cq->g, Cq->u, F->u, NULL));
    F->u_ghost_ok = 0;
    return 0;
}

/* prolongation fused into the first post-smoothing sweep: u <- Jacobi(u + P u_c)  (src/solver.c:1540-1542) */
static int prolong_smooth(mg_solver *s, int l) {
    mg_level *F = &s->L[l], *Cq = &s->L[l + 1];
    const int v0 = s->cfg.v[0];
    if (!(s->cfg.fuse & 2) || s->cfg.dim != 3 || s->cfg.ksp_type != MG_KSP_RICHARDSON || v0 < 1) {
        CHK(prolong_from(s, l));
        return smooth(s, l, v0);
    }
    mgk_geom gc = Cq->g;
    const double *ucoarse = Cq->u;
    if (F->distributed && !Cq->distributed) {
        int c0 = s->zstart[s->cfg.rank], c1 = s->zstart[s->cfg.rank + 1];
        gc.nz = c1 - c0;
        ucoarse = Cq->u + (long)c0 * Cq->g.plane;
    } else {
        CHK(ensure_u_ghosts(s, Cq));
    }
    CHK(ensure_u_ghosts(s, F));          /* the neighbours' boundary planes BEFORE the correction */
    /* not counted by the profile: that one times the plain sweep kernel (bench.py roofline leg) */
    CHK(mgk_prolong_jacobi_f64(s->ctx, &F->g, &gc, F->coef, F->dinv, s->cfg.scale, F->b, ucoarse, F->u, F->tmp, NULL));
    swap_ptr(&F->u, &F->tmp);
    F->u_ghost_ok = 0; F->u_ghost_pending = 0;
    return smooth(s, l, v0 - 1);
}

/* ---- mixed precision (BASELINE config 5): fp32 correction cycle inside an fp64 defect-correction loop ---- */
static int smooth32(mg_solver *s, int l, int maxit, int guess_nonzero) {
    mg_level *L = &s->L[l];
    if (maxit == 0 && !guess_nonzero) CHK(mgk_memset0(s->ctx, L->u32, sizeof(float) * (size_t)L->g32.total, NULL));
    for (int it = 0; it < maxit; it++) {
        if (it == 0 && !guess_nonzero) CHK(mgk_jacobi_zero_f32(s->ctx, &L->g32, L->dinv, s->cfg.scale, L->b32, L->tmp32, NULL));
        else {
            void *t = prof_begin(s, l);
            CHK(mgk_jacobi_f32(s->ctx, &L->g32, L->coef, L->dinv, s->cfg.scale, L->b32, L->u32, L->tmp32, NULL));
            prof_end(s, t);
        }
        float *q = L->u32; L->u32 = L->tmp32; L->tmp32 = q;
    }
    return 0;
}

/* one outer iteration: e = Vcycle32((float) r) from e = 0;  u += (double) e;  r = b - A u (fp64), ||r|| */
static int vcycle_once_mixed(mg_solver *s) {
    const int levels = s->levels, *v = s->cfg.v;
    CHK(smooth32(s, 0, v[0], 0));
    for (int l = 1; l < levels; l++) {
        mg_level *F = &s->L[l - 1], *Cq = &s->L[l];
        CHK(mgk_residual_f32(s->ctx, &F->g32, F->coef, F->b32, F->u32, F->rv32, NULL));
        CHK(mgk_restrict_fw_f32(s->ctx, &F->g32, &Cq->g32, F->rv32, Cq->b32, NULL));
        CHK(smooth32(s, l, l == levels - 1 ? v[1] : v[0], 0));
    }
    for (int l = levels - 2; l >= 0; l--) {
        mg_level *F = &s->L[l];
        if ((s->cfg.fuse & 2) && v[0] >= 1) {
            CHK(mgk_prolong_jacobi_f32(s->ctx, &F->g32, &s->L[l + 1].g32, F->coef, F->dinv, s->cfg.scale, F->b32, s->L[l + 1].u32, F->u32, F->tmp32, NULL));
            float *q = F->u32; F->u32 = F->tmp32; F->tmp32 = q;
            CHK(smooth32(s, l, v[0] - 1, 1));
        } else {
            CHK(mgk_prolong_add_f32(s->ctx, &F->g32, &s->L[l + 1].g32, s->L[l + 1].u32, F->u32, NULL));
            CHK(smooth32(s, l, v[0], 1));
        }
    }
    mg_level *L = &s->L[0];
    double ss;
    CHK(mgk_correct_f64_from_f32(s->ctx, &L->g, &L->g32, L->u32, L->u, NULL));
    CHK(mgk_residual_f64_to_f32(s->ctx, &L->g, &L->g32, L->coef, L->b, L->u, L->b32, &ss, NULL));
    s->rchk = sqrt(ss);
    s->iter++;
    if (s->iter < s->rnorm_cap) s->rnorm[s->iter] = s->rchk;
    return 0;
}

/* one step of the descent: b_l = R(b_{l-1} - A u_{l-1}); smooth level l from a zero guess (src/solver.c:1534-1537) */
static int descend(mg_solver *s, int l) {
    const int levels = s->levels, *v = s->cfg.v;
    mg_level *F = &s->L[l - 1];
    if ((s->cfg.fuse & 4) && s->cfg.dim == 3 && !F->distributed && F->n + 1 <= 1024) {
        /* :1534-1535 in one pass: b_l = R (b - A u), the fine residual is never written */
        CHK(mgk_residual_restrict_f64(s->ctx, &F->g, &s->L[l].g, F->coef, F->b, F->u, s->L[l].b, NULL));
    } else {
        CHK(residual(s, l - 1));                                        /* :1534 */
        CHK(restrict_to(s, l));                                         /* :1535 */
    }
    CHK(smooth(s, l, l == levels - 1 ? v[1] : v[0]));                   /* :1536 */
    if (l != levels - 1) s->L[l].guess_nonzero = 1;                     /* :1537 */
    return 0;
}

/* levels lg..L-1: down from lg-1 and back up to lg.  Every buffer pointer is the same at entry of every cycle
 * (each level swaps u/tmp an even number of times per cycle; the coarsest is copied back when v1 is odd), so the
 * recorded kernels stay valid. */
static int coarse_part(mg_solver *s, int lg) {
    const int levels = s->levels;
    for (int l = lg; l < levels; l++) CHK(descend(s, l));
    if (s->cfg.v[1] & 1) {                                              /* restore the coarsest level's buffer identity */
        mg_level *Cz = &s->L[levels - 1];
        CHK(mgk_d2d(s->ctx, Cz->tmp, Cz->u, sizeof(double) * (size_t)Cz->g.total, NULL));
        swap_ptr(&Cz->u, &Cz->tmp);
    }
    for (int l = levels - 2; l >= lg; l--) {
        CHK(prolong_smooth(s, l));                                      /* :1540-1542 */
        s->L[l].guess_nonzero = 0;                                      /* :1543 (l != 0 here) */
    }
    return 0;
}

/* body of the while loop, src/solver.c:1531-1549 */
static int vcycle_once(mg_solver *s) {
    if (s->cfg.precision == MG_PREC_MIXED) return vcycle_once_mixed(s);
    const int levels = s->levels, *v = s->cfg.v;
    const int lg = s->lgraph ? s->lgraph : levels;                      /* levels >= lg run as one HIP graph */
    CHK(smooth(s, 0, v[0]));                                            /* :1531 */
    if (s->iter == 0) s->L[0].guess_nonzero = 1;                        /* :1532 */
    for (int l = 1; l < lg; l++) CHK(descend(s, l));
    if (lg < levels) {
        if (!s->coarse_graph) {                                         /* record once ... */
            CHK(mgk_capture_begin(s->ctx));
            int rc = coarse_part(s, lg);
            void *ge = NULL;
            int rc2 = mgk_capture_end(s->ctx, &ge);
            if (rc || rc2) return mgfail(rc ? rc : rc2, "HIP graph capture of the coarse levels");
            s->coarse_graph = ge;
        }
        CHK(mgk_graph_launch(s->ctx, s->coarse_graph));                 /* ... replay every cycle */
    }
    for (int l = (lg < levels ? lg - 1 : levels - 2); l >= 0; l--) {
        CHK(prolong_smooth(s, l));                                      /* :1540-1542 */
        if (l != 0) s->L[l].guess_nonzero = 0;                          /* :1543 */
    }
    /* :1545-1546  r0 = b0 - A0 u0 ; ||r0|| */
    mg_level *L = &s->L[0];
    double ss;
    CHK(ensure_u_ghosts(s, L));
    if (s->cfg.fuse & 1) CHK(mgk_residual_sumsq_f64(s->ctx, &L->g, L->coef, L->b, L->u, &ss, NULL));
    else {
        CHK(mgk_residual_f64(s->ctx, &L->g, L->coef, L->b, L->u, L->rv, NULL));
        CHK(mgk_sumsq_f64(s->ctx, &L->g, L->rv, &ss, NULL));
    }
    CHK(norm_from_sumsq(s, ss, &s->rchk));
    s->iter++;
    if (s->iter < s->rnorm_cap) s->rnorm[s->iter] = s->rchk;            /* :1549 */
    return 0;
}

/* src/solver.c:1512-1523 */
static int start(mg_solver *s) {
    mg_level *L = &s->L[0];
    double ss;
    CHK(mgk_sumsq_f64(s->ctx, &L->g, L->b, &ss, NULL));                 /* VecNorm(b[0]) :1512 */
    CHK(norm_from_sumsq(s, ss, &s->bnorm));
    for (int l = 0; l < s->levels; l++) { s->L[l].guess_nonzero = 0; s->L[l].u_ghost_ok = 0; s->L[l].u_ghost_pending = 0; }
    CHK(mgk_memset0(s->ctx, L->u, sizeof(double) * (size_t)L->g.total, NULL));   /* VecSet(u[0],0) :1514 */
    /* rv = A u - b with u = 0 (:1516-1517); ||A u - b|| = ||b - A u||, evaluated by the same residual kernel */
    if (s->cfg.precision == MG_PREC_MIXED) CHK(mgk_residual_f64_to_f32(s->ctx, &L->g, &L->g32, L->coef, L->b, L->u, L->b32, &ss, NULL));
    else CHK(mgk_residual_sumsq_f64(s->ctx, &L->g, L->coef, L->b, L->u, &ss, NULL));
    CHK(norm_from_sumsq(s, ss, &s->rchk));
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
    while (s->iter < s->cfg.maxiter && 100000000 * s->bnorm > s->rchk && s->rchk > s->cfg.rtol * s->bnorm)   /* :1530 */
        CHK(vcycle_once(s));
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
    for (int q = 0; q < ncycles; q++) CHK(vcycle_once(s));
    return 0;
}

int mg_solver_sync(mg_solver *s) { CHK(mgk_sync(s->ctx, NULL)); return 0; }
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
    const mg_level *L = &s->L[0];
    return (long)L->g.nx * L->g.ny * L->g.nz;
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
