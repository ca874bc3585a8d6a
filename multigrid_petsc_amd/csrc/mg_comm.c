/*
 * mg_comm.c -- halo exchange / reductions for the z-slab decomposed V-cycle (include/mg_comm.h).
 *
 * What it replaces in the reference: PETSc's MPIAIJ VecScatter inside every MatMult/KSPSolve and the
 * MPI_Allreduce inside VecNorm (reference call sites src/solver.c:1531-1546).
 *
 * rccl back end: librccl is dlopen()ed (no link-time dependency, single-GPU runs never load it);
 *   neighbour planes travel as grouped ncclSend/ncclRecv -- xGMI is point-to-point, a z-slab has at
 *   most two neighbours, each reached over its own link, so there is no ring and no bucketing.
 * loopback back end: ranks are threads of one process on one GPU (tests on a single-GPU box).
 */
#include "mg_comm.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_cerr[512] = "ok";
const char *mg_comm_last_error(void) { return g_cerr; }
static int cfail(int code, const char *what, const char *detail) {
    snprintf(g_cerr, sizeof(g_cerr), "%s%s%s (code %d)", what, detail ? ": " : "", detail ? detail : "", code);
    return code;
}
#define CK(call) do { int rc_ = (call); if (rc_) return cfail(rc_, #call, mgk_last_error()); } while (0)

void mg_comm_destroy(mg_comm *c) { if (c && c->destroy) c->destroy(c); }

int mg_comm_halo(mg_comm *c, mgk_ctx *ctx, double *field, const mgk_geom *g) { return c->halo(c, ctx, field, g, 8, NULL); }
int mg_comm_allreduce_sum(mg_comm *c, mgk_ctx *ctx, double *vals, int n) { return c->allreduce_sum(c, ctx, vals, n, NULL); }

static void *stream_of(mgk_ctx *ctx, void *stream) { return stream ? stream : mgk_stream_compute(ctx); }

/* ================================================================== */
/* RCCL                                                                */
/* ================================================================== */
typedef struct { char internal[MG_RCCL_ID_BYTES]; } nccl_uid;      /* ncclUniqueId */
typedef void *nccl_comm_t;
enum { NCCL_SUM = 0, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8 };          /* ncclSum, ncclFloat, ncclDouble */

typedef struct rccl_api {
    void *dl;
    int (*GetUniqueId)(nccl_uid *);
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid, int);
    int (*CommDestroy)(nccl_comm_t);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, void *);
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, void *);
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, void *);
    const char *(*GetErrorString)(int);
} rccl_api;

static rccl_api g_rccl;
static pthread_mutex_t g_rccl_lock = PTHREAD_MUTEX_INITIALIZER;

static int rccl_load(void) {
    pthread_mutex_lock(&g_rccl_lock);
    if (g_rccl.dl) { pthread_mutex_unlock(&g_rccl_lock); return 0; }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", NULL};
    void *dl = NULL;
    for (int q = 0; names[q] && !dl; q++) dl = dlopen(names[q], RTLD_NOW | RTLD_GLOBAL);
    if (!dl) { pthread_mutex_unlock(&g_rccl_lock); return cfail(MGK_ECOMM, "dlopen(librccl)", dlerror()); }
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(dl, name); \
        if (!g_rccl.field) { pthread_mutex_unlock(&g_rccl_lock); return cfail(MGK_ECOMM, "dlsym", name); } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.dl = dl;
    pthread_mutex_unlock(&g_rccl_lock);
    return 0;
}
#define NCK(call) do { int rc_ = (call); if (rc_) return cfail(MGK_ECOMM, #call, g_rccl.GetErrorString(rc_)); } while (0)

typedef struct rccl_impl {
    nccl_comm_t comm;
    int device;
    double *scratch;       /* device, 64 doubles */
} rccl_impl;

int mg_comm_rccl_unique_id(void *id_out) {
    int rc = rccl_load();
    if (rc) return rc;
    nccl_uid id;
    NCK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

static int rccl_halo(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *g, int esz, void *stream) {
    rccl_impl *im = (rccl_impl *)c->impl;
    if (c->nranks == 1) return 0;
    void *s = stream_of(ctx, stream);
    const size_t cnt = (size_t)g->plane, pb = (size_t)esz * (size_t)g->plane;      /* elements / bytes per padded plane */
    const int dt = (esz == 8) ? NCCL_FLOAT64 : NCCL_FLOAT32;
    char *f = (char *)field;
    NCK(g_rccl.GroupStart());
    if (c->rank > 0) {
        NCK(g_rccl.Send(f + pb, cnt, dt, c->rank - 1, im->comm, s));                       /* first interior plane */
        NCK(g_rccl.Recv(f, cnt, dt, c->rank - 1, im->comm, s));                            /* lo ghost */
    }
    if (c->rank < c->nranks - 1) {
        NCK(g_rccl.Send(f + (size_t)g->nz * pb, cnt, dt, c->rank + 1, im->comm, s));       /* last interior plane */
        NCK(g_rccl.Recv(f + (size_t)(g->nz + 1) * pb, cnt, dt, c->rank + 1, im->comm, s)); /* hi ghost */
    }
    NCK(g_rccl.GroupEnd());
    return 0;
}

static int rccl_allgather_planes(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *gf, const int *zstart, int esz, void *stream) {
    rccl_impl *im = (rccl_impl *)c->impl;
    if (c->nranks == 1) return 0;
    void *s = stream_of(ctx, stream);
    const int me = c->rank;
    const size_t pb = (size_t)esz * (size_t)gf->plane;
    const int dt = (esz == 8) ? NCCL_FLOAT64 : NCCL_FLOAT32;
    char *f = (char *)field;
    char *mine = f + (size_t)(zstart[me] + 1) * pb;
    const size_t mycnt = (size_t)(zstart[me + 1] - zstart[me]) * (size_t)gf->plane;
    NCK(g_rccl.GroupStart());
    for (int r = 0; r < c->nranks; r++) {
        if (r == me) continue;
        const size_t cnt = (size_t)(zstart[r + 1] - zstart[r]) * (size_t)gf->plane;
        if (mycnt) NCK(g_rccl.Send(mine, mycnt, dt, r, im->comm, s));
        if (cnt) NCK(g_rccl.Recv(f + (size_t)(zstart[r] + 1) * pb, cnt, dt, r, im->comm, s));
    }
    NCK(g_rccl.GroupEnd());
    return 0;
}

/* Self-test of the point-to-point entry points: a grouped ncclSend/ncclRecv of `count` elements from src to dst
 * with this rank as its own peer (the only send/recv a 1-GPU box can run).  Same call shape as rccl_halo. */
int mg_comm_rccl_self_sendrecv(mg_comm *c, mgk_ctx *ctx, const void *src, void *dst, long count, int esz) {
    if (!c || c->halo != rccl_halo) return cfail(MGK_EINVAL, "mg_comm_rccl_self_sendrecv", "not an RCCL communicator");
    rccl_impl *im = (rccl_impl *)c->impl;
    void *s = stream_of(ctx, NULL);
    const int dt = (esz == 8) ? NCCL_FLOAT64 : NCCL_FLOAT32;
    NCK(g_rccl.GroupStart());
    NCK(g_rccl.Send(src, (size_t)count, dt, c->rank, im->comm, s));
    NCK(g_rccl.Recv(dst, (size_t)count, dt, c->rank, im->comm, s));
    NCK(g_rccl.GroupEnd());
    CK(mgk_sync(ctx, s));
    return 0;
}

static int rccl_allreduce_sum(mg_comm *c, mgk_ctx *ctx, double *vals, int n, void *stream) {
    rccl_impl *im = (rccl_impl *)c->impl;
    if (n > 64) return cfail(MGK_EINVAL, "allreduce_sum", "at most 64 values");
    if (!im->scratch) { void *p = NULL; CK(mgk_malloc(ctx, &p, 64 * sizeof(double))); im->scratch = (double *)p; }
    (void)stream;
    CK(mgk_h2d(ctx, im->scratch, vals, sizeof(double) * (size_t)n));          /* ordered on the compute stream */
    NCK(g_rccl.AllReduce(im->scratch, im->scratch, (size_t)n, NCCL_FLOAT64, NCCL_SUM, im->comm, mgk_stream_compute(ctx)));
    CK(mgk_d2h(ctx, vals, im->scratch, sizeof(double) * (size_t)n));          /* synchronises */
    return 0;
}

static int rccl_barrier(mg_comm *c, mgk_ctx *ctx) {
    double z = 0.0;
    return rccl_allreduce_sum(c, ctx, &z, 1, NULL);
}

static void rccl_destroy(mg_comm *c) {
    if (!c) return;
    rccl_impl *im = (rccl_impl *)c->impl;
    if (im) {
        if (im->comm) g_rccl.CommDestroy(im->comm);
        free(im);     /* scratch is released with its context */
    }
    free(c);
}

mg_comm *mg_comm_rccl_create(int rank, int nranks, const void *id, int device) {
    if (rccl_load()) return NULL;
    if (mgk_set_device(device)) { cfail(MGK_ECOMM, "mgk_set_device", mgk_last_error()); return NULL; }
    nccl_uid uid;
    memcpy(&uid, id, sizeof(uid));
    rccl_impl *im = (rccl_impl *)calloc(1, sizeof(rccl_impl));
    im->device = device;
    int rc = g_rccl.CommInitRank(&im->comm, nranks, uid, rank);
    if (rc) { cfail(MGK_ECOMM, "ncclCommInitRank", g_rccl.GetErrorString(rc)); free(im); return NULL; }
    mg_comm *c = (mg_comm *)calloc(1, sizeof(mg_comm));
    c->rank = rank; c->nranks = nranks; c->impl = im;
    c->halo = rccl_halo; c->allgather_planes = rccl_allgather_planes;
    c->allreduce_sum = rccl_allreduce_sum; c->barrier = rccl_barrier; c->destroy = rccl_destroy;
    return c;
}

/* ================================================================== */
/* loopback: ranks = threads of one process sharing one GPU            */
/* ================================================================== */
typedef struct loop_shared {
    int nranks;
    pthread_barrier_t bar;
    void **field;             /* posted field pointer per rank */
    int *nz;                  /* posted local plane count per rank */
    double *red;              /* nranks x 64 */
} loop_shared;

typedef struct loop_impl { loop_shared *sh; } loop_impl;

void *mg_comm_loopback_shared_create(int nranks) {
    loop_shared *sh = (loop_shared *)calloc(1, sizeof(loop_shared));
    sh->nranks = nranks;
    pthread_barrier_init(&sh->bar, NULL, (unsigned)nranks);
    sh->field = (void **)calloc((size_t)nranks, sizeof(void *));
    sh->nz = (int *)calloc((size_t)nranks, sizeof(int));
    sh->red = (double *)calloc((size_t)nranks * 64, sizeof(double));
    return sh;
}
void mg_comm_loopback_shared_destroy(void *p) {
    loop_shared *sh = (loop_shared *)p;
    if (!sh) return;
    pthread_barrier_destroy(&sh->bar);
    free(sh->field); free(sh->nz); free(sh->red); free(sh);
}

static int loop_halo(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *g, int esz, void *stream) {
    loop_shared *sh = ((loop_impl *)c->impl)->sh;
    void *s = stream_of(ctx, stream);
    CK(mgk_sync(ctx, s));                         /* my planes are final */
    sh->field[c->rank] = field; sh->nz[c->rank] = g->nz;
    pthread_barrier_wait(&sh->bar);
    const size_t pb = (size_t)esz * (size_t)g->plane;
    char *f = (char *)field;
    if (c->rank > 0)
        CK(mgk_d2d(ctx, f, (char *)sh->field[c->rank - 1] + (size_t)sh->nz[c->rank - 1] * pb, pb, s));
    if (c->rank < c->nranks - 1)
        CK(mgk_d2d(ctx, f + (size_t)(g->nz + 1) * pb, (char *)sh->field[c->rank + 1] + pb, pb, s));
    CK(mgk_sync(ctx, s));
    pthread_barrier_wait(&sh->bar);               /* nobody overwrites a plane a neighbour still reads */
    return 0;
}

static int loop_allgather_planes(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *gf, const int *zstart, int esz, void *stream) {
    loop_shared *sh = ((loop_impl *)c->impl)->sh;
    void *s = stream_of(ctx, stream);
    CK(mgk_sync(ctx, s));
    sh->field[c->rank] = field;
    pthread_barrier_wait(&sh->bar);
    const size_t pb = (size_t)esz * (size_t)gf->plane;
    for (int r = 0; r < c->nranks; r++) {
        if (r == c->rank) continue;
        const size_t off = (size_t)(zstart[r] + 1) * pb;
        const size_t bytes = (size_t)(zstart[r + 1] - zstart[r]) * pb;
        if (bytes) CK(mgk_d2d(ctx, (char *)field + off, (char *)sh->field[r] + off, bytes, s));
    }
    CK(mgk_sync(ctx, s));
    pthread_barrier_wait(&sh->bar);
    return 0;
}

static int loop_allreduce_sum(mg_comm *c, mgk_ctx *ctx, double *vals, int n, void *stream) {
    (void)ctx; (void)stream;
    loop_shared *sh = ((loop_impl *)c->impl)->sh;
    if (n > 64) return cfail(MGK_EINVAL, "allreduce_sum", "at most 64 values");
    memcpy(sh->red + (size_t)c->rank * 64, vals, sizeof(double) * (size_t)n);
    pthread_barrier_wait(&sh->bar);
    for (int q = 0; q < n; q++) {
        double sum = 0.0;
        for (int r = 0; r < c->nranks; r++) sum += sh->red[(size_t)r * 64 + q];    /* rank order: same bits everywhere */
        vals[q] = sum;
    }
    pthread_barrier_wait(&sh->bar);
    return 0;
}

static int loop_barrier(mg_comm *c, mgk_ctx *ctx) {
    (void)ctx;
    pthread_barrier_wait(&((loop_impl *)c->impl)->sh->bar);
    return 0;
}
static void loop_destroy(mg_comm *c) { if (c) { free(c->impl); free(c); } }

mg_comm *mg_comm_loopback_create(void *shared, int rank) {
    loop_shared *sh = (loop_shared *)shared;
    if (!sh || rank < 0 || rank >= sh->nranks) { cfail(MGK_EINVAL, "mg_comm_loopback_create", "bad rank"); return NULL; }
    loop_impl *im = (loop_impl *)calloc(1, sizeof(loop_impl));
    im->sh = sh;
    mg_comm *c = (mg_comm *)calloc(1, sizeof(mg_comm));
    c->rank = rank; c->nranks = sh->nranks; c->impl = im;
    c->halo = loop_halo; c->allgather_planes = loop_allgather_planes;
    c->allreduce_sum = loop_allreduce_sum; c->barrier = loop_barrier; c->destroy = loop_destroy;
    return c;
}
