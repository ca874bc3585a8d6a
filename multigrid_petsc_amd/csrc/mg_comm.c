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
int mg_comm_halo_n(mg_comm *c, mgk_ctx *ctx, int nf, void *const *fields, const mgk_geom *const *geoms, int esz, void *stream) {
    if (nf < 1) return 0;
    if (c->halo_n) return c->halo_n(c, ctx, nf, fields, geoms, esz, stream);
    for (int q = 0; q < nf; q++) { int rc = c->halo(c, ctx, fields[q], geoms[q], esz, stream); if (rc) return rc; }
    return 0;
}

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
/* inside ncclGroupStart .. ncclGroupEnd: a failing call closes the group before returning, so that the communicator is not left
 * inside an open group (every later call on it would be queued into that group and never issued) */
#define NCKG(call) do { int rc_ = (call); if (rc_) { cfail(MGK_ECOMM, #call, g_rccl.GetErrorString(rc_)); g_rccl.GroupEnd(); return MGK_ECOMM; } } while (0)

typedef struct rccl_impl {
    nccl_comm_t comm;
    int device;
    double *scratch;       /* device, 64 doubles */
    double *pin;           /* pinned host, 64 doubles */
} rccl_impl;

/* One communicator, one stream: every RCCL call goes to the comm stream of the context (NULL selects it; any other stream
 * is refused, so that a new call site cannot drive the communicator from two streams by accident). */
static int rccl_stream(mgk_ctx *ctx, void *stream, void **out) {
    void *ms = mgk_stream_comm(ctx);
    if (stream && stream != ms) return cfail(MGK_EINVAL, "rccl back end", "every RCCL call must be issued on the context's comm stream");
    *out = ms;
    return 0;
}

int mg_comm_rccl_unique_id(void *id_out) {
    int rc = rccl_load();
    if (rc) return rc;
    nccl_uid id;
    NCK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

/* the halos of nf fields as ONE group: per neighbour nf sends and nf receives, matched in field order on both sides */
static int rccl_halo_n(mg_comm *c, mgk_ctx *ctx, int nf, void *const *fields, const mgk_geom *const *geoms, int esz, void *stream) {
    rccl_impl *im = (rccl_impl *)c->impl;
    if (c->nranks == 1 || nf < 1) return 0;
    void *s = NULL;
    int rc = rccl_stream(ctx, stream, &s);
    if (rc) return rc;
    const int dt = (esz == 8) ? NCCL_FLOAT64 : NCCL_FLOAT32;
    NCK(g_rccl.GroupStart());
    for (int q = 0; q < nf; q++) {
        const mgk_geom *g = geoms[q];
        const size_t cnt = (size_t)g->plane, pb = (size_t)esz * (size_t)g->plane;  /* elements / bytes per padded plane */
        char *f = (char *)fields[q];
        if (c->rank > 0) {
            NCKG(g_rccl.Send(f + pb, cnt, dt, c->rank - 1, im->comm, s));                       /* first interior plane */
            NCKG(g_rccl.Recv(f, cnt, dt, c->rank - 1, im->comm, s));                            /* lo ghost */
        }
        if (c->rank < c->nranks - 1) {
            NCKG(g_rccl.Send(f + (size_t)g->nz * pb, cnt, dt, c->rank + 1, im->comm, s));       /* last interior plane */
            NCKG(g_rccl.Recv(f + (size_t)(g->nz + 1) * pb, cnt, dt, c->rank + 1, im->comm, s)); /* hi ghost */
        }
    }
    NCK(g_rccl.GroupEnd());
    return 0;
}
/* the general form: any plane-sized pieces, one group */
static int rccl_exchange(mg_comm *c, mgk_ctx *ctx, int n, const void *const *send_lo, const void *const *send_hi,
                         void *const *recv_lo, void *const *recv_hi, const size_t *bytes, void *stream) {
    rccl_impl *im = (rccl_impl *)c->impl;
    if (c->nranks == 1 || n < 1) return 0;
    void *s = NULL;
    int rc = rccl_stream(ctx, stream, &s);
    if (rc) return rc;
    for (int q = 0; q < n; q++) if (bytes[q] % 4) return cfail(MGK_EINVAL, "rccl exchange", "piece size is not a multiple of 4 bytes");
    NCK(g_rccl.GroupStart());
    for (int q = 0; q < n; q++) {
        const size_t cnt = bytes[q] / 4;                          /* as 32-bit words: the bytes are what matters */
        if (c->rank > 0) {
            if (send_lo[q]) NCKG(g_rccl.Send(send_lo[q], cnt, NCCL_FLOAT32, c->rank - 1, im->comm, s));
            if (recv_lo[q]) NCKG(g_rccl.Recv(recv_lo[q], cnt, NCCL_FLOAT32, c->rank - 1, im->comm, s));
        }
        if (c->rank < c->nranks - 1) {
            if (send_hi[q]) NCKG(g_rccl.Send(send_hi[q], cnt, NCCL_FLOAT32, c->rank + 1, im->comm, s));
            if (recv_hi[q]) NCKG(g_rccl.Recv(recv_hi[q], cnt, NCCL_FLOAT32, c->rank + 1, im->comm, s));
        }
    }
    NCK(g_rccl.GroupEnd());
    return 0;
}
static int rccl_halo(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *g, int esz, void *stream) {
    void *const f[1] = {field};
    const mgk_geom *const gg[1] = {g};
    return rccl_halo_n(c, ctx, 1, f, gg, esz, stream);
}

static int rccl_allgather_planes(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *gf, const int *zstart, int esz, void *stream) {
    rccl_impl *im = (rccl_impl *)c->impl;
    if (c->nranks == 1) return 0;
    void *s = NULL;
    int rcs = rccl_stream(ctx, stream, &s);
    if (rcs) return rcs;
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
        if (mycnt) NCKG(g_rccl.Send(mine, mycnt, dt, r, im->comm, s));
        if (cnt) NCKG(g_rccl.Recv(f + (size_t)(zstart[r] + 1) * pb, cnt, dt, r, im->comm, s));
    }
    NCK(g_rccl.GroupEnd());
    return 0;
}

/* Self-test of the point-to-point entry points: a grouped ncclSend/ncclRecv of `count` elements from src to dst
 * with this rank as its own peer (the only send/recv a 1-GPU box can run).  Same call shape as rccl_halo. */
int mg_comm_rccl_self_sendrecv(mg_comm *c, mgk_ctx *ctx, const void *src, void *dst, long count, int esz) {
    if (!c || c->halo != rccl_halo) return cfail(MGK_EINVAL, "mg_comm_rccl_self_sendrecv", "not an RCCL communicator");
    rccl_impl *im = (rccl_impl *)c->impl;
    void *s = mgk_stream_comm(ctx);
    CK(mgk_stream_wait(ctx, s, mgk_stream_compute(ctx)));        /* src was filled on the compute stream */
    const int dt = (esz == 8) ? NCCL_FLOAT64 : NCCL_FLOAT32;
    NCK(g_rccl.GroupStart());
    NCKG(g_rccl.Send(src, (size_t)count, dt, c->rank, im->comm, s));
    NCKG(g_rccl.Recv(dst, (size_t)count, dt, c->rank, im->comm, s));
    NCK(g_rccl.GroupEnd());
    CK(mgk_sync(ctx, s));
    return 0;
}

/* The same grouped send/recv QUEUED on the comm stream and left running (no wait for the compute stream, no synchronisation): the
 * measurement aid of tools/rccl_overlap.py -- does RCCL's send/recv kernel run beside a marching kernel that fills the chip? */
int mg_comm_rccl_self_sendrecv_async(mg_comm *c, mgk_ctx *ctx, const void *src, void *dst, long count, int esz) {
    if (!c || c->halo != rccl_halo) return cfail(MGK_EINVAL, "mg_comm_rccl_self_sendrecv_async", "not an RCCL communicator");
    rccl_impl *im = (rccl_impl *)c->impl;
    void *s = mgk_stream_comm(ctx);
    const int dt = (esz == 8) ? NCCL_FLOAT64 : NCCL_FLOAT32;
    NCK(g_rccl.GroupStart());
    NCKG(g_rccl.Send(src, (size_t)count, dt, c->rank, im->comm, s));
    NCKG(g_rccl.Recv(dst, (size_t)count, dt, c->rank, im->comm, s));
    NCK(g_rccl.GroupEnd());
    return 0;
}

/* in place on the device, queued on the comm stream: the caller ties it to the producer of dvals with mgk_stream_wait */
static int rccl_allreduce_sum_dev(mg_comm *c, mgk_ctx *ctx, double *dvals, int n, void *stream) {
    rccl_impl *im = (rccl_impl *)c->impl;
    void *s = NULL;
    int rc = rccl_stream(ctx, stream, &s);
    if (rc) return rc;
    if (n < 1) return 0;
    NCK(g_rccl.AllReduce(dvals, dvals, (size_t)n, NCCL_FLOAT64, NCCL_SUM, im->comm, s));
    return 0;
}
/* host values, blocking: staged through pinned memory, everything on the comm stream (it first waits for the compute
 * stream, so the call is also ordered after all work queued so far); the only synchronisation is on the comm stream */
static int rccl_allreduce_sum(mg_comm *c, mgk_ctx *ctx, double *vals, int n, void *stream) {
    rccl_impl *im = (rccl_impl *)c->impl;
    if (n > 64) return cfail(MGK_EINVAL, "allreduce_sum", "at most 64 values");
    void *s = NULL;
    int rc = rccl_stream(ctx, stream, &s);
    if (rc) return rc;
    if (!im->scratch) { void *p = NULL; CK(mgk_malloc(ctx, &p, 64 * sizeof(double))); im->scratch = (double *)p; }
    if (!im->pin) { void *p = NULL; CK(mgk_host_alloc(ctx, &p, 64 * sizeof(double))); im->pin = (double *)p; }
    CK(mgk_stream_wait(ctx, s, mgk_stream_compute(ctx)));
    memcpy(im->pin, vals, sizeof(double) * (size_t)n);
    CK(mgk_h2d_async(ctx, im->scratch, im->pin, sizeof(double) * (size_t)n, s));
    NCK(g_rccl.AllReduce(im->scratch, im->scratch, (size_t)n, NCCL_FLOAT64, NCCL_SUM, im->comm, s));
    CK(mgk_d2h_async(ctx, im->pin, im->scratch, sizeof(double) * (size_t)n, s));
    CK(mgk_sync(ctx, s));
    memcpy(vals, im->pin, sizeof(double) * (size_t)n);
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
        if (im->pin) mgk_host_free(NULL, im->pin);
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
    c->halo_n = rccl_halo_n; c->allreduce_sum_dev = rccl_allreduce_sum_dev; c->exchange = rccl_exchange;
    return c;
}

/* ================================================================== */
/* loopback: ranks = threads of one process sharing one GPU            */
/* ================================================================== */
#define LOOP_MAXF 8
typedef struct loop_shared {
    int nranks;
    pthread_barrier_t bar;
    void **field;             /* posted field pointer per rank */
    int *nz;                  /* posted local plane count per rank */
    double *red;              /* nranks x 64 */
    /* grouped exchange (halo_n): what every rank posted for the group it is in */
    int *gnf, *gesz;          /* per rank: number of fields, element size */
    void **gfield;            /* nranks x LOOP_MAXF */
    int *gnz;                 /* nranks x LOOP_MAXF */
    long *gplane;             /* nranks x LOOP_MAXF */
    const void **xs_lo, **xs_hi;  /* general exchange: posted send pointers, nranks x LOOP_MAXF */
    int *xr_lo, *xr_hi;       /* ... and whether the rank receives in that slot / direction */
    int fault_rank;           /* test aid (mg_comm_loopback_inject_fault): this rank receives a WRONG plane as its lo ghost; -1: none */
} loop_shared;

typedef struct loop_impl { loop_shared *sh; } loop_impl;

void *mg_comm_loopback_shared_create(int nranks) {
    loop_shared *sh = (loop_shared *)calloc(1, sizeof(loop_shared));
    sh->nranks = nranks;
    pthread_barrier_init(&sh->bar, NULL, (unsigned)nranks);
    sh->field = (void **)calloc((size_t)nranks, sizeof(void *));
    sh->nz = (int *)calloc((size_t)nranks, sizeof(int));
    sh->red = (double *)calloc((size_t)nranks * 64, sizeof(double));
    sh->gnf = (int *)calloc((size_t)nranks, sizeof(int));
    sh->gesz = (int *)calloc((size_t)nranks, sizeof(int));
    sh->gfield = (void **)calloc((size_t)nranks * LOOP_MAXF, sizeof(void *));
    sh->gnz = (int *)calloc((size_t)nranks * LOOP_MAXF, sizeof(int));
    sh->gplane = (long *)calloc((size_t)nranks * LOOP_MAXF, sizeof(long));
    sh->xs_lo = (const void **)calloc((size_t)nranks * LOOP_MAXF, sizeof(void *)); sh->xs_hi = (const void **)calloc((size_t)nranks * LOOP_MAXF, sizeof(void *));
    sh->xr_lo = (int *)calloc((size_t)nranks * LOOP_MAXF, sizeof(int)); sh->xr_hi = (int *)calloc((size_t)nranks * LOOP_MAXF, sizeof(int));
    sh->fault_rank = -1;
    return sh;
}
/* test aid: from now on rank `rank` (>= 1) gets the lo neighbour's FIRST plane instead of its last one as lo ghost -- a transport that
 * delivers a wrong plane, for the self-test gate to catch (-1 switches it off) */
void mg_comm_loopback_inject_fault(void *p, int rank) { if (p) ((loop_shared *)p)->fault_rank = rank; }
void mg_comm_loopback_shared_destroy(void *p) {
    loop_shared *sh = (loop_shared *)p;
    if (!sh) return;
    pthread_barrier_destroy(&sh->bar);
    free(sh->field); free(sh->nz); free(sh->red);
    free(sh->gnf); free(sh->gesz); free(sh->gfield); free(sh->gnz); free(sh->gplane);
    free((void *)sh->xs_lo); free((void *)sh->xs_hi); free(sh->xr_lo); free(sh->xr_hi); free(sh);
}

static int loop_halo(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *g, int esz, void *stream) {
    loop_shared *sh = ((loop_impl *)c->impl)->sh;
    void *s = stream_of(ctx, stream);
    CK(mgk_sync(ctx, s));                         /* my planes are final */
    sh->field[c->rank] = field; sh->nz[c->rank] = g->nz;
    pthread_barrier_wait(&sh->bar);
    const size_t pb = (size_t)esz * (size_t)g->plane;
    char *f = (char *)field;
    int rc = 0;                                   /* (an error must not skip the closing barrier: the other ranks wait there) */
    if (c->rank > 0) {
        const int src_plane = (sh->fault_rank == c->rank) ? 1 : sh->nz[c->rank - 1];
        rc = mgk_d2d(ctx, f, (char *)sh->field[c->rank - 1] + (size_t)src_plane * pb, pb, s);
    }
    if (!rc && c->rank < c->nranks - 1)
        rc = mgk_d2d(ctx, f + (size_t)(g->nz + 1) * pb, (char *)sh->field[c->rank + 1] + pb, pb, s);
    if (!rc) rc = mgk_sync(ctx, s);
    pthread_barrier_wait(&sh->bar);               /* nobody overwrites a plane a neighbour still reads */
    return rc ? cfail(rc, "loopback halo", mgk_last_error()) : 0;
}

/* The grouped exchange.  Every rank posts what it believes the group to be -- number of fields, element size, plane size of every
 * field -- and compares it with its neighbours' posts before a byte moves: on a real transport a rank whose u_ghost_ok / b_ghost_ok /
 * bfar_ok flags differ from its neighbour's would send nf planes into nf' receives (a hang, or planes landing in the wrong field);
 * here that is an error every slab test would show. */
static int loop_halo_n(mg_comm *c, mgk_ctx *ctx, int nf, void *const *fields, const mgk_geom *const *geoms, int esz, void *stream) {
    loop_shared *sh = ((loop_impl *)c->impl)->sh;
    if (c->nranks == 1 || nf < 1) return 0;
    if (nf > LOOP_MAXF) return cfail(MGK_EINVAL, "loopback halo_n", "more fields than LOOP_MAXF");
    void *s = stream_of(ctx, stream);
    int rc = mgk_sync(ctx, s);
    const int me = c->rank;
    sh->gnf[me] = nf; sh->gesz[me] = esz;
    for (int q = 0; q < nf; q++) {
        sh->gfield[me * LOOP_MAXF + q] = fields[q];
        sh->gnz[me * LOOP_MAXF + q] = geoms[q]->nz;
        sh->gplane[me * LOOP_MAXF + q] = geoms[q]->plane;
    }
    pthread_barrier_wait(&sh->bar);
    for (int nb = me - 1; nb <= me + 1 && !rc; nb += 2) {
        if (nb < 0 || nb >= c->nranks) continue;
        if (sh->gnf[nb] != nf || sh->gesz[nb] != esz) {
            char d[160];
            snprintf(d, sizeof(d), "rank %d groups %d field(s) of %d bytes, its neighbour %d groups %d of %d", me, nf, esz, nb, sh->gnf[nb], sh->gesz[nb]);
            rc = cfail(MGK_ECOMM, "loopback halo_n: neighbours disagree about the group", d);
            break;
        }
        for (int q = 0; q < nf; q++)
            if (sh->gplane[nb * LOOP_MAXF + q] != geoms[q]->plane) {
                rc = cfail(MGK_ECOMM, "loopback halo_n", "neighbours disagree about the plane size of a field of the group");
                break;
            }
    }
    for (int q = 0; q < nf && !rc; q++) {
        const size_t pb = (size_t)esz * (size_t)geoms[q]->plane;
        char *f = (char *)fields[q];
        if (me > 0) {
            const int src_plane = (sh->fault_rank == me) ? 1 : sh->gnz[(me - 1) * LOOP_MAXF + q];
            rc = mgk_d2d(ctx, f, (char *)sh->gfield[(me - 1) * LOOP_MAXF + q] + (size_t)src_plane * pb, pb, s);
        }
        if (!rc && me < c->nranks - 1)
            rc = mgk_d2d(ctx, f + (size_t)(geoms[q]->nz + 1) * pb, (char *)sh->gfield[(me + 1) * LOOP_MAXF + q] + pb, pb, s);
        if (rc) cfail(rc, "loopback halo_n", mgk_last_error());
    }
    if (!rc) { rc = mgk_sync(ctx, s); if (rc) cfail(rc, "loopback halo_n", mgk_last_error()); }
    pthread_barrier_wait(&sh->bar);
    return rc;
}

/* the general exchange: every rank posts its send pointers and what it expects to receive; a slot that one neighbour sends and the other does
 * not receive (or the other way round) is an error -- on a real transport it would be a hang */
static int loop_exchange(mg_comm *c, mgk_ctx *ctx, int n, const void *const *send_lo, const void *const *send_hi,
                         void *const *recv_lo, void *const *recv_hi, const size_t *bytes, void *stream) {
    loop_shared *sh = ((loop_impl *)c->impl)->sh;
    if (c->nranks == 1 || n < 1) return 0;
    if (n > LOOP_MAXF) return cfail(MGK_EINVAL, "loopback exchange", "more slots than LOOP_MAXF");
    void *s = stream_of(ctx, stream);
    int rc = mgk_sync(ctx, s);
    const int me = c->rank;
    sh->gnf[me] = n;
    for (int q = 0; q < n; q++) {
        sh->xs_lo[me * LOOP_MAXF + q] = send_lo[q]; sh->xs_hi[me * LOOP_MAXF + q] = send_hi[q];
        sh->xr_lo[me * LOOP_MAXF + q] = recv_lo[q] != NULL; sh->xr_hi[me * LOOP_MAXF + q] = recv_hi[q] != NULL;
        sh->gplane[me * LOOP_MAXF + q] = (long)bytes[q];
    }
    pthread_barrier_wait(&sh->bar);
    for (int q = 0; q < n && !rc; q++) {
        if (me > 0) {
            const int nb = me - 1;
            if (sh->gnf[nb] != n || sh->gplane[nb * LOOP_MAXF + q] != (long)bytes[q] || (sh->xs_hi[nb * LOOP_MAXF + q] != NULL) != (recv_lo[q] != NULL) ||
                (send_lo[q] != NULL) != (sh->xr_hi[nb * LOOP_MAXF + q] != 0)) { rc = cfail(MGK_ECOMM, "loopback exchange", "neighbours disagree about a slot of the group"); break; }
            if (recv_lo[q]) rc = mgk_d2d(ctx, recv_lo[q], sh->xs_hi[nb * LOOP_MAXF + q], bytes[q], s);
        }
        if (!rc && me < c->nranks - 1) {
            const int nb = me + 1;
            if (sh->gnf[nb] != n || sh->gplane[nb * LOOP_MAXF + q] != (long)bytes[q] || (sh->xs_lo[nb * LOOP_MAXF + q] != NULL) != (recv_hi[q] != NULL) ||
                (send_hi[q] != NULL) != (sh->xr_lo[nb * LOOP_MAXF + q] != 0)) { rc = cfail(MGK_ECOMM, "loopback exchange", "neighbours disagree about a slot of the group"); break; }
            if (recv_hi[q]) rc = mgk_d2d(ctx, recv_hi[q], sh->xs_lo[nb * LOOP_MAXF + q], bytes[q], s);
        }
        if (rc && strncmp(g_cerr, "loopback", 8)) cfail(rc, "loopback exchange", mgk_last_error());
    }
    if (!rc) { rc = mgk_sync(ctx, s); if (rc) cfail(rc, "loopback exchange", mgk_last_error()); }
    pthread_barrier_wait(&sh->bar);
    return rc;
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

/* device values: read back, summed in rank order like the host form, written back (a test transport: it may block) */
static int loop_allreduce_sum_dev(mg_comm *c, mgk_ctx *ctx, double *dvals, int n, void *stream) {
    void *s = stream_of(ctx, stream);
    if (n < 1) return 0;
    double *h = (double *)malloc(sizeof(double) * (size_t)n);
    if (!h) return cfail(MGK_EINVAL, "allreduce_sum_dev", "out of host memory");
    int rc = mgk_sync(ctx, s);
    if (!rc) rc = mgk_d2h(ctx, h, dvals, sizeof(double) * (size_t)n);
    for (int q = 0; q < n && !rc; q += 64) rc = loop_allreduce_sum(c, ctx, h + q, n - q < 64 ? n - q : 64, NULL);
    if (!rc) rc = mgk_h2d(ctx, dvals, h, sizeof(double) * (size_t)n);
    free(h);
    if (rc) return cfail(rc, "loopback allreduce_sum_dev", mgk_last_error());
    return mgk_sync(ctx, NULL);
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
    c->allreduce_sum_dev = loop_allreduce_sum_dev; c->halo_n = loop_halo_n; c->exchange = loop_exchange;
    return c;
}

/* ================================================================== */
/* phantom: one rank of an N-rank run alone on its GPU (timing aid)    */
/* ================================================================== */
typedef struct phantom_impl { double lat_us, gbs; } phantom_impl;

static int phantom_hold(mg_comm *c, mgk_ctx *ctx, double bytes_per_direction, void *s) {
    const phantom_impl *im = (const phantom_impl *)c->impl;
    double us = im->lat_us + (im->gbs > 0.0 ? bytes_per_direction / (im->gbs * 1.0e3) : 0.0);   /* GB/s = 1e3 B/us */
    return mgk_delay_us(ctx, us, s);
}
/* an exchange = ONE wavefront that holds the stream for latency + bytes per direction / link bandwidth (lo and hi planes travel
 * over different links at the same time), i.e. what a send/recv kernel's residency looks like to the rest of the chip; then the
 * ghost planes are filled from the rank's own boundary planes (mirror: same bytes written, defined values) by 32 small workgroups
 * -- not by a chip-wide copy kernel, which would queue behind the marching kernel it is supposed to overlap with */
static int phantom_halo_n(mg_comm *c, mgk_ctx *ctx, int nf, void *const *fields, const mgk_geom *const *geoms, int esz, void *stream) {
    if (c->nranks == 1 || nf < 1) return 0;
    void *s = stream_of(ctx, stream);
    double bytes = 0.0;
    for (int q = 0; q < nf; q++) bytes += (double)esz * (double)geoms[q]->plane;
    CK(phantom_hold(c, ctx, bytes, s));
    for (int q = 0; q < nf; q++) {
        const mgk_geom *g = geoms[q];
        const size_t pb = (size_t)esz * (size_t)g->plane;
        char *f = (char *)fields[q];
        if (c->rank > 0) CK(mgk_paced_copy(ctx, f, f + pb, pb, 0.0, 32, s));
        if (c->rank < c->nranks - 1) CK(mgk_paced_copy(ctx, f + (size_t)(g->nz + 1) * pb, f + (size_t)g->nz * pb, pb, 0.0, 32, s));
    }
    return 0;
}
static int phantom_exchange(mg_comm *c, mgk_ctx *ctx, int n, const void *const *send_lo, const void *const *send_hi,
                            void *const *recv_lo, void *const *recv_hi, const size_t *bytes, void *stream) {
    if (c->nranks == 1 || n < 1) return 0;
    void *s = stream_of(ctx, stream);
    double most = 0.0, tlo = 0.0, thi = 0.0;
    for (int q = 0; q < n; q++) { if (send_lo[q]) tlo += (double)bytes[q]; if (send_hi[q]) thi += (double)bytes[q]; }
    most = tlo > thi ? tlo : thi;                               /* lo and hi travel over different links at the same time */
    CK(phantom_hold(c, ctx, most, s));
    for (int q = 0; q < n; q++) {                               /* what arrives: a copy of what I send the other way (defined values, same bytes written) */
        if (c->rank > 0 && recv_lo[q]) CK(mgk_paced_copy(ctx, recv_lo[q], send_lo[q] ? send_lo[q] : send_hi[q], bytes[q], 0.0, 32, s));
        if (c->rank < c->nranks - 1 && recv_hi[q]) CK(mgk_paced_copy(ctx, recv_hi[q], send_hi[q] ? send_hi[q] : send_lo[q], bytes[q], 0.0, 32, s));
    }
    return 0;
}
static int phantom_halo(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *g, int esz, void *stream) {
    void *const f[1] = {field};
    const mgk_geom *const gg[1] = {g};
    return phantom_halo_n(c, ctx, 1, f, gg, esz, stream);
}
static int phantom_allgather_planes(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *gf, const int *zstart, int esz, void *stream) {
    if (c->nranks == 1) return 0;
    void *s = stream_of(ctx, stream);
    const size_t pb = (size_t)esz * (size_t)gf->plane;
    const int me = c->rank, mine = zstart[me + 1] - zstart[me];
    double most = 0.0;
    for (int r = 0; r < c->nranks; r++) {
        if (r == me) continue;
        const int cnt = zstart[r + 1] - zstart[r];
        /* fill the other ranks' planes with copies of mine (one contiguous copy per rank, like one message per peer; a rank
         * that owns more planes than I do gets my run repeated), so that the gathered level is defined */
        for (int k = 0; k < cnt && mine > 0; k += mine) {
            const int run = cnt - k < mine ? cnt - k : mine;
            CK(mgk_d2d(ctx, (char *)field + (size_t)(zstart[r] + 1 + k) * pb, (char *)field + (size_t)(zstart[me] + 1) * pb, (size_t)run * pb, s));
        }
        if ((double)cnt * (double)pb > most) most = (double)cnt * (double)pb;
    }
    return phantom_hold(c, ctx, most, s);          /* one chunk per link, all links at once */
}
static int phantom_allreduce_sum(mg_comm *c, mgk_ctx *ctx, double *vals, int n, void *stream) {
    (void)stream;
    for (int q = 0; q < n; q++) vals[q] *= (double)c->nranks;       /* every rank is taken to contribute what I do */
    void *ms = mgk_stream_comm(ctx);
    CK(mgk_stream_wait(ctx, ms, mgk_stream_compute(ctx)));
    CK(phantom_hold(c, ctx, 0.0, ms));
    CK(mgk_sync(ctx, ms));
    return 0;
}
static int phantom_allreduce_sum_dev(mg_comm *c, mgk_ctx *ctx, double *dvals, int n, void *stream) {
    void *s = stream_of(ctx, stream);
    if (n < 1) return 0;
    CK(mgk_flat_scale(ctx, (long)n, (double)c->nranks, dvals, s));
    return phantom_hold(c, ctx, 0.0, s);
}
static int phantom_barrier(mg_comm *c, mgk_ctx *ctx) { (void)c; (void)ctx; return 0; }
static void phantom_destroy(mg_comm *c) { if (c) { free(c->impl); free(c); } }

mg_comm *mg_comm_phantom_create(int rank, int nranks, double lat_us, double link_gbs) {
    if (nranks < 1 || rank < 0 || rank >= nranks || lat_us < 0.0 || link_gbs < 0.0) { cfail(MGK_EINVAL, "mg_comm_phantom_create", "bad arguments"); return NULL; }
    phantom_impl *im = (phantom_impl *)calloc(1, sizeof(phantom_impl));
    mg_comm *c = (mg_comm *)calloc(1, sizeof(mg_comm));
    if (!im || !c) { free(im); free(c); cfail(MGK_EINVAL, "mg_comm_phantom_create", "out of host memory"); return NULL; }
    im->lat_us = lat_us; im->gbs = link_gbs;
    c->rank = rank; c->nranks = nranks; c->impl = im;
    c->halo = phantom_halo; c->allgather_planes = phantom_allgather_planes;
    c->allreduce_sum = phantom_allreduce_sum; c->barrier = phantom_barrier; c->destroy = phantom_destroy;
    c->halo_n = phantom_halo_n; c->allreduce_sum_dev = phantom_allreduce_sum_dev; c->exchange = phantom_exchange;
    return c;
}

/* ================================================================== */
/* peer: IPC-mapped mailboxes + flag words, copy-engine plane copies   */
/* ================================================================== */
/* (round 3)  One process per GPU.  Every rank owns, in fine-grained device memory (mgk_ipc_alloc):
 *   a MAILBOX  [from_lo | from_hi], each `nfmax` slots of `pmax` bytes: where its neighbours drop their boundary planes;
 *   a GATHER BOX of `gbytes`: where all ranks drop their planes of a level that becomes replicated;
 *   a FLAG BLOCK of 8-byte words: 0/1 data from lo/hi has landed (exchange number), 2/3 lo/hi has drained what I sent (free to overwrite),
 *   4+r gather data of rank r has landed, 20+r rank r has drained the gather box I wrote into, 64.. the all-reduce slots.
 * A halo exchange number k, all on the caller's stream (the comm stream of the solver's context):
 *   wait (one wave) until both neighbours acknowledged exchange k-1  ->  peer copies of my boundary planes into THEIR mailboxes
 *   (mgk_peer_copy: between devices the copy engines move the planes -- no workgroup, nothing to squeeze in beside the marching
 *   kernels)  ->  one wave stores k into their "landed" words  ->  one wave waits for k in my own "landed" words  ->  copies
 *   mailbox -> ghost planes  ->  one wave stores k into their "drained" words.
 * The flag kernels use a handful of registers: a wave slot is free beside the one-block-per-CU marching kernels, where RCCL's send/recv
 * kernel finds none.  Every rank must issue the same sequence of exchanges (the solver does).  A wait that does
 * not see its number within `timeout_s` raises the status word; the next host-synchronising hook returns MGK_ECOMM.
 * Bootstrap: mg_comm_peer_create returns the rank's 192-byte blob of IPC handles; the launcher all-gathers the blobs (bench.py: gloo)
 * and hands all of them to mg_comm_peer_connect. */
#define PEER_W_LANDED 0
#define PEER_W_DRAINED 2
#define PEER_W_GLANDED 4
#define PEER_W_GDRAINED 20
#define PEER_W_RED 64
#define PEER_FLAG_WORDS (PEER_W_RED + 2 * MGK_PEER_MAX * 65)
typedef struct peer_impl {
    mgk_ctx *ctx;                         /* owner of the allocations */
    int device, nfmax, connected;
    size_t pmax, gbytes;
    char *mbox, *gbox;                    /* mine */
    unsigned long long *flags;            /* mine */
    char *nb_mbox[2];                     /* lo / hi neighbour's mailbox as mapped here */
    char *all_gbox[MGK_PEER_MAX];
    unsigned long long *all_flags[MGK_PEER_MAX];
    unsigned long long seq_halo, seq_gather, seq_red;
    unsigned int *status;                 /* pinned host word: a flag wait timed out */
    double *red_dev;                      /* 64 doubles (device) */
    double *red_pin;                      /* 64 doubles (pinned host) */
    double timeout_s;
} peer_impl;

static int peer_check(mg_comm *c, const char *where) {
    peer_impl *im = (peer_impl *)c->impl;
    if (im->status && *im->status) return cfail(MGK_ECOMM, where, "a flag wait of the peer transport timed out (a neighbour never signalled)");
    return 0;
}
static int peer_exchange(mg_comm *c, mgk_ctx *ctx, int n, const void *const *send_lo, const void *const *send_hi,
                         void *const *recv_lo, void *const *recv_hi, const size_t *bytes, void *stream) {
    peer_impl *im = (peer_impl *)c->impl;
    if (c->nranks == 1 || n < 1) return 0;
    if (!im->connected) return cfail(MGK_EINVAL, "peer exchange", "mg_comm_peer_connect has not been called");
    if (n > im->nfmax) return cfail(MGK_EINVAL, "peer exchange", "more slots in one exchange than the mailbox has");
    for (int q = 0; q < n; q++) if (bytes[q] > im->pmax) return cfail(MGK_EINVAL, "peer exchange", "a piece is larger than a mailbox slot");
    void *s = stream_of(ctx, stream);
    const int me = c->rank, lo = me > 0, hi = me < c->nranks - 1;
    const unsigned long long k = ++im->seq_halo;
    const size_t box = (size_t)im->nfmax * im->pmax;                /* from_lo box at 0, from_hi box at `box` */
    void *w[2];
    int nw = 0;
    /* 1. the neighbours have drained what exchange k-1 put into their mailboxes */
    if (lo) w[nw++] = im->flags + PEER_W_DRAINED + 0;
    if (hi) w[nw++] = im->flags + PEER_W_DRAINED + 1;
    CK(mgk_flags_wait(ctx, w, nw, k - 1, im->timeout_s, im->status, s));
    /* 2. my pieces into their mailboxes: I am the HI neighbour of rank-1 and the LO neighbour of rank+1 */
    for (int q = 0; q < n; q++) {
        if (lo && send_lo[q]) CK(mgk_peer_copy(ctx, im->nb_mbox[0] + box + (size_t)q * im->pmax, send_lo[q], bytes[q], s));
        if (hi && send_hi[q]) CK(mgk_peer_copy(ctx, im->nb_mbox[1] + (size_t)q * im->pmax, send_hi[q], bytes[q], s));
    }
    /* 3. tell them */
    nw = 0;
    if (lo) w[nw++] = im->all_flags[me - 1] + PEER_W_LANDED + 1;
    if (hi) w[nw++] = im->all_flags[me + 1] + PEER_W_LANDED + 0;
    CK(mgk_flags_set(ctx, w, nw, k, s));
    /* 4. theirs have landed in mine */
    nw = 0;
    if (lo) w[nw++] = im->flags + PEER_W_LANDED + 0;
    if (hi) w[nw++] = im->flags + PEER_W_LANDED + 1;
    CK(mgk_flags_wait(ctx, w, nw, k, im->timeout_s, im->status, s));
    /* 5. mailbox -> where the pieces belong */
    for (int q = 0; q < n; q++) {
        if (lo && recv_lo[q]) CK(mgk_d2d(ctx, recv_lo[q], im->mbox + (size_t)q * im->pmax, bytes[q], s));
        if (hi && recv_hi[q]) CK(mgk_d2d(ctx, recv_hi[q], im->mbox + box + (size_t)q * im->pmax, bytes[q], s));
    }
    /* 6. drained: they may overwrite */
    nw = 0;
    if (lo) w[nw++] = im->all_flags[me - 1] + PEER_W_DRAINED + 1;
    if (hi) w[nw++] = im->all_flags[me + 1] + PEER_W_DRAINED + 0;
    CK(mgk_flags_set(ctx, w, nw, k, s));
    return 0;
}
static int peer_check_hook(mg_comm *c) { return peer_check(c, "peer transport"); }
static int peer_halo_n(mg_comm *c, mgk_ctx *ctx, int nf, void *const *fields, const mgk_geom *const *geoms, int esz, void *stream) {
    if (c->nranks == 1 || nf < 1) return 0;
    if (nf > MGK_PEER_MAX) return cfail(MGK_EINVAL, "peer halo", "too many fields in one exchange");
    const void *slo[MGK_PEER_MAX], *shi[MGK_PEER_MAX];
    void *rlo[MGK_PEER_MAX], *rhi[MGK_PEER_MAX];
    size_t nb[MGK_PEER_MAX];
    for (int q = 0; q < nf; q++) {
        const size_t pb = (size_t)esz * (size_t)geoms[q]->plane;
        char *f = (char *)fields[q];
        slo[q] = f + pb; shi[q] = f + (size_t)geoms[q]->nz * pb; rlo[q] = f; rhi[q] = f + (size_t)(geoms[q]->nz + 1) * pb; nb[q] = pb;
    }
    return peer_exchange(c, ctx, nf, slo, shi, rlo, rhi, nb, stream);
}
static int peer_halo(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *g, int esz, void *stream) {
    void *const f[1] = {field};
    const mgk_geom *const gg[1] = {g};
    return peer_halo_n(c, ctx, 1, f, gg, esz, stream);
}
static int peer_allgather_planes(mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *gf, const int *zstart, int esz, void *stream) {
    peer_impl *im = (peer_impl *)c->impl;
    if (c->nranks == 1) return 0;
    if (!im->connected) return cfail(MGK_EINVAL, "peer allgather", "mg_comm_peer_connect has not been called");
    const size_t pb = (size_t)esz * (size_t)gf->plane;
    if ((size_t)(gf->nz + 2) * pb > im->gbytes) return cfail(MGK_EINVAL, "peer allgather", "the level is larger than the gather box");
    void *s = stream_of(ctx, stream);
    const int me = c->rank, P = c->nranks;
    const unsigned long long k = ++im->seq_gather;
    void *w[MGK_PEER_MAX];
    int nw = 0;
    for (int r = 0; r < P; r++) if (r != me) w[nw++] = im->flags + PEER_W_GDRAINED + r;
    CK(mgk_flags_wait(ctx, w, nw, k - 1, im->timeout_s, im->status, s));
    const size_t off = (size_t)(zstart[me] + 1) * pb, mine = (size_t)(zstart[me + 1] - zstart[me]) * pb;
    for (int r = 0; r < P; r++) if (r != me && mine) CK(mgk_peer_copy(ctx, im->all_gbox[r] + off, (char *)field + off, mine, s));
    nw = 0;
    for (int r = 0; r < P; r++) if (r != me) w[nw++] = im->all_flags[r] + PEER_W_GLANDED + me;
    CK(mgk_flags_set(ctx, w, nw, k, s));
    nw = 0;
    for (int r = 0; r < P; r++) if (r != me) w[nw++] = im->flags + PEER_W_GLANDED + r;
    CK(mgk_flags_wait(ctx, w, nw, k, im->timeout_s, im->status, s));
    for (int r = 0; r < P; r++) {
        if (r == me) continue;
        const size_t o = (size_t)(zstart[r] + 1) * pb, n = (size_t)(zstart[r + 1] - zstart[r]) * pb;
        if (n) CK(mgk_d2d(ctx, (char *)field + o, im->gbox + o, n, s));
    }
    nw = 0;
    for (int r = 0; r < P; r++) if (r != me) w[nw++] = im->all_flags[r] + PEER_W_GDRAINED + me;
    CK(mgk_flags_set(ctx, w, nw, k, s));
    return 0;
}
static int peer_allreduce_sum_dev(mg_comm *c, mgk_ctx *ctx, double *dvals, int n, void *stream) {
    peer_impl *im = (peer_impl *)c->impl;
    if (n < 1) return 0;
    if (!im->connected && c->nranks > 1) return cfail(MGK_EINVAL, "peer allreduce", "mg_comm_peer_connect has not been called");
    void *s = stream_of(ctx, stream);
    void *blocks[MGK_PEER_MAX];
    for (int r = 0; r < c->nranks; r++) blocks[r] = im->all_flags[r] + PEER_W_RED;
    for (int q = 0; q < n; q += 64)
        CK(mgk_peer_allreduce(ctx, blocks, c->nranks, c->rank, ++im->seq_red, dvals + q, n - q < 64 ? n - q : 64, im->timeout_s, im->status, s));
    return 0;
}
static int peer_allreduce_sum(mg_comm *c, mgk_ctx *ctx, double *vals, int n, void *stream) {
    peer_impl *im = (peer_impl *)c->impl;
    if (n > 64) return cfail(MGK_EINVAL, "allreduce_sum", "at most 64 values");
    void *s = stream ? stream : mgk_stream_comm(ctx);
    CK(mgk_stream_wait(ctx, s, mgk_stream_compute(ctx)));
    memcpy(im->red_pin, vals, sizeof(double) * (size_t)n);
    CK(mgk_h2d_async(ctx, im->red_dev, im->red_pin, sizeof(double) * (size_t)n, s));
    int rc = peer_allreduce_sum_dev(c, ctx, im->red_dev, n, s);
    if (rc) return rc;
    CK(mgk_d2h_async(ctx, im->red_pin, im->red_dev, sizeof(double) * (size_t)n, s));
    CK(mgk_sync(ctx, s));
    memcpy(vals, im->red_pin, sizeof(double) * (size_t)n);
    return peer_check(c, "peer allreduce_sum");
}
static int peer_barrier(mg_comm *c, mgk_ctx *ctx) {
    double z = 0.0;
    return peer_allreduce_sum(c, ctx, &z, 1, NULL);
}
static void peer_destroy(mg_comm *c) {
    if (!c) return;
    peer_impl *im = (peer_impl *)c->impl;
    if (im) {
        for (int q = 0; q < 2; q++) if (im->nb_mbox[q]) mgk_ipc_close(im->ctx, im->nb_mbox[q]);
        for (int r = 0; r < c->nranks; r++) {
            if (r == c->rank) continue;
            if (im->all_gbox[r]) mgk_ipc_close(im->ctx, im->all_gbox[r]);
            if (im->all_flags[r]) mgk_ipc_close(im->ctx, im->all_flags[r]);
        }
        if (im->mbox) mgk_free(im->ctx, im->mbox);
        if (im->gbox) mgk_free(im->ctx, im->gbox);
        if (im->flags) mgk_free(im->ctx, im->flags);
        if (im->red_dev) mgk_free(im->ctx, im->red_dev);
        if (im->red_pin) mgk_host_free(im->ctx, im->red_pin);
        if (im->status) mgk_host_free(im->ctx, im->status);
        if (im->ctx) mgk_ctx_destroy(im->ctx);
        free(im);
    }
    free(c);
}
mg_comm *mg_comm_peer_create(int rank, int nranks, int device, size_t plane_bytes_max, int fields_max, size_t gather_bytes, void *blob_out) {
    if (nranks < 1 || nranks > MGK_PEER_MAX || rank < 0 || rank >= nranks || !plane_bytes_max || fields_max < 1 || !gather_bytes || !blob_out) {
        cfail(MGK_EINVAL, "mg_comm_peer_create", "bad arguments (at most 16 ranks)");
        return NULL;
    }
    peer_impl *im = (peer_impl *)calloc(1, sizeof(peer_impl));
    mg_comm *c = (mg_comm *)calloc(1, sizeof(mg_comm));
    if (!im || !c) { free(im); free(c); cfail(MGK_EINVAL, "mg_comm_peer_create", "out of host memory"); return NULL; }
    c->rank = rank; c->nranks = nranks; c->impl = im;
    im->device = device; im->nfmax = fields_max;
    im->pmax = (plane_bytes_max + 255) & ~(size_t)255; im->gbytes = gather_bytes;
    { const char *e = getenv("MG_PEER_TIMEOUT_S"); im->timeout_s = (e && atof(e) > 0.0) ? atof(e) : 60.0; }
    char *blob = (char *)blob_out;
    void *p = NULL;
    int rc = mgk_ctx_create(&im->ctx, device);
    if (!rc) { rc = mgk_ipc_alloc(im->ctx, 2 * (size_t)fields_max * im->pmax, &p, blob); im->mbox = (char *)p; }
    if (!rc) { rc = mgk_ipc_alloc(im->ctx, sizeof(unsigned long long) * PEER_FLAG_WORDS, &p, blob + MGK_IPC_HANDLE_BYTES); im->flags = (unsigned long long *)p; }
    if (!rc) { rc = mgk_ipc_alloc(im->ctx, gather_bytes, &p, blob + 2 * MGK_IPC_HANDLE_BYTES); im->gbox = (char *)p; }
    if (!rc) { rc = mgk_malloc(im->ctx, &p, 64 * sizeof(double)); im->red_dev = (double *)p; }
    if (!rc) { rc = mgk_host_alloc(im->ctx, &p, 64 * sizeof(double)); im->red_pin = (double *)p; }
    if (!rc) { rc = mgk_host_alloc(im->ctx, &p, 64); im->status = (unsigned int *)p; if (!rc) *im->status = 0; }
    if (rc) { cfail(rc, "mg_comm_peer_create", mgk_last_error()); peer_destroy(c); return NULL; }
    im->all_flags[rank] = im->flags; im->all_gbox[rank] = im->gbox;
    c->halo = peer_halo; c->halo_n = peer_halo_n; c->exchange = peer_exchange; c->allgather_planes = peer_allgather_planes;
    c->allreduce_sum = peer_allreduce_sum; c->allreduce_sum_dev = peer_allreduce_sum_dev; c->barrier = peer_barrier; c->destroy = peer_destroy;
    c->check = peer_check_hook;
    if (nranks == 1) im->connected = 1;
    return c;
}
int mg_comm_peer_connect(mg_comm *c, const void *all_blobs) {
    if (!c || c->halo != peer_halo || !all_blobs) return cfail(MGK_EINVAL, "mg_comm_peer_connect", "not a peer communicator");
    peer_impl *im = (peer_impl *)c->impl;
    if (im->connected) return 0;
    const char *b = (const char *)all_blobs;
    for (int r = 0; r < c->nranks; r++) {
        if (r == c->rank) continue;
        const char *br = b + (size_t)r * MG_PEER_BLOB_BYTES;
        void *p = NULL;
        CK(mgk_ipc_open(im->ctx, br + MGK_IPC_HANDLE_BYTES, &p)); im->all_flags[r] = (unsigned long long *)p;
        CK(mgk_ipc_open(im->ctx, br + 2 * MGK_IPC_HANDLE_BYTES, &p)); im->all_gbox[r] = (char *)p;
        if (r == c->rank - 1 || r == c->rank + 1) { CK(mgk_ipc_open(im->ctx, br, &p)); im->nb_mbox[r == c->rank - 1 ? 0 : 1] = (char *)p; }
    }
    im->connected = 1;
    return 0;
}

/* ================================================================== */
/* self-test of a transport (collective)                               */
/* ================================================================== */
static int st_fail(const char *what, int rank, double got, double want) {
    char d[256];
    snprintf(d, sizeof(d), "rank %d: %s: got %.17g, expected %.17g", rank, what, got, want);
    return cfail(MGK_ECOMM, "mg_comm_selftest", d);
}
/* plane p (0 .. nz+1, ghosts included) of a field whose owner is `rank`, field number q */
static double st_code(int rank, int q, int p) { return 1000.0 * (rank + 1) + 100.0 * q + p; }

/* Collective-safe on failure: the gate exists for the run in which something IS wrong, so a rank that sees a wrong
 * plane, a wrong sum or a failing hook must not leave the sequence -- its peers would sit in the matching send/recv for ever.  Every
 * buffer is set up before the first collective and the ranks agree on that through ONE all-reduce of a status word; from then on
 * every rank makes the same fixed sequence of collectives whatever it observes, records the FIRST failure (code + text) and returns
 * it at the end. */
int mg_comm_selftest(mg_comm *c, mgk_ctx *ctx) {
    if (!c || !ctx) return cfail(MGK_EINVAL, "mg_comm_selftest", "null argument");
    snprintf(g_cerr, sizeof(g_cerr), "ok");
    const int P = c->nranks, me = c->rank;
    void *cs = mgk_stream_compute(ctx), *ms = mgk_stream_comm(ctx);
    int first = 0;
    char first_err[sizeof(g_cerr)] = "";
#define NOTE(code, what) do { if (!first) { first = (code); if (!strncmp(g_cerr, "ok", 2)) cfail((code), (what), mgk_last_error()); \
                                            snprintf(first_err, sizeof(first_err), "%s", g_cerr); } } while (0)
    /* ---- setup (local): halo fields (two of different plane counts) in fp64 and fp32, the all-gather level, the reduce slot ---- */
    mgk_geom g[2][2], gg;
    void *d[2][2] = {{NULL, NULL}, {NULL, NULL}}, *dgat = NULL, *dred = NULL;
    char *h[2][2] = {{NULL, NULL}, {NULL, NULL}};
    double *hgat = NULL;
    int *zs = NULL;
    int bad = 0;
    mgk_geom_init(&g[0][0], 3, 15, 7, 3); mgk_geom_init(&g[0][1], 3, 15, 7, 2);
    mgk_geom_init_f32(&g[1][0], 3, 15, 7, 3); mgk_geom_init_f32(&g[1][1], 3, 15, 7, 2);
    mgk_geom_init(&gg, 3, 7, 7, 2 * P);
    for (int t = 0; t < 2 && !bad; t++) {
        const int esz = t ? 4 : 8;
        for (int q = 0; q < 2 && !bad; q++) {
            const size_t bytes = (size_t)esz * (size_t)g[t][q].total;
            h[t][q] = (char *)calloc(1, bytes);
            if (!h[t][q] || mgk_malloc(ctx, &d[t][q], bytes)) { bad = 1; break; }
            for (int p = 0; p <= g[t][q].nz + 1; p++)
                for (long e = 0; e < g[t][q].plane; e++) {
                    const double v = (p == 0 || p == g[t][q].nz + 1) ? -1.0 : st_code(me, q, p);
                    if (esz == 8) ((double *)h[t][q])[(long)p * g[t][q].plane + e] = v; else ((float *)h[t][q])[(long)p * g[t][q].plane + e] = (float)v;
                }
            if (mgk_h2d(ctx, d[t][q], h[t][q], bytes)) bad = 1;
        }
    }
    zs = (int *)calloc((size_t)P + 1, sizeof(int));
    hgat = (double *)calloc((size_t)gg.total, sizeof(double));
    if (!bad && (!zs || !hgat || mgk_malloc(ctx, &dgat, sizeof(double) * (size_t)gg.total) || mgk_malloc(ctx, &dred, 2 * sizeof(double)))) bad = 1;
    if (!bad) {
        for (int r = 0; r <= P; r++) zs[r] = 2 * r;      /* rank r produces planes [2r, 2r+2) of a level with 2P planes */
        for (int p = zs[me]; p < zs[me + 1]; p++)
            for (long e = 0; e < gg.plane; e++) hgat[(long)(p + 1) * gg.plane + e] = st_code(me, 7, p);
        double v[2] = {(double)(me + 1), 0.25};
        if (mgk_h2d(ctx, dgat, hgat, sizeof(double) * (size_t)gg.total) || mgk_h2d(ctx, dred, v, sizeof(v))) bad = 1;
    }
    /* ---- agreement: did the setup succeed everywhere?  (the first collective; every rank enters it) ---- */
    {
        double st = bad ? 1.0 : 0.0;
        int rc = c->allreduce_sum(c, ctx, &st, 1, NULL);
        if (rc) NOTE(rc, "mg_comm_selftest: all-reduce of the setup status");
        else if (st != 0.0) {
            char dd[96];
            snprintf(dd, sizeof(dd), "buffers could not be set up on %g rank(s)%s", st, bad ? " (this one among them)" : "");
            first = cfail(MGK_ECOMM, "mg_comm_selftest", dd);
            snprintf(first_err, sizeof(first_err), "%s", g_cerr);
        }
        if (first) goto done;               /* nothing was exchanged yet, and every rank takes this exit together (or the transport's
                                             * own all-reduce is broken, in which case nothing can be agreed on at all) */
    }
    /* ---- halo (one field) and halo_n (two fields of different plane counts), fp64 then fp32 ---- */
    for (int t = 0; t < 2; t++) {
        const int esz = t ? 4 : 8;
        for (int pass = 0; pass < 2; pass++) {               /* pass 0: halo on field 0; pass 1: halo_n on both */
            int rc = mgk_stream_wait(ctx, ms, cs);
            if (rc) NOTE(rc, "mg_comm_selftest: stream wait");
            if (pass == 0) rc = c->halo(c, ctx, d[t][0], &g[t][0], esz, ms);
            else {
                void *const ff[2] = {d[t][0], d[t][1]};
                const mgk_geom *const gq[2] = {&g[t][0], &g[t][1]};
                rc = mg_comm_halo_n(c, ctx, 2, ff, gq, esz, ms);
            }
            if (rc) NOTE(rc, pass ? "mg_comm_selftest: halo_n" : "mg_comm_selftest: halo");
            rc = mgk_stream_wait(ctx, cs, ms);
            if (rc) NOTE(rc, "mg_comm_selftest: stream wait");
            for (int q = 0; q < (pass ? 2 : 1); q++) {
                const size_t bytes = (size_t)esz * (size_t)g[t][q].total;
                rc = mgk_d2h(ctx, h[t][q], d[t][q], bytes);
                if (rc) { NOTE(rc, "mg_comm_selftest: read-back"); continue; }
                for (int p = 0; p <= g[t][q].nz + 1; p++) {
                    double want = st_code(me, q, p);
                    if (p == 0) want = me > 0 ? st_code(me - 1, q, g[t][q].nz) : -1.0;
                    if (p == g[t][q].nz + 1) want = me < P - 1 ? st_code(me + 1, q, 1) : -1.0;
                    for (long e = 0; e < g[t][q].plane; e += (g[t][q].plane - 1 > 0 ? g[t][q].plane - 1 : 1)) {   /* first and last element */
                        const double got = esz == 8 ? ((double *)h[t][q])[(long)p * g[t][q].plane + e] : (double)((float *)h[t][q])[(long)p * g[t][q].plane + e];
                        if (got != want && !first) { first = st_fail(pass ? "halo_n plane" : "halo plane", me, got, want); snprintf(first_err, sizeof(first_err), "%s", g_cerr); }
                    }
                }
            }
        }
    }
    /* ---- the general exchange (optional hook): planes sent from INSIDE a field, one slot that travels downwards only ---- */
    if (c->exchange) {
        const mgk_geom *ga = &g[0][0], *gb = &g[0][1];                 /* fp64, nz = 3 and nz = 2, same plane size */
        const size_t pb = sizeof(double) * (size_t)ga->plane;
        char *fa = (char *)d[0][0], *fb = (char *)d[0][1];
        const void *slo[2] = {fa + 2 * pb, fa + 3 * pb}, *shi[2] = {fa + 2 * pb, NULL};
        void *rlo[2] = {fb, NULL}, *rhi[2] = {fb + (size_t)(gb->nz + 1) * pb, fa + (size_t)(ga->nz + 1) * pb};
        const size_t nb[2] = {pb, pb};
        int rc = mgk_stream_wait(ctx, ms, cs);
        if (rc) NOTE(rc, "mg_comm_selftest: stream wait");
        rc = c->exchange(c, ctx, 2, slo, shi, rlo, rhi, nb, ms);
        if (rc) NOTE(rc, "mg_comm_selftest: exchange");
        rc = mgk_stream_wait(ctx, cs, ms);
        if (rc) NOTE(rc, "mg_comm_selftest: stream wait");
        for (int q = 0; q < 2; q++) {
            rc = mgk_d2h(ctx, h[0][q], d[0][q], sizeof(double) * (size_t)g[0][q].total);
            if (rc) NOTE(rc, "mg_comm_selftest: read-back");
        }
        if (!rc) {
            const double *ha = (const double *)h[0][0], *hb = (const double *)h[0][1];
            const double got[3] = {hb[0], hb[(long)(gb->nz + 1) * gb->plane + gb->plane - 1], ha[(long)(ga->nz + 1) * ga->plane]};
            /* b's lo ghost: plane 2 of the lower neighbour's field a; b's hi ghost: plane 2 of the upper neighbour's; a's hi ghost: its plane 3.
             * Without a neighbour the ghost keeps what the halo tests left there (-1) */
            const double want[3] = {me > 0 ? st_code(me - 1, 0, 2) : -1.0, me < P - 1 ? st_code(me + 1, 0, 2) : -1.0, me < P - 1 ? st_code(me + 1, 0, 3) : -1.0};
            for (int q = 0; q < 3; q++)
                if (got[q] != want[q] && !first) { first = st_fail("exchanged plane", me, got[q], want[q]); snprintf(first_err, sizeof(first_err), "%s", g_cerr); }
        }
    }
    /* ---- all-gather of planes ---- */
    {
        int rc = mgk_stream_wait(ctx, ms, cs);
        if (rc) NOTE(rc, "mg_comm_selftest: stream wait");
        rc = c->allgather_planes(c, ctx, dgat, &gg, zs, 8, ms);
        if (rc) NOTE(rc, "mg_comm_selftest: allgather_planes");
        rc = mgk_stream_wait(ctx, cs, ms);
        if (rc) NOTE(rc, "mg_comm_selftest: stream wait");
        rc = mgk_d2h(ctx, hgat, dgat, sizeof(double) * (size_t)gg.total);
        if (rc) NOTE(rc, "mg_comm_selftest: read-back");
        for (int r = 0; r < P && !rc; r++)
            for (int p = zs[r]; p < zs[r + 1]; p++) {
                const double got = hgat[(long)(p + 1) * gg.plane + gg.plane - 1], want = st_code(r, 7, p);
                if (got != want && !first) { first = st_fail("all-gathered plane", me, got, want); snprintf(first_err, sizeof(first_err), "%s", g_cerr); }
            }
    }
    /* ---- all-reduce: host form and device form ---- */
    {
        double v[3] = {(double)(me + 1), 1.0, 0.5};
        int rc = c->allreduce_sum(c, ctx, v, 3, NULL);
        if (rc) NOTE(rc, "mg_comm_selftest: allreduce_sum");
        const double want[3] = {0.5 * P * (P + 1), (double)P, 0.5 * P};
        for (int q = 0; q < 3 && !rc; q++)
            if (v[q] != want[q] && !first) { first = st_fail("all-reduce (host values)", me, v[q], want[q]); snprintf(first_err, sizeof(first_err), "%s", g_cerr); }
    }
    if (c->allreduce_sum_dev) {
        double v[2] = {0.0, 0.0};
        int rc = mgk_stream_wait(ctx, ms, cs);
        if (rc) NOTE(rc, "mg_comm_selftest: stream wait");
        rc = c->allreduce_sum_dev(c, ctx, (double *)dred, 2, ms);
        if (rc) NOTE(rc, "mg_comm_selftest: allreduce_sum_dev");
        rc = mgk_stream_wait(ctx, cs, ms);
        if (rc) NOTE(rc, "mg_comm_selftest: stream wait");
        rc = mgk_d2h(ctx, v, dred, sizeof(v));
        if (rc) NOTE(rc, "mg_comm_selftest: read-back");
        const double want[2] = {0.5 * P * (P + 1), 0.25 * P};
        for (int q = 0; q < 2 && !rc; q++)
            if (v[q] != want[q] && !first) { first = st_fail("all-reduce (device values)", me, v[q], want[q]); snprintf(first_err, sizeof(first_err), "%s", g_cerr); }
    }
done:
#undef NOTE
    for (int t = 0; t < 2; t++) for (int q = 0; q < 2; q++) { if (d[t][q]) mgk_free(ctx, d[t][q]); free(h[t][q]); }
    if (dgat) mgk_free(ctx, dgat);
    if (dred) mgk_free(ctx, dred);
    free(zs); free(hgat);
    if (first) snprintf(g_cerr, sizeof(g_cerr), "%s", first_err);
    else snprintf(g_cerr, sizeof(g_cerr), "ok");
    return first;
}
