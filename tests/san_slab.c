/* tests/san_slab.c -- TEST INFRASTRUCTURE: the slab (multi-rank) host logic of csrc/mg_solver.c + csrc/mg_comm.c under
 * AddressSanitizer / UBSan on the CPU, over tests/mock_mgk.cpp.  P loopback ranks (threads) solve the 3-D problem; the
 * concatenated solution must equal the single-rank one bit for bit; the transport self-test runs on every rank, and the
 * phantom communicator drives one rank of 8 through a few cycles.  usage: san_slab P npts levels dist_min_n [mixed] */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mgsolve.h"
#include "mg_comm.h"

typedef struct { pthread_barrier_t bar; char blobs[MGK_PEER_MAX][MG_PEER_BLOB_BYTES]; } peer_boot;      /* the launcher's all-gather of the handle blobs */
typedef struct { int rank, P, npts, levels, dmin, mixed, rc; void *shared; double *u; long n; int it; peer_boot *boot; } job;

static void cfg(mg_config *c, const job *j, int rank, int P) {
    mg_config_default(c);
    c->dim = 3; c->npts = j->npts; c->levels = j->levels; c->scale = 6.0 / 7.0; c->maxiter = 60;
    c->rank = rank; c->nranks = P; c->dist_min_n = j->dmin; c->precision = j->mixed ? MG_PREC_MIXED : MG_PREC_FP64;
    c->pair_min_n = 7;                /* two-sweep passes on the small test levels too */
}
static void *work(void *p) {
    job *j = (job *)p;
    mg_comm *cm = NULL;
    if (j->P > 1 && j->boot) {
        /* the peer transport (mailboxes + flag words): planes of the finest level, 5 fields per exchange, gather box for the whole 3-D level */
        mgk_geom g0; mgk_geom_init(&g0, 3, j->npts - 2, j->npts - 2, j->npts - 2);
        cm = mg_comm_peer_create(j->rank, j->P, 0, sizeof(double) * (size_t)g0.plane, 5, sizeof(double) * (size_t)g0.total, j->boot->blobs[j->rank]);
        pthread_barrier_wait(&j->boot->bar);
        if (!cm || mg_comm_peer_connect(cm, j->boot->blobs)) { fprintf(stderr, "peer: %s\n", mg_comm_last_error()); return NULL; }
        pthread_barrier_wait(&j->boot->bar);
    } else if (j->P > 1) cm = mg_comm_loopback_create(j->shared, j->rank);
    mg_config c; cfg(&c, j, j->rank, j->P);
    mg_solver *s = NULL;
    j->rc = 1;
    if (j->P > 1) {
        mgk_ctx *ctx = NULL;
        if (mgk_ctx_create(&ctx, 0) || mg_comm_selftest(cm, ctx)) { fprintf(stderr, "selftest: %s\n", mg_comm_last_error()); return NULL; }
        mgk_ctx_destroy(ctx);
    }
    if (mg_solver_create(&s, &c, cm)) { fprintf(stderr, "create: %s\n", mg_last_error()); return NULL; }
    if (mg_solver_set_rhs_problem(s) || mg_solver_solve(s)) { fprintf(stderr, "solve: %s\n", mg_last_error()); return NULL; }
    j->n = mg_solver_local_unknowns(s);
    j->u = (double *)malloc(sizeof(double) * (size_t)j->n);
    j->it = mg_solver_iterations(s);
    double e[3];
    if (mg_solver_get_solution(s, j->u) || mg_solver_error_norms(s, e)) { fprintf(stderr, "fetch: %s\n", mg_last_error()); return NULL; }
    /* fixed-count cycling with deferred norms as bench.py does it */
    if (mg_solver_reset(s) || mg_solver_cycles(s, 3) || mg_solver_sync(s)) { fprintf(stderr, "cycles: %s\n", mg_last_error()); return NULL; }
    mg_solver_destroy(s);
    if (cm) mg_comm_destroy(cm);
    j->rc = 0;
    return NULL;
}

/* the first-run gate with a transport that delivers ONE wrong plane (rank `bad` gets a wrong lo ghost): every rank must come back
 * from mg_comm_selftest -- nobody left inside a collective --, the rank that saw the wrong plane with MGK_ECOMM and its message,
 * the others with 0 */
typedef struct { int rank; void *shared; int rc; char msg[512]; } fjob;
static void *fault_work(void *p) {
    fjob *j = (fjob *)p;
    mg_comm *cm = mg_comm_loopback_create(j->shared, j->rank);
    mgk_ctx *ctx = NULL;
    j->rc = -1;
    if (!cm || mgk_ctx_create(&ctx, 0)) return NULL;
    j->rc = mg_comm_selftest(cm, ctx);
    snprintf(j->msg, sizeof(j->msg), "%s", mg_comm_last_error());
    mgk_ctx_destroy(ctx);
    mg_comm_destroy(cm);
    return NULL;
}
static int fault_run(int P, int bad_rank) {
    fjob *js = (fjob *)calloc((size_t)P, sizeof(fjob));
    pthread_t *th = (pthread_t *)calloc((size_t)P, sizeof(pthread_t));
    void *shared = mg_comm_loopback_shared_create(P);
    mg_comm_loopback_inject_fault(shared, bad_rank);
    for (int r = 0; r < P; r++) { js[r].rank = r; js[r].shared = shared; pthread_create(&th[r], NULL, fault_work, &js[r]); }
    int bad = 0;
    for (int r = 0; r < P; r++) {
        pthread_join(th[r], NULL);
        const int want = (r == bad_rank) ? MGK_ECOMM : 0;
        if (js[r].rc != want) { fprintf(stderr, "fault run: rank %d returned %d (%s), expected %d\n", r, js[r].rc, js[r].msg, want); bad = 1; }
        if (r == bad_rank && !strstr(js[r].msg, "halo plane")) { fprintf(stderr, "fault run: rank %d does not name the first mismatch: %s\n", r, js[r].msg); bad = 1; }
    }
    mg_comm_loopback_shared_destroy(shared);
    free(js); free(th);
    printf("SAN_FAULT_%s P=%d bad_rank=%d\n", bad ? "FAILED" : "OK", P, bad_rank);
    return bad;
}

/* the peer transport when a neighbour never shows up: rank 0 of 2 exchanges a halo and all-reduces; rank 1 maps the memory and then does
 * nothing.  With MG_PEER_TIMEOUT_S = 1 rank 0 must come back from both calls within seconds and the all-reduce (a host-synchronising hook)
 * must report MGK_ECOMM -- never a hang */
static void *timeout_work(void *p) {
    job *j = (job *)p;
    mgk_geom g; mgk_geom_init(&g, 3, 15, 7, 3);
    mg_comm *cm = mg_comm_peer_create(j->rank, 2, 0, sizeof(double) * (size_t)g.plane, 2, sizeof(double) * (size_t)g.total, j->boot->blobs[j->rank]);
    pthread_barrier_wait(&j->boot->bar);
    j->rc = -1;
    if (!cm || mg_comm_peer_connect(cm, j->boot->blobs)) return NULL;
    pthread_barrier_wait(&j->boot->bar);
    if (j->rank == 0) {
        mgk_ctx *ctx = NULL;
        void *f = NULL;
        double v = 1.0;
        if (mgk_ctx_create(&ctx, 0) || mgk_malloc(ctx, &f, sizeof(double) * (size_t)g.total)) return NULL;
        int r1 = cm->halo(cm, ctx, f, &g, 8, mgk_stream_comm(ctx));
        int r2 = cm->allreduce_sum(cm, ctx, &v, 1, NULL);
        j->rc = (r1 == 0 && r2 == MGK_ECOMM && cm->check(cm) == MGK_ECOMM) ? 0 : 100 + r2;
        mgk_free(ctx, f);
        mgk_ctx_destroy(ctx);
    } else j->rc = 0;
    pthread_barrier_wait(&j->boot->bar);                      /* rank 1 keeps its memory mapped until rank 0 is through */
    mg_comm_destroy(cm);
    return NULL;
}
static int timeout_run(void) {
    setenv("MG_PEER_TIMEOUT_S", "1", 1);
    peer_boot *boot = (peer_boot *)calloc(1, sizeof(peer_boot));
    pthread_barrier_init(&boot->bar, NULL, 2);
    job js[2]; memset(js, 0, sizeof(js));
    pthread_t th[2];
    for (int r = 0; r < 2; r++) { js[r].rank = r; js[r].boot = boot; pthread_create(&th[r], NULL, timeout_work, &js[r]); }
    for (int r = 0; r < 2; r++) pthread_join(th[r], NULL);
    const int bad = js[0].rc || js[1].rc;
    if (bad) fprintf(stderr, "timeout run: rank 0 rc %d, rank 1 rc %d (%s)\n", js[0].rc, js[1].rc, mg_comm_last_error());
    pthread_barrier_destroy(&boot->bar); free(boot);
    printf("SAN_PEER_TIMEOUT_%s\n", bad ? "FAILED" : "OK");
    return bad;
}

int main(int argc, char **argv) {
    if (argc == 2 && !strcmp(argv[1], "peer_timeout")) return timeout_run();
    if (argc == 4 && !strcmp(argv[1], "fault")) return fault_run(atoi(argv[2]), atoi(argv[3]));
    if (argc < 5) { fprintf(stderr, "usage: san_slab P npts levels dist_min_n [mixed|peer] | san_slab fault P bad_rank\n"); return 2; }
    const int P = atoi(argv[1]);
    job base; memset(&base, 0, sizeof(base));
    base.npts = atoi(argv[2]); base.levels = atoi(argv[3]); base.dmin = atoi(argv[4]); base.mixed = argc > 5 && !strcmp(argv[5], "mixed");
    const int use_peer = argc > 5 && !strcmp(argv[5], "peer");        /* the ranks talk through the peer transport instead of loopback */
    job one = base; one.P = 1;
    work(&one);
    if (one.rc) return 1;
    job *js = (job *)calloc((size_t)P, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)P, sizeof(pthread_t));
    void *shared = mg_comm_loopback_shared_create(P);
    peer_boot *boot = NULL;
    if (use_peer) { boot = (peer_boot *)calloc(1, sizeof(peer_boot)); pthread_barrier_init(&boot->bar, NULL, (unsigned)P); }
    for (int r = 0; r < P; r++) { js[r] = base; js[r].rank = r; js[r].P = P; js[r].shared = shared; js[r].boot = boot; pthread_create(&th[r], NULL, work, &js[r]); }
    long off = 0; int bad = 0;
    for (int r = 0; r < P; r++) {
        pthread_join(th[r], NULL);
        if (js[r].rc || js[r].it != one.it) { fprintf(stderr, "rank %d failed (rc %d, %d vs %d cycles)\n", r, js[r].rc, js[r].it, one.it); bad = 1; continue; }
        if (off + js[r].n > one.n || memcmp(js[r].u, one.u + off, sizeof(double) * (size_t)js[r].n)) { fprintf(stderr, "rank %d: solution differs from the single-rank one\n", r); bad = 1; }
        off += js[r].n;
        free(js[r].u);
    }
    if (off != one.n) { fprintf(stderr, "slabs do not add up\n"); bad = 1; }
    mg_comm_loopback_shared_destroy(shared);
    /* one rank of 8 on the phantom communicator (timing aid): the code path only */
    mg_comm *ph = mg_comm_phantom_create(3, 8, 1.0, 50.0);
    mg_config c; job pj = base; pj.dmin = base.dmin; cfg(&c, &pj, 3, 8);
    mg_solver *s = NULL;
    if (!ph || mg_solver_create(&s, &c, ph) || mg_solver_set_rhs_problem(s) || mg_solver_cycles(s, 2) || mg_solver_sync(s)) { fprintf(stderr, "phantom: %s\n", mg_last_error()); bad = 1; }
    if (s) mg_solver_destroy(s);
    if (ph) mg_comm_destroy(ph);
    free(one.u); free(js); free(th);
    if (boot) { pthread_barrier_destroy(&boot->bar); free(boot); }
    printf("SAN_SLAB_%s P=%d cycles=%d%s\n", bad ? "FAILED" : "OK", P, one.it, use_peer ? " transport=peer" : "");
    return bad;
}
