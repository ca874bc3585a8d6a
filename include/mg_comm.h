/*
 * mg_comm.h -- communication hooks of the slab-decomposed V-cycle (nranks > 1).
 *
 * Replaces what PETSc does implicitly under the reference on more than one MPI rank
 * (SURVEY.md 2.3 C1/C2): the VecScatter halo inside every MatMult/KSPSolve and the
 * MPI_Allreduce inside VecNorm.  One process per GPU; planes are contiguous in the padded
 * layout, so a halo is one plane-sized message per neighbour with no pack kernel.
 *
 * Back ends:
 *   rccl      ncclSend/ncclRecv (grouped) + ncclAllReduce over xGMI, librccl loaded with dlopen;
 *             communicator bootstrapped from a 128-byte unique id distributed by the launcher
 *   loopback  all ranks are threads of ONE process sharing one GPU (device-to-device copies and
 *             a pthread barrier): exercises every line of the slab logic on a single-GPU box
 */
#ifndef MG_COMM_H
#define MG_COMM_H
#include "mgk.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mg_comm {
    int rank, nranks;
    void *impl;
    /* fill the z ghost planes of `field` from the neighbouring slabs: my first interior plane ->
     * hi ghost of rank-1, my last interior plane -> lo ghost of rank+1.  Ordered after all work
     * already queued on `stream`; work queued on `stream` afterwards sees the ghosts. */
    int (*halo)(struct mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *g, int esz, void *stream);   /* esz: 8 (fp64) or 4 (fp32) */
    /* `field` has the geometry of the WHOLE level (gfull); rank r produced planes
     * [zstart[r], zstart[r+1]); afterwards every rank holds all planes. */
    int (*allgather_planes)(struct mg_comm *c, mgk_ctx *ctx, void *field, const mgk_geom *gfull,
                            const int *zstart, int esz, void *stream);
    /* in-place sum over ranks of n host doubles (blocking) */
    int (*allreduce_sum)(struct mg_comm *c, mgk_ctx *ctx, double *vals, int n, void *stream);
    int (*barrier)(struct mg_comm *c, mgk_ctx *ctx);
    void (*destroy)(struct mg_comm *c);
} mg_comm;

#define MG_RCCL_ID_BYTES 128
/* rank 0 calls this and ships the bytes to the other ranks (bench.py uses torch.distributed for that) */
int      mg_comm_rccl_unique_id(void *id_out);
mg_comm *mg_comm_rccl_create(int rank, int nranks, const void *id, int device);
/* test aid: grouped ncclSend/ncclRecv with this rank as its own peer (count elements of esz 8 or 4 bytes) */
int  mg_comm_rccl_self_sendrecv(mg_comm *c, mgk_ctx *ctx, const void *src, void *dst, long count, int esz);

/* loopback: create the shared state once, then one handle per rank-thread */
void    *mg_comm_loopback_shared_create(int nranks);
void     mg_comm_loopback_shared_destroy(void *shared);
mg_comm *mg_comm_loopback_create(void *shared, int rank);

const char *mg_comm_last_error(void);
void mg_comm_destroy(mg_comm *c);       /* calls c->destroy */
/* plain-call forms of the hooks (bindings, tests) */
int  mg_comm_halo(mg_comm *c, mgk_ctx *ctx, double *field, const mgk_geom *g);
int  mg_comm_allreduce_sum(mg_comm *c, mgk_ctx *ctx, double *vals, int n);

#ifdef __cplusplus
}
#endif
#endif
