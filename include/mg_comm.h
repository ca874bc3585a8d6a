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
 *             communicator bootstrapped from a 128-byte unique id distributed by the launcher.
 *             EVERY RCCL call of a communicator is issued on the comm stream of the context it is
 *             used with (one communicator, one stream: one total order on every rank); a hook given
 *             any other stream fails with MGK_EINVAL.
 *   loopback  all ranks are threads of ONE process sharing one GPU (device-to-device copies and
 *             a pthread barrier): exercises every line of the slab logic on a single-GPU box
 *   peer      (round 3) IPC-mapped mailboxes + flag words: plane copies by the copy engines, sequenced by one-wave flag kernels; see below
 *   phantom   ONE rank of an N-rank run alone on a GPU: ghost planes are filled by device copies of
 *             the rank's own boundary planes and every exchange holds its stream for
 *             latency + bytes / link bandwidth (mgk_delay_us): times one rank's share of the
 *             N-GPU cycle, exchanges included, on a single GPU.  Results are meaningless.
 */
#ifndef MG_COMM_H
#define MG_COMM_H
#include <stddef.h>
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
    /* ---- optional hooks (NULL: mg_comm_halo_n / the solver fall back to the forms above) ---- */
    /* the halo of `nf` fields (same element size) as ONE exchange: one ncclGroup, one latency */
    int (*halo_n)(struct mg_comm *c, mgk_ctx *ctx, int nf, void *const *fields, const mgk_geom *const *geoms, int esz, void *stream);
    /* in-place sum over ranks of n DEVICE doubles, queued on `stream`: no host synchronisation */
    int (*allreduce_sum_dev)(struct mg_comm *c, mgk_ctx *ctx, double *dvals, int n, void *stream);
    /* (round 3) the general neighbour exchange, ONE group of n slots: slot q sends `bytes[q]` bytes from send_lo[q] to the rank below
     * (which receives them at ITS recv_hi[q]) and from send_hi[q] to the rank above (its recv_lo[q]); a NULL send_lo[q] / recv_hi[q] pair
     * (or send_hi / recv_lo) means the slot carries nothing in that direction -- the same on every rank.  halo_n is the special case
     * send = first / last interior plane, recv = ghost planes; with this hook the solver sends the neighbours' second and third planes
     * straight from the field instead of staging them in the far fields first. */
    int (*exchange)(struct mg_comm *c, mgk_ctx *ctx, int n, const void *const *send_lo, const void *const *send_hi,
                    void *const *recv_lo, void *const *recv_hi, const size_t *bytes, void *stream);
    /* (round 3) has an asynchronous operation of this transport failed since the last call (the peer transport: a flag wait that timed
     * out)?  0 or MGK_ECOMM.  Called by the solver wherever the host has just synchronised with the comm stream, so that a run whose
     * exchanges did not arrive ends with an error instead of numbers computed from stale ghost planes. */
    int (*check)(struct mg_comm *c);
} mg_comm;

#define MG_RCCL_ID_BYTES 128
/* rank 0 calls this and ships the bytes to the other ranks (bench.py uses torch.distributed for that) */
int      mg_comm_rccl_unique_id(void *id_out);
mg_comm *mg_comm_rccl_create(int rank, int nranks, const void *id, int device);
/* test aid: grouped ncclSend/ncclRecv with this rank as its own peer (count elements of esz 8 or 4 bytes) */
int  mg_comm_rccl_self_sendrecv(mg_comm *c, mgk_ctx *ctx, const void *src, void *dst, long count, int esz);
/* the same, queued on the comm stream and left running (measurement aid: tools/rccl_overlap.py) */
int  mg_comm_rccl_self_sendrecv_async(mg_comm *c, mgk_ctx *ctx, const void *src, void *dst, long count, int esz);

/* phantom: rank `rank` of `nranks` alone on its GPU (measurement aid, see above).  lat_us: latency of one exchange,
 * link_gbs: one-directional bandwidth of one link; an exchange holds its stream for lat_us + bytes_per_direction/link_gbs */
mg_comm *mg_comm_phantom_create(int rank, int nranks, double lat_us, double link_gbs);

/* peer (round 3): every rank owns a mailbox, a gather box and a block of flag words in fine-grained device memory, exported with hipIpc*
 * and mapped by its neighbours; a halo is a peer copy of the boundary planes into the neighbour's mailbox (copy engines between devices: no
 * workgroup), sequenced by ONE-wave flag kernels -- nothing that has to find a free CU beside the one-block-per-CU marching kernels, which
 * is what a send/recv kernel must.  All-reduce: 8-byte stores into every rank's slot block, summed in rank order.
 * At most MGK_PEER_MAX (16) ranks of ONE node.  plane_bytes_max / fields_max: largest plane and most fields of one grouped exchange;
 * gather_bytes: (nz + 2) planes of the largest level that is all-gathered.  blob_out receives this rank's MG_PEER_BLOB_BYTES of handles;
 * the launcher all-gathers the blobs of all ranks (rank order) and passes them to mg_comm_peer_connect on every rank.
 * Every rank must issue the same sequence of exchanges.  MG_PEER_TIMEOUT_S (default 60): a flag that does not arrive in time makes the
 * next host-synchronising hook (allreduce_sum, barrier) fail with MGK_ECOMM instead of hanging the stream.
 * Tested with two and three PROCESSES sharing one GPU (IPC handles of the same device); its bandwidth over xGMI is unmeasured. */
#define MG_PEER_BLOB_BYTES (3 * MGK_IPC_HANDLE_BYTES)
mg_comm *mg_comm_peer_create(int rank, int nranks, int device, size_t plane_bytes_max, int fields_max, size_t gather_bytes, void *blob_out);
int      mg_comm_peer_connect(mg_comm *c, const void *all_blobs);

/* First-run gate of a transport: rank-coded planes through halo / halo_n / allgather_planes, known sums through both
 * all-reduce forms, everything read back and compared on every rank.  0, or MGK_ECOMM with mg_comm_last_error() naming the
 * first mismatch.  Collective: every rank of the communicator must call it -- and every rank RETURNS from it: a rank that
 * sees a mismatch or a failing hook keeps taking part in the remaining collectives of the fixed sequence and reports at the
 * end, so that a fault on one rank never leaves its peers inside a send/recv (they return 0 if what THEY saw was right). */
int  mg_comm_selftest(mg_comm *c, mgk_ctx *ctx);

/* loopback: create the shared state once, then one handle per rank-thread */
void    *mg_comm_loopback_shared_create(int nranks);
void     mg_comm_loopback_shared_destroy(void *shared);
mg_comm *mg_comm_loopback_create(void *shared, int rank);
/* test aid: rank `rank` (>= 1) receives a wrong plane as its lo ghost from now on (-1: off) -- what the gate must catch */
void     mg_comm_loopback_inject_fault(void *shared, int rank);

const char *mg_comm_last_error(void);
void mg_comm_destroy(mg_comm *c);       /* calls c->destroy */
/* plain-call forms of the hooks (bindings, tests) */
int  mg_comm_halo(mg_comm *c, mgk_ctx *ctx, double *field, const mgk_geom *g);
int  mg_comm_allreduce_sum(mg_comm *c, mgk_ctx *ctx, double *vals, int n);
/* halo of several fields in one exchange where the back end can (halo_n), else one after the other */
int  mg_comm_halo_n(mg_comm *c, mgk_ctx *ctx, int nf, void *const *fields, const mgk_geom *const *geoms, int esz, void *stream);

#ifdef __cplusplus
}
#endif
#endif
