/*
 * mgk.h -- kernel-level C ABI of the MI355X multigrid V-cycle hot path.
 *
 * This is the drop-in boundary UNDER the PETSc-surface shim (include/petscksp.h):
 * plain C, opaque context, device pointers and sizes, no C++/torch types.
 * One entry point per PETSc operation that the reference executes inside
 * MultigridVcycle (src/solver.c:1414-1575); the reference call each one
 * replaces is cited at its declaration (paths relative to /root/reference).
 *
 * Conventions
 *   - every function returns 0 on success, otherwise a nonzero code
 *     (hipError_t value or MGK_E*), never throws, never allocates on the hot
 *     path; mgk_last_error() returns a static description of the last failure.
 *   - `stream` is a hipStream_t passed as void*; NULL selects the context's
 *     compute stream.  Kernels are asynchronous on their stream.
 *   - device arrays use the padded layout described by mgk_geom (below).
 *   - arithmetic: IEEE fp64, no fused multiply-add, terms summed in ascending
 *     column order of the assembled row (see DESIGN.md "canonical arithmetic").
 *
 * Device layout of a level field (one rank's slab)
 *   interior unknown (k,i,j), k plane (z), i row (y), j column (x):
 *       offset = org + k*plane + i*pitch + j
 *   with one ghost row / plane on every side (i=-1, i=ny, k=-1, k=nz) and the
 *   x ghosts inside the row padding (j=-1, j=nx).  Ghosts of a global boundary
 *   hold 0 (homogeneous Dirichlet, src/solver.c:239-251 drops those
 *   neighbours); ghost planes of an internal slab boundary hold halo data.
 *   The row starts 128-byte aligned; j=0 sits at column MGK_XOFF so that the
 *   nx+1 = 2^m doubles (interior + right ghost) of a row are whole 128-B lines.
 */
#ifndef MGK_H
#define MGK_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MGK_XOFF 16          /* column of interior x=0 inside a padded row */
#define MGK_EINVAL  10001    /* bad argument / shape */
#define MGK_ENOGPU  10002    /* no usable HIP device: the product path never falls back to the CPU */
#define MGK_ECOMM   10003    /* RCCL failure */

typedef struct mgk_ctx mgk_ctx;

typedef struct mgk_geom {
    int  dim;            /* 2 or 3 */
    int  nx, ny, nz;     /* local interior unknowns per axis (2-D: nz = 1, no ghost planes) */
    int  pitch;          /* doubles per padded row */
    long plane;          /* doubles per padded plane = pitch*(ny+2) */
    long org;            /* offset of interior (0,0,0) */
    long total;          /* doubles to allocate */
} mgk_geom;

/* host helper: fill a geometry for nx x ny (x nz) local unknowns.  returns 0 or MGK_EINVAL. */
int  mgk_geom_init(mgk_geom *g, int dim, int nx, int ny, int nz);

/* ---- context, memory, streams (thin wrappers so that host code stays C) ---- */
int  mgk_device_count(void);
int  mgk_set_device(int device);                                  /* hipSetDevice for the calling thread */
int  mgk_ctx_create(mgk_ctx **ctx, int device);
void mgk_ctx_destroy(mgk_ctx *ctx);
const char *mgk_last_error(void);
/* > 0: the 3-D marching kernels launched on this context cut z into chunks of about `planes` planes instead of the long streams
 * a single GPU prefers -- a slab rank sets it so that the workgroups of an exchange (RCCL send/recv kernels on the high-priority
 * comm stream) are dispatched within one short block's time instead of behind a chip-filling kernel; 0: built-in choice */
int  mgk_ctx_set_chunk_planes(mgk_ctx *ctx, int planes);
void *mgk_stream_compute(mgk_ctx *ctx);
void *mgk_stream_comm(mgk_ctx *ctx);
int  mgk_malloc(mgk_ctx *ctx, void **dptr, size_t bytes);          /* zero-filled */
int  mgk_free(mgk_ctx *ctx, void *dptr);
int  mgk_memset0(mgk_ctx *ctx, void *dptr, size_t bytes, void *stream);   /* VecSet(x,0.0) solver.c:1514 */
int  mgk_h2d(mgk_ctx *ctx, void *dst, const void *src, size_t bytes);     /* synchronous */
int  mgk_d2h(mgk_ctx *ctx, void *dst, const void *src, size_t bytes);     /* synchronous */
int  mgk_d2d(mgk_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream);
int  mgk_sync(mgk_ctx *ctx, void *stream);                         /* NULL: whole device */
/* pinned host memory + stream-ordered copies: a reduction result leaves the device without a device-wide synchronisation */
int  mgk_host_alloc(mgk_ctx *ctx, void **hptr, size_t bytes);      /* zero-filled */
int  mgk_host_free(mgk_ctx *ctx, void *hptr);
int  mgk_d2h_async(mgk_ctx *ctx, void *dst_pinned, const void *src, size_t bytes, void *stream);
int  mgk_h2d_async(mgk_ctx *ctx, void *dst, const void *src_pinned, size_t bytes, void *stream);
/* one wavefront busy-waits `us` microseconds on `stream` (nothing else is touched): stands for the time a halo plane spends
 * on a link when one rank's share of an N-GPU run is timed on one GPU (mg_comm phantom back end); never on the product path */
int  mgk_delay_us(mgk_ctx *ctx, double us, void *stream);
/* copies `bytes` (multiple of 16) with `blocks` small workgroups that then stay resident until `us` microseconds have passed: the
 * footprint of a send/recv kernel that moves a plane at link speed; phantom back end only */
int  mgk_paced_copy(mgk_ctx *ctx, void *dst, const void *src, size_t bytes, double us, int blocks, void *stream);
/* stream-ordered event timing for bench.py's roofline leg */
int  mgk_timer_create(mgk_ctx *ctx, void **timer);
int  mgk_timer_start(mgk_ctx *ctx, void *timer, void *stream);
int  mgk_timer_stop(mgk_ctx *ctx, void *timer, void *stream);
int  mgk_timer_elapsed_ms(mgk_ctx *ctx, void *timer, double *ms);  /* synchronises on the stop event */
void mgk_timer_destroy(mgk_ctx *ctx, void *timer);
/* cross-stream dependency: work queued later on `waiter` starts after everything queued so far on `signaller` */
int  mgk_stream_wait(mgk_ctx *ctx, void *waiter, void *signaller);

/* HIP-graph capture of everything queued on the compute stream between begin and end (kernels, async memsets;
 * no synchronous call may happen in between); the executable graph replays on the compute stream */
int  mgk_capture_begin(mgk_ctx *ctx);
int  mgk_capture_end(mgk_ctx *ctx, void **graph_exec);
int  mgk_graph_launch(mgk_ctx *ctx, void *graph_exec);
void mgk_graph_destroy(mgk_ctx *ctx, void *graph_exec);

/* compact lexicographic (k*ny+i)*nx+j  <->  padded layout (VecGetArray/VecSetValue side, src/solver.c:588-617,1255) */
int  mgk_pack_f64(mgk_ctx *ctx, const mgk_geom *g, const double *compact_dev, double *padded_dev, void *stream);
int  mgk_unpack_f64(mgk_ctx *ctx, const mgk_geom *g, const double *padded_dev, double *compact_dev, void *stream);

/* ---- K1: KSPSolve, KSPRICHARDSON + PCJACOBI, one sweep (src/solver.c:1531,1536,1542) ----
 * coef: ascending-column stencil, 2-D {(i-1),(j-1),C,(j+1),(i+1)}  (OpA, src/problem.c:3-22;
 *       row fill src/solver.c:239-251), 3-D {(k-1),(i-1),(j-1),C,(j+1),(i+1),(k+1)}.
 * unew = u + scale*((b - A u)*dinv);  u and unew must be different buffers. */
int  mgk_jacobi_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                    const double *b, const double *u, double *unew, void *stream);
/* the same sweep restricted to the marching planes [zbeg, zend) (3-D: z planes, 2-D: rows): lets a slab rank
 * update its two boundary planes first, ship them, and sweep the interior while the halo travels */
int  mgk_jacobi_range_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                          const double *b, const double *u, double *unew, int zbeg, int zend, void *stream);
/* first sweep after KSPSolve zero-filled the guess: unew = scale*(b*dinv), u is not read */
int  mgk_jacobi_zero_f64(mgk_ctx *ctx, const mgk_geom *g, double dinv, double scale,
                         const double *b, double *unew, void *stream);
/* KSPCHEBYSHEV recurrence step: pkp1 = (c_km1*pkm1 + c_k*pk) + c_z*((b - A pk)*dinv) */
int  mgk_cheby_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv,
                   double c_km1, double c_k, double c_z,
                   const double *b, const double *pk, const double *pkm1, double *pkp1, void *stream);

/* ---- K2/K5: KSPBuildResidual / MatMult+VecAXPY (src/solver.c:1516-1517,1534,1545): r = b - A u ---- */
int  mgk_residual_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef,
                      const double *b, const double *u, double *r, void *stream);
/* K2+K6 fused: sum over the slab of (b - A u)^2, r is not written.  *sumsq_host is valid after return. */
int  mgk_residual_sumsq_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef,
                            const double *b, const double *u, double *sumsq_host, void *stream);
/* TWO Jacobi sweeps in one pass, unew = J(J(u)) (temporal blocking: the intermediate field lives in LDS only); 3-D, whole
 * grids, nx + 1 <= 1024; bit-identical to two mgk_jacobi_f64 calls */
int  mgk_jacobi2_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                     const double *b, const double *u, double *unew, void *stream);
int  mgk_jacobi2_f32(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                     const float *b, const float *u, float *unew, void *stream);
/* 2-D version (constant coefficients; x tiles overlap by one lane, y chunks recompute two rows of the first sweep) */
int  mgk_jacobi2_2d_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                        const double *b, const double *u, double *unew, void *stream);
/* the same on a z-slab: `far` is a field of geometry gfar = (nx, ny, nz = 2) whose lo / hi ghost planes hold plane nz-2 of the
 * rank below / plane 1 of the rank above (exchange it like any field after copying this rank's planes 1 and nz-2 into its two
 * interior planes); u's ghost planes hold the neighbours' last / first plane, b's ghost planes their b */
int  mgk_jacobi2_slab_f64(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                          const double *b, const double *u, double *unew, const double *far, int has_lo, int has_hi,
                          int zbeg, int zend, void *stream);   /* output planes [zbeg, zend): planes 2 .. nz-3 need no ghost data */
int  mgk_jacobi2_slab_f32(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                          const float *b, const float *u, float *unew, const float *far, int has_lo, int has_hi,
                          int zbeg, int zend, void *stream);
/* one Jacobi sweep unew = u + scale*dinv*(b - A u) that also returns ||b - A u||^2 (residual of the INPUT u): the norm
 * that closes a cycle (src/solver.c:1545-1546) fused with the first sweep of the next one (:1531) */
int  mgk_jacobi_sumsq_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                          const double *b, const double *u, double *unew, double *sumsq_host, void *stream);

/* the same on the marching planes [zbeg, zend): per-block partial sums land in the context's partial buffer from slot
 * part_off on (*nparts slots) and are NOT reduced; after the last range call mgk_partials_finish(nparts_total) reduces
 * slots 0 .. nparts_total-1 in a fixed order.  Lets a slab rank sweep its interior while the ghost planes travel. */
int  mgk_jacobi_sumsq_range_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                                const double *b, const double *u, double *unew, int zbeg, int zend,
                                int part_off, int *nparts, void *stream);
int  mgk_partials_finish(mgk_ctx *ctx, int nparts, double *sumsq_host, void *stream);

/* residual on the planes (3-D) / rows (2-D) [zbeg, zend) only */
int  mgk_residual_range_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, const double *b, const double *u, double *r,
                            int zbeg, int zend, void *stream);
int  mgk_residual_range_f32(mgk_ctx *ctx, const mgk_geom *g, const double *coef, const float *b, const float *u, float *r,
                            int zbeg, int zend, void *stream);
/* z-slab of a rank that is not the last (nzf = 2 nzc): mgk_residual_restrict_* leaves the last coarse plane partial (its
 * dk = 0, 1 terms); this adds the dk = 2 terms from r's hi ghost plane (the neighbour's first residual plane) in the order
 * of the row of res, so the result equals the whole-grid sum bit for bit */
int  mgk_restrict_finish_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *r, double *bc, void *stream);
int  mgk_restrict_finish_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const float *r, float *bc, void *stream);
/* K2+K3 fused: b_coarse = R (b - A u), the fine residual is never written (src/solver.c:1534-1535).
 * Whole grids and last slabs (gf->nz == 2*gc->nz + 1) complete; inner z-slabs (gf->nz == 2*gc->nz) leave the last coarse plane
 * partial, to be closed by mgk_restrict_finish_* (above).  3-D. */
int  mgk_residual_restrict_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                               const double *b, const double *u, double *bc, void *stream);
/* the same for the coarse planes [kcbeg, kcend) only (reads the fine planes 2 kcbeg - 1 .. 2 kcend + 1): a slab rank restricts
 * its inner coarse planes, which need no ghost plane of u, while the halo travels */
int  mgk_residual_restrict_range_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                                     const double *b, const double *u, double *bc, int kcbeg, int kcend, void *stream);
int  mgk_residual_restrict_range_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                                     const float *b, const float *u, float *bc, int kcbeg, int kcend, void *stream);
/* z-slab form without the partial plane: `far` (geometry gfar = (nx, ny, 2), as for mgk_jacobi2_slab_*) carries plane 1 of the rank
 * above in its hi ghost plane, u and b have valid hi ghost planes; an inner slab (has_hi) then evaluates the residual of the
 * neighbour's first plane itself -- same operands and arithmetic as its owner, same bits -- and completes its last coarse plane:
 * one exchange before the kernel instead of two dependent ones and a finishing kernel.  has_hi == 0: the last slab / a whole grid */
int  mgk_residual_restrict_slab_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *coef,
                                    const double *b, const double *u, const double *far, int has_hi, double *bc,
                                    int kcbeg, int kcend, void *stream);
int  mgk_residual_restrict_slab_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *coef,
                                    const float *b, const float *u, const float *far, int has_hi, float *bc,
                                    int kcbeg, int kcend, void *stream);
/* the same (whole grids), also writing the coarse level's first sweep from a zero guess, uc0 = scale_c * (bc * dinv_c): what
 * mgk_jacobi_zero_* would compute from bc */
int  mgk_residual_restrict_jz_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const double *b,
                                  const double *u, double *bc, double *uc0, double dinv_c, double scale_c, void *stream);
int  mgk_residual_restrict_jz_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const float *b,
                                  const float *u, float *bc, float *uc0, double dinv_c, double scale_c, void *stream);

/* 2-D form of K2+K3 (constant coefficients): b_coarse = R (b - A u) in one pass, optionally also the coarse level's first sweep from
 * a zero guess (uc0 = scale_c * (bc * dinv_c); NULL: not written).  Bit-identical to mgk_residual_f64 + mgk_restrict_fw_f64. */
int  mgk_residual_restrict_2d_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const double *b,
                                  const double *u, double *bc, double *uc0, double dinv_c, double scale_c, void *stream);

/* ---- K3: MatMult(res[l], r, b[l+1]) full weighting (src/solver.c:1535, matrix :1071-1092) ----
 * coarse (kc,ic,jc) gathers fine (2kc+dk, 2ic+di, 2jc+dj), d in {0,1,2}.  gc->nz coarse planes are
 * produced from fine planes 0..2*gc->nz (plane gf->nz is the fine ghost plane in the slab case). */
int  mgk_restrict_fw_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc,
                         const double *rf, double *bc, void *stream);

/* ---- K4: MatMult(pro[l], u[l+1], rv) + VecAXPY(u[l],1.0,rv) (src/solver.c:1540-1541, matrix :1131-1152) ---- */
int  mgk_prolong_add_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc,
                         const double *uc, double *uf, void *stream);

/* The I-cycle's coupled level operator for G grids in one level (src/solver.c:489-510; 2-D, square grids, constant stencils):
 * lower triangle + diagonal as the cascade s_0 = A_0 x_0, s_g = R s_(g-1) + A_g x_g (mgk_apply_f64, mgk_restrict_fw_f64 and)
 *   mgk_apply_add_f64:   y += A x
 * and the upper blocks (fillProlongationPortion :347-470: A_g1 P^(g0-g1) cut to P's window) by
 *   mgk_window_add_f64:  yf(i,j) += sum over the coarse points (ic,jc) with |i - (S ic + S-1)|, |j - (S jc + S-1)| <= S-1 of
 *                        wtab[(i - S ic)*(2S-1) + (j - S jc)] * xc(ic,jc);  S = stride = 2^(g0-g1), nf + 1 = S (nc + 1);
 *                        wtab: (2S-1)^2 DEVICE doubles; xc with a zero ghost ring. */
int  mgk_apply_add_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, const double *x, double *y, void *stream);
int  mgk_window_add_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, int stride, const double *wtab_dev,
                        const double *xc, double *yf, void *stream);

/* KSPBuildResidual + VecNorm + the first sweep of the KSPSolve that follows (src/solver.c:1545-1546,1531) in one pass over u and b (2-D):
 * r = b - A u (stored), *sumsq_host = sum r^2, unew = u + scale*(r*dinv).  coef or (ctab, dtab: row tables, then dinv is ignored). */
int  mgk_jacobi_sumsq_store_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                                const double *b, const double *u, double *unew, double *r, double *sumsq_host, void *stream);

/* y = B x, B dense row-major m x n on the device: the exact coarse-grid solve of PCMG (PCLU is PETSc's default coarse solver,
 * src/solver.c:1931-1932), B = A^-1 of the coarsest grid inverted once on the host; x, y compact (unpadded) device arrays */
int  mgk_dense_mult_f64(mgk_ctx *ctx, int m, int n, const double *B_dev, const double *x, double *y, void *stream);

/* K4 fused into the first post-smoothing sweep (src/solver.c:1540-1542): unew = Jacobi(u + P uc); the corrected
 * u is never written.  Needs valid z ghost planes of u AND of uc on a slab.  3-D only. */
int  mgk_prolong_jacobi_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv,
                            double scale, const double *b, const double *uc, const double *u, double *unew, void *stream);
int  mgk_prolong_jacobi_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv,
                            double scale, const float *b, const float *uc, const float *u, float *unew, void *stream);
/* output planes [zbeg, zend) only.  On an inner slab the planes 2 .. nz-2 read neither a ghost plane of u nor one of uc */
int  mgk_prolong_jacobi_range_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv,
                                  double scale, const double *b, const double *uc, const double *u, double *unew,
                                  int zbeg, int zend, void *stream);
int  mgk_prolong_jacobi_range_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv,
                                  double scale, const float *b, const float *uc, const float *u, float *unew,
                                  int zbeg, int zend, void *stream);

/* ---- the tail of the hierarchy in one kernel (src/solver.c:1533-1544 restricted to the levels t .. L-1) ----
 * One 1024-lane workgroup keeps u, its ping-pong partner and b of the `nlev` coarsest levels in LDS (n[0] <= mgk_tail_max_n(dim):
 * 15 in 3-D, 63 in 2-D; n[l-1] = 2 n[l] + 1) and runs: v0 sweeps from a zero guess on the first level (v1 on the last one),
 * residual + full weighting + sweeps down to the last level, then prolongation + v0 sweeps back up; Richardson + Jacobi with
 * the level's 7 (2-D: 5 used; layout of mgk_jacobi_f64's coef, 7 doubles per level) coefficients and 1/diag.  Every operation
 * evaluates the expressions of the kernel it replaces, so the result is bit-identical to the kernel-per-operation loop.
 * b: right-hand side of the first tail level (padded field of geometry g0), u: its solution after the post-smoothing. */
int  mgk_tail_cycle_f64(mgk_ctx *ctx, const mgk_geom *g0, int nlev, const int *n, const double *coef7, const double *dinv,
                        double scale, int v0, int v1, const double *b, double *u, void *stream);
int  mgk_tail_cycle_f32(mgk_ctx *ctx, const mgk_geom *g0, int nlev, const int *n, const double *coef7, const double *dinv,
                        double scale, int v0, int v1, const float *b, float *u, void *stream);
/* ... with its own damping factor on the COARSEST level (2-D fp64; either coef7 + dinv or the row tables ctab + dtab, the other pair NULL):
 * PCMG's exact coarse solve on a 1 x 1 grid is one undamped Jacobi sweep from the zero guess (v1 = 1, coarse_scale = 1) */
int  mgk_tail_cycle_cs_f64(mgk_ctx *ctx, const mgk_geom *g0, int nlev, const int *n, const double *coef7, const double *dinv,
                           const double *const *ctab, const double *const *dtab, double scale, double coarse_scale, int v0, int v1,
                           const double *b, double *u, void *stream);
int  mgk_tail_max_n(int dim);
/* profiling aid: the tail kernels this thread launches deposit (s_memrealtime [100 MHz], s_memtime [shader clock]) pairs at each of their
 * barriers into dev[0 .. 255] and the number of pairs into dev[256] (257 long longs of DEVICE memory; NULL switches it off) */
void mgk_debug_tail_stamps(long long *dev);

/* ---- K6: VecNorm(NORM_2) (src/solver.c:1512,1518,1546): returns sum of squares of the interior ---- */
int  mgk_sumsq_f64(mgk_ctx *ctx, const mgk_geom *g, const double *x, double *sumsq_host, void *stream);

/* ---- setup-side device helpers ---- */
/* b(k,i,j) = (cx[j]*sy[i])*sz[k]  (levelvecb + Ffunc, src/solver.c:586-594, src/problem.c:24-28);
 * cx/sy/sz are device arrays of nx/ny/nz doubles (2-D: sz unused) */
int  mgk_fill_separable_f64(mgk_ctx *ctx, const mgk_geom *g, const double *cx, const double *sy,
                            const double *sz, double *out, void *stream);
/* GetError (src/solver.c:1211-1237) against sol = (sx[j]*sy[i])*sz[k]: err3_host = {max|e|, sum|e|, sum e^2} */
int  mgk_error_sums_f64(mgk_ctx *ctx, const mgk_geom *g, const double *u, const double *sx,
                        const double *sy, const double *sz, double *err3_host, void *stream);

/* y = A x (MatMult on a stencil operator; src/solver.c:1516) */
int  mgk_apply_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, const double *x, double *y, void *stream);

/* 2-D operators whose coefficients depend on the grid row only: the reference's stretched meshes (-mesh 1/2; metrics are
 * functions of y, src/mesh.c:45-107, OpA src/problem.c:3-22, row fill src/solver.c:231-251).  ctab: ny x 5 device doubles
 * {(i-1),(j-1),C,(j+1),(i+1)} per grid row; dtab: ny device doubles 1/diag.  mode 0: unew = u + scale*((b-Au)*dtab[i]);
 * mode 1: out = b - A u; mode 4: out = A u. */
int  mgk_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *g, int mode, const double *ctab, const double *dtab, double scale,
                     const double *b, const double *u, double *out, void *stream);
/* Chebyshev step on the same operators (KSPCHEBYSHEV + PCJACOBI on -mesh 1/2): pkp1 = (c_km1*pkm1 + c_k*pk) + c_z*((b - A pk)*dtab[i]) */
int  mgk_cheby_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *g, const double *ctab, const double *dtab, double c_km1, double c_k, double c_z,
                           const double *b, const double *pk, const double *pkm1, double *pkp1, void *stream);
int  mgk_jacobi_zero_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *g, const double *dtab, double scale,
                                 const double *b, double *unew, void *stream);
/* the fused forms of the cycle on a row-table operator (the constant-coefficient entry points of the same names with the
 * tables in place of coef / dinv; same arithmetic per point, so a stretched-mesh cycle stays bit-identical to the
 * kernel-per-operation one): */
int  mgk_jacobi2_2d_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *g, const double *ctab, const double *dtab, double scale,
                                const double *b, const double *u, double *unew, void *stream);          /* mgk_jacobi2_2d_f64 */
int  mgk_jacobi_sumsq_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *g, const double *ctab, const double *dtab, double scale,
                                  const double *b, const double *u, double *unew, double *sumsq_host, void *stream);   /* mgk_jacobi_sumsq_f64 */
int  mgk_residual_sumsq_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *g, const double *ctab, const double *b, const double *u,
                                    double *sumsq_host, void *stream);                                  /* mgk_residual_sumsq_f64 */
int  mgk_prolong_jacobi_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *ctab, const double *dtab,
                                    double scale, const double *b, const double *uc, const double *u, double *unew, void *stream);   /* mgk_prolong_jacobi_f64 */
/* ctab_f: table of the FINE level; dtab_c: 1/diag table of the COARSE level (needed only with uc0) */
int  mgk_residual_restrict_2d_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *ctab_f, const double *b,
                                          const double *u, double *bc, double *uc0, const double *dtab_c, double scale_c, void *stream);
/* ctab[l] / dtab[l]: the device tables of tail level l (n[l] x 5 and n[l] doubles) */
int  mgk_jacobi2_2d_sumsq_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *g, const double *ctab, const double *dtab, double scale,
                                      const double *b, const double *u, double *unew, double *sumsq_host, void *stream);      /* mgk_jacobi2_2d_sumsq_f64 */
/* ctab_f / dtab_f: tables of the FINE level; dtab_c: 1/diag table of the COARSE level (needed only with uc0) */
int  mgk_sweep_residual_restrict_2d_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *ctab_f, const double *dtab_f,
                                                double scale, const double *b, const double *u, double *unew, double *bc, double *uc0,
                                                const double *dtab_c, double scale_c, void *stream);
int  mgk_tail_cycle_rowcoef_f64(mgk_ctx *ctx, const mgk_geom *g0, int nlev, const int *n, const double *const *ctab,
                                const double *const *dtab, double scale, int v0, int v1, const double *b, double *u, void *stream);

/* two sweeps in one pass and sumsq = || b - A J(u) ||^2: the residual of the FIRST sweep's output J(u), which is NOT stored --
 * the norm that closes a cycle whose last post-smoothing sweep is this pass's first one (the caller keeps u and owes that sweep
 * if the iteration stops).  Shapes as mgk_jacobi2_sumsq_f64. */
int  mgk_jacobi2_sumsq_mid_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                               const double *b, const double *u, double *unew, double *sumsq_host, void *stream);
/* the prolongation, its correction and the first TWO post-smoothing sweeps in one pass: unew = J(J(u + P uc))
 * (src/solver.c:1540-1542).  fp64, rows of 512 / 1024 (n = 511, 1023), whole grid; _ok_ tells. */
int  mgk_prolong_jacobi2_ok_f64(const mgk_geom *gf, const mgk_geom *gc);
int  mgk_prolong_jacobi2_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                             const double *b, const double *uc, const double *u, double *unew, void *stream);
/* THREE sweeps from a zero initial guess in one pass that reads b alone (the first one is pointwise: mgk_jacobi_zero_*):
 * unew = J(J(J0(b))) -- the whole of a pre-smoothing KSPSolve with max_it = 3 on a coarse level (src/solver.c:1536), 16 B per unknown
 * (fp32: 8) instead of 8 + 24.  Whole 3-D grids of full-row shape: fp32 n = 255 .. 1023, fp64 n = 127 .. 1023; _ok_ tells (1 / 0). */
int  mgk_jacobi2_zero_ok_f64(const mgk_geom *g);
int  mgk_jacobi2_zero_ok_f32(const mgk_geom *g32);
int  mgk_jacobi2_zero_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, double *unew, void *stream);
int  mgk_jacobi2_zero_f32(mgk_ctx *ctx, const mgk_geom *g32, const double *coef, double dinv, double scale, const float *b, float *unew, void *stream);
/* two sweeps in one pass AND sumsq = || b - A u ||^2 of the INPUT field (the first sweep forms that residual anyway): the norm
 * that closes cycle k (src/solver.c:1545-1546) out of the pass that makes the first two pre-smoothing sweeps of cycle k+1
 * (:1531).  fp64, full-row 3-D shapes (n = 127, 255, 511, 1023), whole grid; mgk_jacobi2_sumsq_ok_f64 tells (1 / 0). */
int  mgk_jacobi2_sumsq_ok_f64(const mgk_geom *g);
int  mgk_jacobi2_sumsq_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                           const double *b, const double *u, double *unew, double *sumsq_host, void *stream);
/* The LAST pre-smoothing sweep fused with the residual and its full weighting (src/solver.c:1531, last Richardson iteration,
 * + :1534-1535): unew = J(u); bc = R (b - A unew); uc0 (optional) = scale_c * (bc * dinv_c) -- 26 B per fine unknown instead of
 * 24 + 18 B for the sweep and the fused residual+restriction as two passes.  Built for full-row 3-D shapes (n = 127, 255, 511,
 * 1023), whole grid; mgk_sweep_residual_restrict_ok_f64 tells (1 / 0). */
int  mgk_sweep_residual_restrict_ok_f64(const mgk_geom *gf, const mgk_geom *gc);
int  mgk_sweep_residual_restrict_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                     const double *b, const double *u, double *unew, double *bc, double *uc0,
                                     double dinv_c, double scale_c, void *stream);
/* the 2-D forms of the two (any vertex-centred 2-D grid): */
int  mgk_jacobi2_2d_sumsq_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                              const double *b, const double *u, double *unew, double *sumsq_host, void *stream);
int  mgk_sweep_residual_restrict_2d_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                        const double *b, const double *u, double *unew, double *bc, double *uc0,
                                        double dinv_c, double scale_c, void *stream);
/* the two on a z-slab of a multi-GPU run.  far / far2 / bfar: fields of geometry gfar = (nx, ny, 2) (as for mgk_jacobi2_slab_f64) whose
 * ghost planes hold, after a halo exchange, -- far: lo = plane nz-2 of the rank below, hi = plane 1 of the rank above (the
 * sender puts its planes 1 and nz-2 into the two interior planes); far2: hi = plane 2 of u of the rank above (sender: interior
 * plane 0 = its plane 2); bfar: hi = plane 1 of b of the rank above (sender: interior plane 0 = its b plane 1).  u's and b's own ghost
 * planes must be valid.  A slab with a rank above (has_hi) has nzf = 2 nzc and completes its last coarse plane itself; the last one
 * has nzf = 2 nzc + 1.  [kcbeg, kcend): coarse planes of this launch; [zbeg, zend) / part_off / nparts as mgk_jacobi_sumsq_range_f64. */
int  mgk_jacobi2_sumsq_slab_f64(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                                const double *b, const double *u, double *unew, const double *far, int has_lo, int has_hi,
                                int zbeg, int zend, int part_off, int *nparts, void *stream);
/* (round 3, second session) the 91-byte fine level on z-slabs: mgk_jacobi2_sumsq_mid_f64 and mgk_prolong_jacobi2_f64 on the planes [zbeg, zend) of
 * a slab.  The prolongation pass also reads uc's ghost planes (the neighbours' coarse boundary planes) and, from `cfar` (geometry gcfar = (nxc, nyc, 2)),
 * the rank below's coarse plane nzc-2 (its lo ghost plane); u's ghost planes and `far` hold the neighbours' planes BEFORE the correction -- every rank
 * corrects and sweeps the planes -2 .. nz+1 it needs itself (same operands, same arithmetic as their owner: same bits).  zbeg must be even.
 * Replaces the same reference calls as the whole-grid forms (src/solver.c:1540-1542, :1545-1546 + :1531). */
int  mgk_jacobi2_sumsq_mid_slab_f64(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                                    const double *b, const double *u, double *unew, const double *far, int has_lo, int has_hi,
                                    int zbeg, int zend, int part_off, int *nparts, void *stream);
int  mgk_prolong_jacobi2_slab_ok_f64(const mgk_geom *gf, const mgk_geom *gc, int has_hi);
int  mgk_prolong_jacobi2_slab_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const mgk_geom *gcfar, const double *coef,
                                  double dinv, double scale, const double *b, const double *uc, const double *u, double *unew,
                                  const double *far, const double *cfar, int has_lo, int has_hi, int zbeg, int zend, void *stream);
int  mgk_sweep_residual_restrict_slab_ok_f64(const mgk_geom *gf, const mgk_geom *gc);
int  mgk_sweep_residual_restrict_slab_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *coef,
                                          double dinv, double scale, const double *b, const double *u, double *unew,
                                          const double *far, const double *far2, const double *bfar, int has_lo, int has_hi,
                                          double *bc, int kcbeg, int kcend, void *stream);

/* ---- flat BLAS-1 / AIJ kernels behind the PETSc-surface shim (include/petscksp.h) ----
 * n counts doubles of a whole allocation (padded fields: ghosts are 0 and stay 0). */
int  mgk_flat_axpy(mgk_ctx *ctx, long n, double a, const double *x, double *y, void *stream);       /* VecAXPY  y += a x (src/solver.c:1517,1541) */
int  mgk_flat_aypx(mgk_ctx *ctx, long n, double a, const double *x, double *y, void *stream);       /* VecAYPX  y = x + a y */
int  mgk_flat_axpbypcz(mgk_ctx *ctx, long n, double a, double b, double g, const double *x, const double *y, double *z, void *stream);
int  mgk_flat_fill(mgk_ctx *ctx, long n, double a, double *z, void *stream);                        /* VecSet (src/solver.c:1514) */
int  mgk_flat_scale(mgk_ctx *ctx, long n, double a, double *z, void *stream);
int  mgk_flat_pointwise_mult(mgk_ctx *ctx, long n, const double *x, const double *y, double *z, void *stream);
/* bandwidth probe: a = b + s*c over n doubles (n even, 16-byte aligned), `blocks` workgroups of 1024 lanes */
int  mgk_stream_triad_f64(mgk_ctx *ctx, long n, double *a, const double *b, const double *c, double s,
                          int blocks, int nontemporal, void *stream);
/* deferred reductions: while `slot_dev` is non-NULL every single-value reduction below (…_sumsq_…, mgk_flat_dot, the fused
 * residual norms) writes its result to that device double in stream order and returns 0.0 without synchronising */
int  mgk_defer_result(mgk_ctx *ctx, double *slot_dev);
int  mgk_flat_dot(mgk_ctx *ctx, long n, const double *x, const double *y, double *dot_host, void *stream);   /* VecDot / VecNorm^2 */
/* generic assembled AIJ: y = A x, or y = addto + alpha*(A x) when addto != NULL (rows: ascending columns).
 * `col` holds element offsets into x (translated by the caller when x is a padded field); when y/addto are padded
 * 2-D grid fields pass row_n > 0: row r lands at row_org + (r / row_n)*row_pitch + r % row_n. */
int  mgk_csr_mult_f64(mgk_ctx *ctx, long nrows, const long *rowptr, const int *col, const double *val,
                      const double *x, double *y, double alpha, const double *addto,
                      int row_n, long row_pitch, long row_org, void *stream);

/* ---- fp32 fields and the fp64<->fp32 bridges of the mixed-precision cycle (BASELINE.json config 5:
 * "fp32 smoother sweeps with fp64 residual/correction"; no reference counterpart).
 * Same kernels with T = float (one lane owns 4 unknowns); same canonical arithmetic evaluated in fp32;
 * coefficients are passed as doubles and rounded to float once.  3-D only. */
int  mgk_geom_init_f32(mgk_geom *g, int dim, int nx, int ny, int nz);      /* geometry of a float field (element units) */
int  mgk_jacobi_f32(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                    const float *b, const float *u, float *unew, void *stream);
int  mgk_jacobi_range_f32(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                          const float *b, const float *u, float *unew, int zbeg, int zend, void *stream);
int  mgk_jacobi_zero_f32(mgk_ctx *ctx, const mgk_geom *g, double dinv, double scale, const float *b, float *unew, void *stream);
int  mgk_residual_f32(mgk_ctx *ctx, const mgk_geom *g, const double *coef, const float *b, const float *u, float *r, void *stream);
int  mgk_restrict_fw_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const float *rf, float *bc, void *stream);
int  mgk_prolong_add_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const float *uc, float *uf, void *stream);
int  mgk_residual_restrict_f32(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                               const float *b, const float *u, float *bc, void *stream);
/* fp64 residual b - A u stored as fp32 + its fp64 sum of squares, one pass (reads 16 B, writes 4 B per unknown) */
int  mgk_residual_f64_to_f32(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *g32, const double *coef,
                             const double *b, const double *u, float *r32, double *sumsq_host, void *stream);
/* u64 += (double) e32 */
/* both of the above in one pass: unew = u + (double) e32, r32 = (float)(b - A unew), *sumsq_host = sum r^2 (32 B/unknown) */
int  mgk_correct_residual_f64_f32(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *g32, const double *coef,
                                  const double *b, const double *u, const float *e32, double *unew, float *r32,
                                  double *sumsq_host, void *stream);
int  mgk_correct_f64_from_f32(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *g32, const float *e32, double *u, void *stream);
/* the two bridges above, also writing e0 = scale * (r32 * dinv) in fp32: the first sweep of the fp32 correction cycle from its zero
 * guess (the arithmetic of mgk_jacobi_zero_f32 on the value just stored): one launch and one read of r32 less per outer step */
int  mgk_residual_f64_to_f32_jz(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *g32, const double *coef,
                                const double *b, const double *u, float *r32, float *e0, double dinv, double scale,
                                double *sumsq_host, void *stream);
int  mgk_correct_residual_f64_f32_jz(mgk_ctx *ctx, const mgk_geom *g, const mgk_geom *g32, const double *coef,
                                     const double *b, const double *u, const float *e32, double *unew, float *r32,
                                     float *e0, double dinv, double scale, double *sumsq_host, void *stream);
int  mgk_pack_f32(mgk_ctx *ctx, const mgk_geom *g32, const double *compact_dev, float *padded_dev, void *stream);
int  mgk_unpack_f32(mgk_ctx *ctx, const mgk_geom *g32, const float *padded_dev, double *compact_dev, void *stream);

/* ---- round 3: THREE sweeps per pass, 2-D (independent-wave kernels, any vertex-centred 2-D grid) ----
 * KSPSolve with max_it = 3 (src/solver.c:1531, :1536, :1542) in ONE pass over the level instead of a sweep + a two-sweep pass:
 *   mgk_jacobi3_2d_f64          unew = J(J(J(u)))                                                                    24 B / unknown
 *   mgk_jacobi3_2d_sumsq_f64    ... and *sumsq_host = || b - A u ||^2 of the INPUT field: the norm that closes cycle k (:1545-1546)
 *                               and all three pre-smoothing sweeps of cycle k+1 (:1531); unew is adopted only if a cycle follows
 *   mgk_jacobi3_2d_zero_f64     from the zero guess (:1536): the first sweep is scale * (b * dinv), u is not read         16 B
 *   mgk_prolong_jacobi3_2d_f64  unew = J(J(J(u + P uc))): MatMult(pro) + VecAXPY + the three post-smoothing sweeps (:1540-1542)  25 B
 * coef / dinv: the level's five constants {(i-1), W, C, E, (i+1)} and 1 / diag; or ctab / dtab != NULL: per-row tables (stretched
 * meshes, as the *_rowcoef_* entry points; coef may then be NULL).  Every stage is the arithmetic of mgk_jacobi_f64: the results
 * equal three separate sweeps bit for bit. */
int  mgk_jacobi3_2d_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                        const double *b, const double *u, double *unew, void *stream);
int  mgk_jacobi3_2d_sumsq_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                              const double *b, const double *u, double *unew, double *sumsq_host, void *stream);
/* ... and r = b - A u of the input field is also stored (8 B more): KSPBuildResidual + VecNorm + the next KSPSolve's three sweeps of the
 * PETSc-surface drop-in (the three-sweep form of mgk_jacobi_sumsq_store_f64) */
int  mgk_jacobi3_2d_sumsq_store_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab,
                                    const double *dtab, const double *b, const double *u, double *unew, double *r, double *sumsq_host, void *stream);
int  mgk_jacobi3_2d_zero_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                             const double *b, double *unew, void *stream);
int  mgk_prolong_jacobi3_2d_f64(mgk_ctx *ctx, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                const double *ctab, const double *dtab, const double *b, const double *uc, const double *u, double *unew,
                                void *stream);

/* the same in 3-D (whole grids, any vertex-centred shape): unew = J(J(J(u))) [and *sumsq_host = || b - A u ||^2 of the input field] in one
 * pass, 24 B per unknown.  Independent waves, one per SIMD (up to 512 registers per lane), 120 x 4 point columns per wave marching along z. */
int  mgk_jacobi3_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                     const double *b, const double *u, double *unew, void *stream);
int  mgk_jacobi3_sumsq_f64(mgk_ctx *ctx, const mgk_geom *g, const double *coef, double dinv, double scale,
                           const double *b, const double *u, double *unew, double *sumsq_host, void *stream);

/* ---- round 3: primitives of the peer halo transport (mg_comm_peer_*, include/mg_comm.h) ----
 * mgk_ipc_alloc: `bytes` of FINE-GRAINED device memory (zeroed; not cached in L2, so that a neighbour's writes over xGMI are what the next
 * load sees) and its 64-byte inter-process handle; mgk_ipc_open maps another process's allocation (any device of the node) into this
 * one; mgk_peer_copy: copy between two device allocations that may live on different GPUs, queued on `stream` (between devices the
 * runtime uses the copy engines: no workgroup).  mgk_flag_set / mgk_flag_wait: ONE-wave kernels that store `value` into a (possibly
 * remote) 8-byte flag word with system-scope release / wait until the local flag word is >= value (acquire; after timeout_s they set
 * *status_u32 (fine-grained or pinned memory) to 1 and return, so that a stream never hangs).  mgk_peer_allreduce: in-place sum of n <= 64
 * device doubles over `nranks` <= MGK_PEER_MAX processes through per-rank slot blocks (2 x nranks x 65 x 8 bytes) in fine-grained memory,
 * peer_blocks[r] = rank r's block as mapped here; summed in rank order: the same bits on every rank; one wave. */
#define MGK_IPC_HANDLE_BYTES 64
#define MGK_PEER_MAX 16
int  mgk_ipc_alloc(mgk_ctx *ctx, size_t bytes, void **ptr, void *handle64);
int  mgk_ipc_open(mgk_ctx *ctx, const void *handle64, void **ptr);
int  mgk_ipc_close(mgk_ctx *ctx, void *ptr);
int  mgk_peer_copy(mgk_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream);
int  mgk_flags_set(mgk_ctx *ctx, void *const *flags, int n, unsigned long long value, void *stream);                 /* n <= MGK_PEER_MAX words, one launch */
int  mgk_flags_wait(mgk_ctx *ctx, void *const *flags, int n, unsigned long long value, double timeout_s, void *status_u32, void *stream);
int  mgk_flag_set(mgk_ctx *ctx, void *flag, unsigned long long value, void *stream);
int  mgk_flag_wait(mgk_ctx *ctx, const void *flag, unsigned long long value, double timeout_s, void *status_u32, void *stream);
int  mgk_peer_allreduce(mgk_ctx *ctx, void *const *peer_blocks, int nranks, int me, unsigned long long seq, double *vals_dev, int n,
                        double timeout_s, void *status_u32, void *stream);

/* tuning knob for the marching stencil kernel (profiling only): <=0 keeps the built-in choice */
void mgk_set_tuning(int variant, int zchunk);

#ifdef __cplusplus
}
#endif
#endif
