/*
 * mgo.c -- CPU ORACLE (test infrastructure only, see mgo.h).
 *
 * Restates, in plain C, what the reference computes on the `-cycle 0` path.
 * Every function names the reference lines it follows.  Build with
 *   gcc -O2 -ffp-contract=off -fopenmp   (see oracle/Makefile)
 * -ffp-contract=off is REQUIRED: the canonical arithmetic of this path has no
 * fused multiply-adds (mgo.h header), and the HIP kernels are compiled the
 * same way so that field values agree bit for bit.
 */
#include "mgo.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* loops shorter than this run serially: the GPU box exposes far more hardware threads than the
 * CPU share a job gets, and a parallel region per tiny coarse-level loop would dominate */
#define MGO_PAR_MIN 32768

int mgo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* integer half                                                        */
/* ------------------------------------------------------------------ */

/* src/matbuild.c:10-25 (square-and-multiply integer power) */
int mgo_ipow(int base, int exp) {
    int r = 1;
    for (; exp; exp >>= 1, base *= base)
        if (exp & 1) r *= base;
    return r;
}

/* src/matbuild.c:120-144: first (totaln % procs) ranks get one extra index */
void mgo_get_ranges(int totaln, int procs, int *ranges) {
    int q = totaln / procs, rem = totaln % procs;
    ranges[0] = 0;
    for (int p = 0; p < procs; p++) ranges[p + 1] = ranges[p] + q + (p < rem ? 1 : 0);
}

/* src/matbuild.c:27-47: one grid per level, the surplus grids go to the last level;
 * ids are handed out consecutively level after level */
int mgo_grid_ids(int totalGrids, int levels, int *ngrids_out, int *gridId_out) {
    int id = 0;
    for (int l = 0; l < levels; l++) {
        ngrids_out[l] = 1;
        if (l == levels - 1) ngrids_out[l] += totalGrids - levels;
        for (int g = 0; g < ngrids_out[l]; g++) gridId_out[id] = id, id++;
    }
    return id;
}

/* src/matbuild.c:62-66 */
int mgo_grid_n(int npts, int g) { return (npts - 1) / mgo_ipow(2, g) - 1; }

int mgo_level_total_2d(int npts, int totalGrids, int levels, int l) {
    int ng[64], ids[256];
    mgo_grid_ids(totalGrids, levels, ng, ids);
    int first = 0;
    for (int q = 0; q < l; q++) first += ng[q];
    int tot = 0;
    for (int g = 0; g < ng[l]; g++) {
        int n = mgo_grid_n(npts, ids[first + g]);
        tot += n * n;
    }
    return tot;
}

/* The three numbering styles of src/matbuild.c:146-323 for one level.
 * Layout conventions as in the reference: global_out[3*idx..] = (i, j, gridId);
 * grid maps are ni x nj row-major, i is the row (y), j the column (x). */
int mgo_mapping_2d(int npts, int totalGrids, int levels, int style, int procs, int l,
                   int *global_out, int *grid_out, int *ranges_out) {
    int ng[64], ids[256];
    mgo_grid_ids(totalGrids, levels, ng, ids);
    int first = 0;
    for (int q = 0; q < l; q++) first += ng[q];
    const int grids = ng[l];
    const int *gid = ids + first;
    int nn[64], off[65];
    off[0] = 0;
    for (int g = 0; g < grids; g++) {
        nn[g] = mgo_grid_n(npts, gid[g]);
        off[g + 1] = off[g] + nn[g] * nn[g];
    }
    const int nf = nn[0];

    if (style == 0) {
        /* matbuild.c:280-309 "grid after grid": all points of a grid are contiguous */
        int count = 0;
        for (int g = 0; g < grids; g++)
            for (int i = 0; i < nn[g]; i++)
                for (int j = 0; j < nn[g]; j++) {
                    grid_out[off[g] + i * nn[g] + j] = count;
                    global_out[3 * count] = i; global_out[3 * count + 1] = j; global_out[3 * count + 2] = gid[g];
                    count++;
                }
        mgo_get_ranges(count, procs, ranges_out);
        return 0;
    }
    if (style == 1) {
        /* matbuild.c:227-278 "through the grids": walk the fine grid; every coarser grid that has a
         * point at the same physical location is numbered right after it; rank boundaries follow
         * the fine-grid split */
        mgo_get_ranges(nf * nf, procs, ranges_out);
        int count = 0, fcount = 0, rank = 0;
        for (int i = 0; i < nf; i++)
            for (int j = 0; j < nf; j++) {
                for (int g = 0; g < grids; g++) {
                    int gf = mgo_ipow(2, gid[g] - gid[0]);
                    if ((i + 1) % gf != 0 || (j + 1) % gf != 0) continue;
                    int ig = (i + 1) / gf - 1, jg = (j + 1) / gf - 1;
                    grid_out[off[g] + ig * nn[g] + jg] = count;
                    global_out[3 * count] = ig; global_out[3 * count + 1] = jg; global_out[3 * count + 2] = gid[g];
                    count++;
                }
                fcount++;
                if (fcount == ranges_out[rank + 1]) { ranges_out[rank + 1] = count; rank++; }
            }
        return 0;
    }
    if (style == 2) {
        /* matbuild.c:146-225 "local grid after grid": the fine grid is split over ranks; inside a
         * rank its points are ordered grid after grid.  NB the reference's membership test breaks
         * out of the grid loop at the first grid that has no point there (`continue` placed before
         * the gfactor update, matbuild.c:178-180), reproduced here. */
        mgo_get_ranges(nf * nf, procs, ranges_out);
        int *gr = (int *)calloc((size_t)procs * grids + 1, sizeof(int));
        int count = 0, rank = 0;
        for (int i = 0; i < nf; i++)
            for (int j = 0; j < nf; j++) {
                int gf = 1;
                for (int g = 0; g < grids; g++) {
                    if ((i + 1) % gf != 0 || (j + 1) % gf != 0) continue;
                    gf *= 2;
                    gr[rank * grids + g] += 1;
                }
                count++;
                if (count == ranges_out[rank + 1]) {
                    int tot = 0;
                    for (int g = 0; g < grids; g++) tot += gr[rank * grids + g];
                    ranges_out[rank + 1] = ranges_out[rank] + tot;
                    rank++;
                }
            }
        gr[procs * grids] = ranges_out[procs];
        for (int q = procs * grids - 1; q >= 0; q--) gr[q] = gr[q + 1] - gr[q];
        for (int g = 0; g < grids; g++) {
            int rk = 0, c = gr[g];
            for (int i = 0; i < nn[g]; i++)
                for (int j = 0; j < nn[g]; j++) {
                    while (c == gr[rk * grids + g + 1]) { rk++; c = gr[rk * grids + g]; }
                    grid_out[off[g] + i * nn[g] + j] = c;
                    global_out[3 * c] = i; global_out[3 * c + 1] = j; global_out[3 * c + 2] = gid[g];
                    c++;
                }
        }
        free(gr);
        return 0;
    }
    return -1;
}

/* src/matbuild.c:422-431 */
void mgo_restriction_stencil(double w[9]) {
    for (int i = 0; i < 3; i++) {
        w[i * 3 + 0] = 0.125 - 0.0625 * fabs((double)(1 - i));
        w[i * 3 + 1] = 0.25 - 0.125 * fabs((double)(1 - i));
        w[i * 3 + 2] = 0.125 - 0.0625 * fabs((double)(1 - i));
    }
}
/* src/matbuild.c:398-407 */
void mgo_prolongation_stencil(double w[9]) {
    for (int i = 0; i < 3; i++) {
        w[i * 3 + 0] = 0.5 - 0.25 * fabs((double)(1 - i));
        w[i * 3 + 1] = 1.0 - 0.5 * fabs((double)(1 - i));
        w[i * 3 + 2] = 0.5 - 0.25 * fabs((double)(1 - i));
    }
}
/* 3-D extension: third tensor factor of the same 1-D weights */
static double w1d_res(int d) { return 0.5 - 0.25 * fabs((double)(1 - d)); }   /* 1/4 1/2 1/4 */
static double w1d_pro(int d) { return 1.0 - 0.5 * fabs((double)(1 - d)); }    /* 1/2 1 1/2 */

/* ------------------------------------------------------------------ */
/* mesh / problem                                                      */
/* ------------------------------------------------------------------ */

/* src/mesh.c:140-171, uniform branches.  Axis 0 accumulates d = (hi-lo)/(n-1), axis>=1
 * accumulates tmp_d = length/(double)(n-1): the same double for bounds [0,1].  Interior points
 * are built by repeated addition, the last point is set to the bound itself. */
void mgo_coords_uniform(int npts, int axis, double *c) {
    (void)axis;
    c[0] = 0.0;
    c[npts - 1] = 1.0;
    double d = (c[npts - 1] - c[0]) / (npts - 1);
    for (int j = 1; j < npts - 1; j++) c[j] = c[j - 1] + d;
}

/* src/mesh.c:145-193: h = sqrt(sum d_i^2); d_0 is the spacing, d_{i>=1} the max |diff| */
double mgo_mesh_h(int dim, int npts) {
    double *c = (double *)malloc(sizeof(double) * npts);
    mgo_coords_uniform(npts, 0, c);
    double d0 = (c[npts - 1] - c[0]) / (npts - 1);
    double d1 = 0.0;
    for (int j = 1; j < npts - 1; j++) d1 = fmax(d1, fabs(c[j] - c[j - 1]));
    d1 = fmax(d1, fabs(c[npts - 2] - c[npts - 1]));
    double h = d0 * d0;
    for (int a = 1; a < dim; a++) h += d1 * d1;
    free(c);
    return sqrt(h);
}

/* src/problem.c:3-22 */
void mgo_opA(const double *m, const double *h, double *A) {
    double hx2 = h[0] * h[0], hy2 = h[1] * h[1];
    A[0] = (m[1] / hy2) - (m[3] / (2 * h[1]));
    A[1] = (m[0] / hx2) - (m[2] / (2 * h[0]));
    A[2] = -2.0 * ((m[0] / hx2) + (m[1] / hy2));
    A[3] = (m[0] / hx2) + (m[2] / (2 * h[0]));
    A[4] = (m[1] / hy2) + (m[3] / (2 * h[1]));
}

/* h of level l: src/matbuild.c:99-104 (1/(n+1)); coefficients: OpA with MetricsUniform
 * (src/mesh.c:29-43).  3-D extension: z terms added the same way, diagonal summed x,y,z. */
void mgo_level_stencil(int dim, int npts, int l, double *As, double *h_out) {
    int n = mgo_grid_n(npts, l);
    double h[3] = {1.0 / (n + 1), 1.0 / (n + 1), 1.0 / (n + 1)};
    double metrics[5] = {1.0, 1.0, 0.0, 0.0, 0.0};
    if (h_out) *h_out = h[0];
    if (dim == 2) { mgo_opA(metrics, h, As); return; }
    double hx2 = h[0] * h[0], hy2 = h[1] * h[1], hz2 = h[2] * h[2];
    double mz = 1.0, mzz = 0.0;
    As[0] = (mz / hz2) - (mzz / (2 * h[2]));
    As[1] = (metrics[1] / hy2) - (metrics[3] / (2 * h[1]));
    As[2] = (metrics[0] / hx2) - (metrics[2] / (2 * h[0]));
    As[3] = -2.0 * (((metrics[0] / hx2) + (metrics[1] / hy2)) + (mz / hz2));
    As[4] = (metrics[0] / hx2) + (metrics[2] / (2 * h[0]));
    As[5] = (metrics[1] / hy2) + (metrics[3] / (2 * h[1]));
    As[6] = (mz / hz2) + (mzz / (2 * h[2]));
}

/* src/problem.c:24-28; 3-D extension: -3 pi^2 sin sin sin */
double mgo_ffunc(int dim, double x, double y, double z) {
    if (dim == 2) return -2 * MGO_PI * MGO_PI * sin(MGO_PI * x) * sin(MGO_PI * y);
    return -3 * MGO_PI * MGO_PI * sin(MGO_PI * x) * sin(MGO_PI * y) * sin(MGO_PI * z);
}
/* src/problem.c:30-34 */
double mgo_solfunc(int dim, double x, double y, double z) {
    if (dim == 2) return sin(MGO_PI * x) * sin(MGO_PI * y);
    return sin(MGO_PI * x) * sin(MGO_PI * y) * sin(MGO_PI * z);
}

/* src/solver.c:586-594: b0[row] = Ffunc(coord[0][j+1], coord[1][i+1]) for row = i*n+j */
void mgo_rhs(int dim, int npts, double *b) {
    int n = npts - 2;
    double *c = (double *)malloc(sizeof(double) * npts);
    mgo_coords_uniform(npts, 0, c);
    if (dim == 2) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) b[(long)i * n + j] = mgo_ffunc(2, c[j + 1], c[i + 1], 0.0);
    } else {
#pragma omp parallel for if (n > 32)
        for (int k = 0; k < n; k++)
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++)
                    b[((long)k * n + i) * n + j] = mgo_ffunc(3, c[j + 1], c[i + 1], c[k + 1]);
    }
    free(c);
}

/* src/solver.c:1211-1237: max |e|, sum |e|, sqrt(sum e^2), accumulated in row-major order */
void mgo_error_norms(int dim, int npts, const double *u, double err[3]) {
    int n = npts - 2;
    double *c = (double *)malloc(sizeof(double) * npts);
    mgo_coords_uniform(npts, 0, c);
    err[0] = err[1] = err[2] = 0.0;
    int nk = dim == 3 ? n : 1;
    for (int k = 0; k < nk; k++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                double sol = mgo_solfunc(dim, c[j + 1], c[i + 1], dim == 3 ? c[k + 1] : 0.0);
                double diff = fabs(u[((long)k * n + i) * n + j] - sol);
                err[0] = fmax(diff, err[0]);
                err[1] = err[1] + diff;
                err[2] = err[2] + diff * diff;
            }
    err[2] = sqrt(err[2]);
    free(c);
}

/* ---- stretched meshes (2-D only, -mesh 1 / 2): src/mesh.c:45-107 (metrics), :154-176 (coords) ---- */
/* y coordinates: NONUNIFORM1 y_j = hi - L*cos(pi/2 * j/(n-1)), NONUNIFORM2 y_j = lo + L*(exp(2 eta)-1)/(exp(2)-1);
 * x stays uniform (src/mesh.c:140-152). */
void mgo_coords_mesh(int npts, int axis, int mesh, double *c) {
    if (axis == 0 || mesh == 0) { mgo_coords_uniform(npts, axis, c); return; }
    c[0] = 0.0; c[npts - 1] = 1.0;
    double length = c[npts - 1] - c[0];
    for (int j = 1; j < npts - 1; j++) {
        if (mesh == 1) c[j] = 1.0 - length * (cos(MGO_PI * 0.5 * (j / (double)(npts - 1))));
        else { double eta = (j / (double)(npts - 1)); c[j] = 0.0 + length * ((exp(2 * eta) - 1) / (exp(2) - 1)); }
    }
}
/* metrics at (x,y) for bounds [0,1]^2: src/mesh.c:29-43 (uniform), :45-75 (NONUNIFORM1), :77-107 (NONUNIFORM2) */
void mgo_metrics(int mesh, double x, double y, double *m) {
    (void)x;
    const double b0 = 0.0, b1 = 1.0, b2 = 0.0, b3 = 1.0;
    if (mesh == 0) { m[0] = 1.0; m[1] = 1.0; m[2] = 0.0; m[3] = 0.0; m[4] = 0.0; return; }
    if (mesh == 1) {
        double temp = ((b3 - b2) * (b3 - b2) - (b3 - y) * (b3 - y));
        m[0] = 1.0;
        m[1] = 4.0 / (MGO_PI * MGO_PI * temp);
        m[2] = 0.0;
        m[3] = (-2.0 * (b3 - y)) / (MGO_PI * sqrt(temp * temp * temp));
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
/* b0 and the error norms on a stretched mesh (src/solver.c:586-594, :1211-1237 with mesh->coord) */
void mgo_rhs_mesh(int npts, int mesh, double *b) {
    int n = npts - 2;
    double *cx = (double *)malloc(sizeof(double) * npts), *cy = (double *)malloc(sizeof(double) * npts);
    mgo_coords_mesh(npts, 0, mesh, cx); mgo_coords_mesh(npts, 1, mesh, cy);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) b[(long)i * n + j] = mgo_ffunc(2, cx[j + 1], cy[i + 1], 0.0);
    free(cx); free(cy);
}
void mgo_error_norms_mesh(int npts, int mesh, const double *u, double err[3]) {
    int n = npts - 2;
    double *cx = (double *)malloc(sizeof(double) * npts), *cy = (double *)malloc(sizeof(double) * npts);
    mgo_coords_mesh(npts, 0, mesh, cx); mgo_coords_mesh(npts, 1, mesh, cy);
    err[0] = err[1] = err[2] = 0.0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double sol = mgo_solfunc(2, cx[j + 1], cy[i + 1], 0.0);
            double diff = fabs(u[(long)i * n + j] - sol);
            err[0] = fmax(diff, err[0]); err[1] = err[1] + diff; err[2] = err[2] + diff * diff;
        }
    err[2] = sqrt(err[2]);
    free(cx); free(cy);
}

/* ------------------------------------------------------------------ */
/* assembled (AIJ) path                                                */
/* ------------------------------------------------------------------ */

typedef struct { int row, col; double val; } coo_t;
typedef struct { coo_t *e; long n, cap; } coo_list;

static void coo_push(coo_list *L, long row, long col, double v) {   /* MatSetValue(..., ADD_VALUES) */
    if (L->n == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 1024;
        L->e = (coo_t *)realloc(L->e, sizeof(coo_t) * L->cap);
    }
    L->e[L->n].row = (int)row; L->e[L->n].col = (int)col; L->e[L->n].val = v; L->n++;
}

/* MatAssemblyEnd: rows sorted by column, duplicates accumulated in insertion order */
static mgo_csr *coo_to_csr(coo_list *L, long nrows, long ncols) {
    mgo_csr *m = (mgo_csr *)calloc(1, sizeof(mgo_csr));
    m->nrows = nrows; m->ncols = ncols;
    long *cnt = (long *)calloc(nrows + 1, sizeof(long));
    for (long q = 0; q < L->n; q++) cnt[L->e[q].row + 1]++;
    for (long r = 0; r < nrows; r++) cnt[r + 1] += cnt[r];
    coo_t *s = (coo_t *)malloc(sizeof(coo_t) * (L->n ? L->n : 1));
    long *pos = (long *)malloc(sizeof(long) * (nrows + 1));
    memcpy(pos, cnt, sizeof(long) * (nrows + 1));
    for (long q = 0; q < L->n; q++) s[pos[L->e[q].row]++] = L->e[q];   /* stable bucket by row */
    m->rowptr = (long *)malloc(sizeof(long) * (nrows + 1));
    m->col = (int *)malloc(sizeof(int) * (L->n ? L->n : 1));
    m->val = (double *)malloc(sizeof(double) * (L->n ? L->n : 1));
    long out = 0;
    for (long r = 0; r < nrows; r++) {
        m->rowptr[r] = out;
        long a = cnt[r], b = cnt[r + 1];
        for (long q = a + 1; q < b; q++) {           /* stable insertion sort by column */
            coo_t t = s[q]; long p = q - 1;
            while (p >= a && s[p].col > t.col) { s[p + 1] = s[p]; p--; }
            s[p + 1] = t;
        }
        for (long q = a; q < b; q++) {
            if (out > m->rowptr[r] && m->col[out - 1] == s[q].col) m->val[out - 1] += s[q].val;
            else { m->col[out] = s[q].col; m->val[out] = s[q].val; out++; }
        }
    }
    m->rowptr[nrows] = out; m->nnz = out;
    free(cnt); free(s); free(pos); free(L->e); L->e = NULL; L->n = L->cap = 0;
    return m;
}

void mgo_csr_free(mgo_csr *m) { if (!m) return; free(m->rowptr); free(m->col); free(m->val); free(m); }
long mgo_csr_nrows(const mgo_csr *m) { return m->nrows; }
long mgo_csr_ncols(const mgo_csr *m) { return m->ncols; }
long mgo_csr_nnz(const mgo_csr *m) { return m->nnz; }
void mgo_csr_row(const mgo_csr *m, long row, int *ncols, int *cols, double *vals) {
    long a = m->rowptr[row], b = m->rowptr[row + 1];
    *ncols = (int)(b - a);
    for (long q = a; q < b; q++) { cols[q - a] = m->col[q]; vals[q - a] = m->val[q]; }
}

/* src/solver.c:185-253 (fillJacobians) + 489-510 (levelMatrixA), one grid per level.
 * The grid->global map of a one-grid level is the lexicographic identity (all three styles,
 * checked in tests/test_oracle.py), so b[(i)*bj+j] is written as the formula. */
/* variable-coefficient rows for -mesh 1/2 (2-D): metrics at the fine-grid point of (i0,j0), OpA with the level's h */
mgo_csr *mgo_build_A_mesh(int npts, int l, int mesh) {
    int n = mgo_grid_n(npts, l);
    double h[2] = {1.0 / (n + 1), 1.0 / (n + 1)};
    double *cx = (double *)malloc(sizeof(double) * npts), *cy = (double *)malloc(sizeof(double) * npts);
    mgo_coords_mesh(npts, 0, mesh, cx); mgo_coords_mesh(npts, 1, mesh, cy);
    coo_list L = {0};
    long N = (long)n * n;
    int f = mgo_ipow(2, l);
    for (long row = 0; row < N; row++) {
        int i0 = (int)(row / n), j0 = (int)(row % n);
        int ifine = f * (i0 + 1) - 1, jfine = f * (j0 + 1) - 1;       /* solver.c:227-228 */
        double metrics[5], As[5];
        mgo_metrics(mesh, cx[jfine + 1], cy[ifine + 1], metrics);     /* :231 */
        mgo_opA(metrics, h, As);                                      /* :232 */
        if (i0 - 1 >= 0) coo_push(&L, row, (long)(i0 - 1) * n + j0, As[0]);
        if (j0 - 1 >= 0) coo_push(&L, row, (long)i0 * n + j0 - 1, As[1]);
        coo_push(&L, row, row, As[2]);
        if (j0 + 1 < n) coo_push(&L, row, (long)i0 * n + j0 + 1, As[3]);
        if (i0 + 1 < n) coo_push(&L, row, (long)(i0 + 1) * n + j0, As[4]);
    }
    free(cx); free(cy);
    return coo_to_csr(&L, N, N);
}

mgo_csr *mgo_build_A(int dim, int npts, int l) {
    int n = mgo_grid_n(npts, l);
    double As[7];
    mgo_level_stencil(dim, npts, l, As, NULL);
    coo_list L = {0};
    if (dim == 2) {
        long N = (long)n * n;
        for (long row = 0; row < N; row++) {
            int i0 = (int)(row / n), j0 = (int)(row % n);
            if (i0 - 1 >= 0) coo_push(&L, row, (long)(i0 - 1) * n + j0, As[0]);
            if (j0 - 1 >= 0) coo_push(&L, row, (long)i0 * n + j0 - 1, As[1]);
            coo_push(&L, row, row, As[2]);
            if (j0 + 1 < n) coo_push(&L, row, (long)i0 * n + j0 + 1, As[3]);
            if (i0 + 1 < n) coo_push(&L, row, (long)(i0 + 1) * n + j0, As[4]);
        }
        return coo_to_csr(&L, N, N);
    }
    long N = (long)n * n * n, nn = (long)n * n;   /* 3-D extension */
    for (long row = 0; row < N; row++) {
        int k0 = (int)(row / nn), i0 = (int)((row / n) % n), j0 = (int)(row % n);
        if (k0 - 1 >= 0) coo_push(&L, row, row - nn, As[0]);
        if (i0 - 1 >= 0) coo_push(&L, row, row - n, As[1]);
        if (j0 - 1 >= 0) coo_push(&L, row, row - 1, As[2]);
        coo_push(&L, row, row, As[3]);
        if (j0 + 1 < n) coo_push(&L, row, row + 1, As[4]);
        if (i0 + 1 < n) coo_push(&L, row, row + n, As[5]);
        if (k0 + 1 < n) coo_push(&L, row, row + nn, As[6]);
    }
    return coo_to_csr(&L, N, N);
}

/* src/solver.c:1071-1092: row = coarse point (i1,j1); fine window starts at
 * i0 = 2*(i1+1)-1-3/2 = 2*i1; weight w[(i-i0)*3+(j-j0)], zero weights skipped */
mgo_csr *mgo_build_R(int dim, int npts, int l) {
    int nf = mgo_grid_n(npts, l), nc = mgo_grid_n(npts, l + 1);
    double w[9];
    mgo_restriction_stencil(w);
    coo_list L = {0};
    if (dim == 2) {
        for (long row = 0; row < (long)nc * nc; row++) {
            int i1 = (int)(row / nc), j1 = (int)(row % nc);
            int i0 = mgo_ipow(2, 1) * (i1 + 1) - 1 - 3 / 2, j0 = mgo_ipow(2, 1) * (j1 + 1) - 1 - 3 / 2;
            for (int i = i0; i < i0 + 3; i++)
                for (int j = j0; j < j0 + 3; j++) {
                    double wt = w[(i - i0) * 3 + (j - j0)];
                    if (wt != 0.0) coo_push(&L, row, (long)i * nf + j, wt);
                }
        }
        return coo_to_csr(&L, (long)nc * nc, (long)nf * nf);
    }
    long ncc = (long)nc * nc;
    for (long row = 0; row < ncc * nc; row++) {   /* 3-D extension: weight = wz * w2d */
        int k1 = (int)(row / ncc), i1 = (int)((row / nc) % nc), j1 = (int)(row % nc);
        int k0 = 2 * k1, i0 = 2 * i1, j0 = 2 * j1;
        for (int k = k0; k < k0 + 3; k++)
            for (int i = i0; i < i0 + 3; i++)
                for (int j = j0; j < j0 + 3; j++) {
                    double wt = w1d_res(k - k0) * w[(i - i0) * 3 + (j - j0)];
                    if (wt != 0.0) coo_push(&L, row, ((long)k * nf + i) * nf + j, wt);
                }
    }
    return coo_to_csr(&L, ncc * nc, (long)nf * nf * nf);
}

/* src/solver.c:1131-1152: column = coarse point; entries (fine row, col, w) inserted column by column */
mgo_csr *mgo_build_P(int dim, int npts, int l) {
    int nf = mgo_grid_n(npts, l), nc = mgo_grid_n(npts, l + 1);
    double w[9];
    mgo_prolongation_stencil(w);
    coo_list L = {0};
    if (dim == 2) {
        for (long col = 0; col < (long)nc * nc; col++) {
            int i1 = (int)(col / nc), j1 = (int)(col % nc);
            int i0 = 2 * (i1 + 1) - 1 - 3 / 2, j0 = 2 * (j1 + 1) - 1 - 3 / 2;
            for (int i = i0; i < i0 + 3; i++)
                for (int j = j0; j < j0 + 3; j++) {
                    double wt = w[(i - i0) * 3 + (j - j0)];
                    if (wt != 0.0) coo_push(&L, (long)i * nf + j, col, wt);
                }
        }
        return coo_to_csr(&L, (long)nf * nf, (long)nc * nc);
    }
    long ncc = (long)nc * nc;
    for (long col = 0; col < ncc * nc; col++) {   /* 3-D extension */
        int k1 = (int)(col / ncc), i1 = (int)((col / nc) % nc), j1 = (int)(col % nc);
        int k0 = 2 * k1, i0 = 2 * i1, j0 = 2 * j1;
        for (int k = k0; k < k0 + 3; k++)
            for (int i = i0; i < i0 + 3; i++)
                for (int j = j0; j < j0 + 3; j++) {
                    double wt = w1d_pro(k - k0) * w[(i - i0) * 3 + (j - j0)];
                    if (wt != 0.0) coo_push(&L, ((long)k * nf + i) * nf + j, col, wt);
                }
    }
    return coo_to_csr(&L, (long)nf * nf * nf, ncc * nc);
}

/* MatMult on AIJ (assumed PETSc semantics, mgo.h) */
void mgo_csr_mult(const mgo_csr *m, const double *x, double *y) {
#pragma omp parallel for schedule(static) if (m->nrows > MGO_PAR_MIN)
    for (long r = 0; r < m->nrows; r++) {
        double sum = 0.0;
        for (long q = m->rowptr[r]; q < m->rowptr[r + 1]; q++) sum += m->val[q] * x[m->col[q]];
        y[r] = sum;
    }
}

void mgo_csr_diag_inv(const mgo_csr *m, double *dinv) {   /* PCJACOBI setup: 1/diag */
#pragma omp parallel for schedule(static) if (m->nrows > MGO_PAR_MIN)
    for (long r = 0; r < m->nrows; r++) {
        double d = 0.0;
        for (long q = m->rowptr[r]; q < m->rowptr[r + 1]; q++) if (m->col[q] == r) d = m->val[q];
        dinv[r] = 1.0 / d;
    }
}

/* KSPBuildResidual default: t = A x ; r = b - t  (solver.c:1534,1545) */
void mgo_residual_csr(const mgo_csr *A, const double *b, const double *x, double *r) {
    mgo_csr_mult(A, x, r);
#pragma omp parallel for schedule(static) if (A->nrows > MGO_PAR_MIN)
    for (long q = 0; q < A->nrows; q++) r[q] = b[q] - r[q];
}

/* KSPSolve with KSPRICHARDSON + KSP_NORM_NONE + PCJACOBI (solver.c:1465-1476,1531) */
void mgo_richardson_csr(const mgo_csr *A, const double *dinv, const double *b, double *x,
                        int maxit, double scale, int guess_nonzero, double *work) {
    long n = A->nrows;
    double *r = work, *z = work + n;
    if (!guess_nonzero) {
        for (long q = 0; q < n; q++) x[q] = 0.0;      /* KSPSolve zero-fills */
        memcpy(r, b, sizeof(double) * n);             /* r = b */
    } else {
        mgo_residual_csr(A, b, x, r);                 /* r = b - A x */
    }
    for (int it = 0; it < maxit; it++) {
#pragma omp parallel for schedule(static) if (n > MGO_PAR_MIN)
        for (long q = 0; q < n; q++) {
            z[q] = r[q] * dinv[q];                    /* PCApply (Jacobi) */
            x[q] = x[q] + scale * z[q];               /* VecAXPY */
        }
        if (it + 1 < maxit) mgo_residual_csr(A, b, x, r);
    }
}

/* KSPCHEBYSHEV, classic three-term form (PETSc <= 3.8 cheby.c; version unpinned by the reference):
 *   scale = 2/(emax+emin); alpha = 1 - scale*emin; mu = 1/alpha; omegaprod = 2/alpha
 *   p_k = p_km1 + scale*B r0 ; then
 *   c_kp1 = 2 mu c_k - c_km1; omega = omegaprod c_k / c_kp1
 *   p_kp1 = (1-omega) p_km1 + omega p_k + omega*Gamma*scale * B(b - A p_k) */
void mgo_chebyshev_csr(const mgo_csr *A, const double *dinv, const double *b, double *x,
                       int maxit, double emin, double emax, int guess_nonzero, double *work) {
    long n = A->nrows;
    double *r = work, *p0 = work + n, *p1 = work + 2 * n, *p2 = work + 3 * n;
    double scale = 2.0 / (emax + emin), alpha = 1.0 - scale * emin, Gamma = 1.0;
    double mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
    double ckm1 = 1.0, ck = mu, ckp1;
    if (!guess_nonzero) { for (long q = 0; q < n; q++) x[q] = 0.0; memcpy(r, b, sizeof(double) * n); }
    else mgo_residual_csr(A, b, x, r);
    double *pkm1 = p0, *pk = p1, *pkp1 = p2, *t;
    memcpy(pkm1, x, sizeof(double) * n);
    for (long q = 0; q < n; q++) { double z = r[q] * dinv[q]; pk[q] = pkm1[q] + scale * z; }
    for (int it = 1; it < maxit; it++) {
        ckp1 = 2.0 * mu * ck - ckm1;
        double omega = omegaprod * ck / ckp1;
        double a = 1.0 - omega, g = omega * Gamma * scale;
        mgo_residual_csr(A, b, pk, r);
        for (long q = 0; q < n; q++) {
            double z = r[q] * dinv[q];
            pkp1[q] = a * pkm1[q] + omega * pk[q] + g * z;
        }
        t = pkm1; pkm1 = pk; pk = pkp1; pkp1 = t;
        ckm1 = ck; ck = ckp1;
    }
    memcpy(x, pk, sizeof(double) * n);
}

/* ------------------------------------------------------------------ */
/* matrix-free path                                                    */
/* ------------------------------------------------------------------ */

/* one row of A times x, in ascending-column order, missing neighbours skipped (== the CSR row) */
static inline double st_row(int dim, int n, int nz, const double *As, const double *x,
                            const double *zlo, const double *zhi, int k, int i, int j) {
    long nn = (long)n * n, c = ((long)k * n + i) * n + j;
    double sum = 0.0;
    if (dim == 3) {
        if (k - 1 >= 0) sum += As[0] * x[c - nn]; else if (zlo) sum += As[0] * zlo[(long)i * n + j];
        if (i - 1 >= 0) sum += As[1] * x[c - n];
        if (j - 1 >= 0) sum += As[2] * x[c - 1];
        sum += As[3] * x[c];
        if (j + 1 < n) sum += As[4] * x[c + 1];
        if (i + 1 < n) sum += As[5] * x[c + n];
        if (k + 1 < nz) sum += As[6] * x[c + nn]; else if (zhi) sum += As[6] * zhi[(long)i * n + j];
    } else {
        if (i - 1 >= 0) sum += As[0] * x[c - n];
        if (j - 1 >= 0) sum += As[1] * x[c - 1];
        sum += As[2] * x[c];
        if (j + 1 < n) sum += As[3] * x[c + 1];
        if (i + 1 < n) sum += As[4] * x[c + n];
    }
    return sum;
}

void mgo_st_apply(int dim, int n, int nz, const double *As, const double *x,
                  const double *zlo, const double *zhi, double *y) {
    if (dim == 2) nz = 1;
#pragma omp parallel for collapse(2) schedule(static) if ((long)nz * n * n > MGO_PAR_MIN)
    for (int k = 0; k < nz; k++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                y[((long)k * n + i) * n + j] = st_row(dim, n, nz, As, x, zlo, zhi, k, i, j);
}

/* one Richardson/Jacobi sweep:  t = A u; r = b - t; z = r*dinv; unew = u + scale*z.
 * zero_guess: u is not read (u == 0: t == 0, r == b, unew = 0 + scale*z = scale*z). */
void mgo_st_jacobi(int dim, int n, int nz, const double *As, double scale, const double *b,
                   const double *u, const double *zlo, const double *zhi, double *unew, int zero_guess) {
    if (dim == 2) nz = 1;
    double dinv = 1.0 / As[dim == 3 ? 3 : 2];
#pragma omp parallel for collapse(2) schedule(static) if ((long)nz * n * n > MGO_PAR_MIN)
    for (int k = 0; k < nz; k++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                long c = ((long)k * n + i) * n + j;
                if (zero_guess) { double z = b[c] * dinv; unew[c] = scale * z; continue; }
                double t = st_row(dim, n, nz, As, u, zlo, zhi, k, i, j);
                double r = b[c] - t;
                double z = r * dinv;
                unew[c] = u[c] + scale * z;
            }
}

/* Chebyshev recurrence step: pkp1 = c_km1*pkm1 + c_k*pk + c_z * ((b - A pk)*dinv) */
void mgo_st_cheby_step(int dim, int n, int nz, const double *As, const double *b,
                       const double *pk, const double *zlo, const double *zhi, const double *pkm1,
                       double c_km1, double c_k, double c_z, double *pkp1) {
    if (dim == 2) nz = 1;
    double dinv = 1.0 / As[dim == 3 ? 3 : 2];
#pragma omp parallel for collapse(2) schedule(static) if ((long)nz * n * n > MGO_PAR_MIN)
    for (int k = 0; k < nz; k++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                long c = ((long)k * n + i) * n + j;
                double t = st_row(dim, n, nz, As, pk, zlo, zhi, k, i, j);
                double r = b[c] - t;
                double z = r * dinv;
                pkp1[c] = c_km1 * pkm1[c] + c_k * pk[c] + c_z * z;
            }
}

void mgo_st_residual(int dim, int n, int nz, const double *As, const double *b, const double *u,
                     const double *zlo, const double *zhi, double *r) {
    if (dim == 2) nz = 1;
#pragma omp parallel for collapse(2) schedule(static) if ((long)nz * n * n > MGO_PAR_MIN)
    for (int k = 0; k < nz; k++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                long c = ((long)k * n + i) * n + j;
                r[c] = b[c] - st_row(dim, n, nz, As, u, zlo, zhi, k, i, j);
            }
}

/* full weighting, sum in ascending fine index (== row of res, solver.c:1081-1090) */
void mgo_st_restrict(int dim, int nf, int nzf, int nzc, const double *rf, const double *fzhi, double *bc) {
    int nc = (nf - 1) / 2;
    double w[9];
    mgo_restriction_stencil(w);
    if (dim == 2) {
#pragma omp parallel for schedule(static) if ((long)nc * nc > MGO_PAR_MIN)
        for (int i1 = 0; i1 < nc; i1++)
            for (int j1 = 0; j1 < nc; j1++) {
                double sum = 0.0;
                for (int di = 0; di < 3; di++)
                    for (int dj = 0; dj < 3; dj++)
                        sum += w[di * 3 + dj] * rf[(long)(2 * i1 + di) * nf + 2 * j1 + dj];
                bc[(long)i1 * nc + j1] = sum;
            }
        return;
    }
    long nnf = (long)nf * nf;
#pragma omp parallel for collapse(2) schedule(static) if ((long)nzc * nc * nc > MGO_PAR_MIN)
    for (int k1 = 0; k1 < nzc; k1++)
        for (int i1 = 0; i1 < nc; i1++)
            for (int j1 = 0; j1 < nc; j1++) {
                double sum = 0.0;
                for (int dk = 0; dk < 3; dk++) {
                    int kf = 2 * k1 + dk;
                    const double *pl = kf < nzf ? rf + kf * nnf : fzhi;
                    if (!pl) continue;       /* beyond the grid: no matrix entry */
                    for (int di = 0; di < 3; di++)
                        for (int dj = 0; dj < 3; dj++)
                            sum += (w1d_res(dk) * w[di * 3 + dj]) * pl[(long)(2 * i1 + di) * nf + 2 * j1 + dj];
                }
                bc[((long)k1 * nc + i1) * nc + j1] = sum;
            }
}

/* uf += P uc: row of pro summed in ascending coarse index, then VecAXPY(u,1.0,rv) (solver.c:1540-1541) */
void mgo_st_prolong_add(int dim, int nf, int nzf, int nzc, const double *uc,
                        const double *czlo, const double *czhi, double *uf) {
    int nc = (nf - 1) / 2;
    int nk = dim == 3 ? nzf : 1;
    long ncc = (long)nc * nc;
#pragma omp parallel for collapse(2) schedule(static) if ((long)nk * nf * nf > MGO_PAR_MIN)
    for (int k = 0; k < nk; k++)
        for (int i = 0; i < nf; i++)
            for (int j = 0; j < nf; j++) {
                /* contributing coarse range per axis: f odd -> c=(f-1)/2 (w 1); f even -> f/2-1, f/2 (w 1/2) */
                int kc0 = 0, kc1 = 0;
                if (dim == 3) { kc0 = (k & 1) ? (k - 1) / 2 : k / 2 - 1; kc1 = (k & 1) ? kc0 : k / 2; }
                int ic0 = (i & 1) ? (i - 1) / 2 : i / 2 - 1, ic1 = (i & 1) ? ic0 : i / 2;
                int jc0 = (j & 1) ? (j - 1) / 2 : j / 2 - 1, jc1 = (j & 1) ? jc0 : j / 2;
                double sum = 0.0;
                for (int kc = kc0; kc <= kc1; kc++) {
                    const double *pl;
                    if (dim == 2) pl = uc;
                    else if (kc < 0) pl = czlo;
                    else if (kc >= nzc) pl = czhi;
                    else pl = uc + kc * ncc;
                    if (!pl) continue;
                    double wk = dim == 3 ? ((k & 1) ? 1.0 : 0.5) : 1.0;
                    for (int ic = ic0; ic <= ic1; ic++) {
                        if (ic < 0 || ic >= nc) continue;
                        double wi = (i & 1) ? 1.0 : 0.5;
                        for (int jc = jc0; jc <= jc1; jc++) {
                            if (jc < 0 || jc >= nc) continue;
                            double wj = (j & 1) ? 1.0 : 0.5;
                            double wt = dim == 3 ? wk * (wi * wj) : wi * wj;
                            sum += wt * pl[(long)ic * nc + jc];
                        }
                    }
                }
                long c = ((long)k * nf + i) * nf + j;
                uf[c] = uf[c] + sum;
            }
}

/* sum of squares with long-double block accumulation (accurate reference value) */
double mgo_sumsq(const double *x, long n) {
    const long B = 4096;
    long nb = (n + B - 1) / B;
    long double *part = (long double *)malloc(sizeof(long double) * (nb ? nb : 1));
#pragma omp parallel for schedule(static) if (nb > 8)
    for (long bidx = 0; bidx < nb; bidx++) {
        long a = bidx * B, e = a + B < n ? a + B : n;
        long double s = 0.0L;
        for (long q = a; q < e; q++) s += (long double)x[q] * (long double)x[q];
        part[bidx] = s;
    }
    long double tot = 0.0L;
    for (long bidx = 0; bidx < nb; bidx++) tot += part[bidx];
    free(part);
    return (double)tot;
}
double mgo_norm2(const double *x, long n) { return sqrt(mgo_sumsq(x, n)); }

/* ------------------------------------------------------------------ */
/* the V-cycle, src/solver.c:1414-1575                                 */
/* ------------------------------------------------------------------ */

typedef struct {
    int n; long N;
    double As[7];
    mgo_csr *A, *R, *P;      /* assembled mode */
    double *dinv;
    double *u, *b, *rv, *work;
    int guess_nonzero;       /* KSPSetInitialGuessNonzero state of ksp[l] */
} olevel;

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

/* KSPSolve(ksp[l], b[l], u[l]) */
static void smooth(const mgo_vcycle_cfg *c, olevel *L, int maxit) {
    int dim = c->dim;
    if (c->use_csr) {
        if (c->ksp_type == 0) mgo_richardson_csr(L->A, L->dinv, L->b, L->u, maxit, c->scale, L->guess_nonzero, L->work);
        else mgo_chebyshev_csr(L->A, L->dinv, L->b, L->u, maxit, c->emin, c->emax, L->guess_nonzero, L->work);
        return;
    }
    long N = L->N;
    double *cur = L->u, *nxt = L->work;
    if (c->ksp_type == 0) {
        for (int it = 0; it < maxit; it++) {
            int zg = (it == 0 && !L->guess_nonzero);
            mgo_st_jacobi(dim, L->n, L->n, L->As, c->scale, L->b, cur, NULL, NULL, nxt, zg);
            double *t = cur; cur = nxt; nxt = t;
        }
        if (maxit == 0 && !L->guess_nonzero) memset(L->u, 0, sizeof(double) * N);
    } else {
        double scale = 2.0 / (c->emax + c->emin), alpha = 1.0 - scale * c->emin, Gamma = 1.0;
        double mu = 1.0 / alpha, omegaprod = 2.0 / alpha, ckm1 = 1.0, ck = mu, ckp1;
        double *pkm1 = L->u, *pk = L->work, *pkp1 = L->work + N;
        if (!L->guess_nonzero) memset(pkm1, 0, sizeof(double) * N);
        mgo_st_jacobi(dim, L->n, L->n, L->As, scale, L->b, pkm1, NULL, NULL, pk, !L->guess_nonzero);
        for (int it = 1; it < maxit; it++) {
            ckp1 = 2.0 * mu * ck - ckm1;
            double omega = omegaprod * ck / ckp1;
            mgo_st_cheby_step(dim, L->n, L->n, L->As, L->b, pk, NULL, NULL, pkm1,
                              1.0 - omega, omega, omega * Gamma * scale, pkp1);
            double *t = pkm1; pkm1 = pk; pk = pkp1; pkp1 = t;
            ckm1 = ck; ck = ckp1;
        }
        cur = pk;
    }
    if (cur != L->u) memcpy(L->u, cur, sizeof(double) * N);
}

static void residual(const mgo_vcycle_cfg *c, olevel *L) {     /* KSPBuildResidual -> rv[l] */
    if (c->use_csr) mgo_residual_csr(L->A, L->b, L->u, L->rv);
    else mgo_st_residual(c->dim, L->n, L->n, L->As, L->b, L->u, NULL, NULL, L->rv);
}

int mgo_vcycle(const mgo_vcycle_cfg *c, double *rnorm, double *u_out, double *bnorm_out, double *solve_seconds) {
    int levels = c->levels, dim = c->dim;
    olevel *L = (olevel *)calloc(levels, sizeof(olevel));
    for (int l = 0; l < levels; l++) {
        L[l].n = mgo_grid_n(c->npts, l);
        L[l].N = dim == 3 ? (long)L[l].n * L[l].n * L[l].n : (long)L[l].n * L[l].n;
        mgo_level_stencil(dim, c->npts, l, L[l].As, NULL);
        L[l].u = (double *)calloc(L[l].N, sizeof(double));
        L[l].b = (double *)calloc(L[l].N, sizeof(double));
        L[l].rv = (double *)calloc(L[l].N, sizeof(double));
        L[l].work = (double *)calloc(4 * L[l].N, sizeof(double));
        if (c->use_csr) {
            L[l].A = (c->mesh != 0 && dim == 2) ? mgo_build_A_mesh(c->npts, l, c->mesh) : mgo_build_A(dim, c->npts, l);
            L[l].dinv = (double *)malloc(sizeof(double) * L[l].N);
            mgo_csr_diag_inv(L[l].A, L[l].dinv);
            if (l < levels - 1) { L[l].R = mgo_build_R(dim, c->npts, l); L[l].P = mgo_build_P(dim, c->npts, l); }
        }
    }
    if (c->mesh != 0 && dim == 2 && c->use_csr) mgo_rhs_mesh(c->npts, c->mesh, L[0].b);
    else mgo_rhs(dim, c->npts, L[0].b);                             /* levelvecb */
    double bnorm = mgo_norm2(L[0].b, L[0].N);                       /* solver.c:1512 */
    memset(L[0].u, 0, sizeof(double) * L[0].N);                     /* :1514 */
    if (c->use_csr) mgo_csr_mult(L[0].A, L[0].u, L[0].rv);          /* :1516 */
    else mgo_st_apply(dim, L[0].n, L[0].n, L[0].As, L[0].u, NULL, NULL, L[0].rv);
    for (long q = 0; q < L[0].N; q++) L[0].rv[q] = L[0].rv[q] + (-1.0) * L[0].b[q];   /* :1517 VecAXPY(rv,-1,b) */
    double rchk = mgo_norm2(L[0].rv, L[0].N);
    rnorm[0] = rchk;
    int iter = 0;
    double rtol = c->rtol > 0 ? c->rtol : 1.e-7;
    double t0 = now_s();
    for (;;) {
        if (c->fixed_cycles > 0) { if (iter >= c->fixed_cycles) break; }
        else if (!(iter < c->maxiter && 100000000 * bnorm > rchk && rchk > rtol * bnorm)) break;   /* :1530 */
        smooth(c, &L[0], c->v0);                                               /* :1531 */
        if (iter == 0) L[0].guess_nonzero = 1;                                 /* :1532 */
        for (int l = 1; l < levels; l++) {
            residual(c, &L[l - 1]);                                            /* :1534 */
            if (c->use_csr) mgo_csr_mult(L[l - 1].R, L[l - 1].rv, L[l].b);     /* :1535 */
            else mgo_st_restrict(dim, L[l - 1].n, L[l - 1].n, L[l].n, L[l - 1].rv, NULL, L[l].b);
            smooth(c, &L[l], l == levels - 1 ? c->v1 : c->v0);                 /* :1536 */
            if (l != levels - 1) L[l].guess_nonzero = 1;                       /* :1537 */
        }
        for (int l = levels - 2; l >= 0; l--) {
            if (c->use_csr) {
                mgo_csr_mult(L[l].P, L[l + 1].u, L[l].rv);                     /* :1540 */
                for (long q = 0; q < L[l].N; q++) L[l].u[q] = L[l].u[q] + 1.0 * L[l].rv[q];   /* :1541 */
            } else {
                mgo_st_prolong_add(dim, L[l].n, L[l].n, L[l + 1].n, L[l + 1].u, NULL, NULL, L[l].u);
            }
            smooth(c, &L[l], c->v0);                                           /* :1542 */
            if (l != 0) L[l].guess_nonzero = 0;                                /* :1543 */
        }
        residual(c, &L[0]);                                                    /* :1545 */
        rchk = mgo_norm2(L[0].rv, L[0].N);                                     /* :1546 */
        iter++;
        rnorm[iter] = rchk;                                                    /* :1549 */
    }
    double t1 = now_s();
    if (solve_seconds) *solve_seconds = t1 - t0;
    if (bnorm_out) *bnorm_out = bnorm;
    if (u_out) memcpy(u_out, L[0].u, sizeof(double) * L[0].N);
    for (int l = 0; l < levels; l++) {
        free(L[l].u); free(L[l].b); free(L[l].rv); free(L[l].work); free(L[l].dinv);
        mgo_csr_free(L[l].A); mgo_csr_free(L[l].R); mgo_csr_free(L[l].P);
    }
    free(L);
    return iter;
}


/* ------------------------------------------------------------------ */
/* -cycle 8: MultigridPetscPCMG, src/solver.c:1884-1989               */
/* ------------------------------------------------------------------ */
/* Outer KSPRICHARDSON (scale 1, KSP_NORM_UNPRECONDITIONED, rtol 1e-7, residual history, :1919-1924) preconditioned
 * by PCMG (:1926-1956).  PCMG is PETSc-internal and its PETSc version is unpinned; restated here is the textbook
 * multiplicative V-cycle PCMG documents (one cycle per application):
 *     x_fine = 0;  level: pre-smooth; r = b - A x; b_coarse = R r; x_coarse = 0; recurse; x += P x_coarse; post-smooth;
 *     coarsest: coarse solve from a zero guess.
 * Level solvers: the smoother of cfg (Richardson/Chebyshev + Jacobi, KSP_NORM_NONE), v0 sweeps on the levels and v1 on
 * the coarsest grid -- PETSc's own defaults (Chebyshev+SOR, LU) are NOT restated.  PARITY UNPINNED by the reference.
 * Levels are indexed as in the rest of this file: 0 = finest. */
static void pcmg_cycle(const mgo_vcycle_cfg *c, olevel *L, int l) {
    int levels = c->levels, dim = c->dim;
    if (l == levels - 1) { L[l].guess_nonzero = 0; smooth(c, &L[l], c->v1); return; }
    L[l].guess_nonzero = 1;                                   /* x was zeroed explicitly */
    smooth(c, &L[l], c->v0);
    residual(c, &L[l]);
    if (c->use_csr) mgo_csr_mult(L[l].R, L[l].rv, L[l + 1].b);
    else mgo_st_restrict(dim, L[l].n, L[l].n, L[l + 1].n, L[l].rv, NULL, L[l + 1].b);
    memset(L[l + 1].u, 0, sizeof(double) * L[l + 1].N);
    pcmg_cycle(c, L, l + 1);
    if (c->use_csr) {
        mgo_csr_mult(L[l].P, L[l + 1].u, L[l].rv);
        for (long q = 0; q < L[l].N; q++) L[l].u[q] = L[l].u[q] + L[l].rv[q];
    } else {
        mgo_st_prolong_add(dim, L[l].n, L[l].n, L[l + 1].n, L[l + 1].u, NULL, NULL, L[l].u);
    }
    smooth(c, &L[l], c->v0);
}

int mgo_pcmg(const mgo_vcycle_cfg *c, double *rnorm, double *u_out, double *bnorm_out, double *solve_seconds) {
    int levels = c->levels, dim = c->dim;
    olevel *L = (olevel *)calloc(levels, sizeof(olevel));
    for (int l = 0; l < levels; l++) {
        L[l].n = mgo_grid_n(c->npts, l);
        L[l].N = dim == 3 ? (long)L[l].n * L[l].n * L[l].n : (long)L[l].n * L[l].n;
        mgo_level_stencil(dim, c->npts, l, L[l].As, NULL);
        L[l].u = (double *)calloc(L[l].N, sizeof(double));
        L[l].b = (double *)calloc(L[l].N, sizeof(double));
        L[l].rv = (double *)calloc(L[l].N, sizeof(double));
        L[l].work = (double *)calloc(4 * L[l].N, sizeof(double));
        if (c->use_csr) {
            L[l].A = mgo_build_A(dim, c->npts, l);
            L[l].dinv = (double *)malloc(sizeof(double) * L[l].N);
            mgo_csr_diag_inv(L[l].A, L[l].dinv);
            if (l < levels - 1) { L[l].R = mgo_build_R(dim, c->npts, l); L[l].P = mgo_build_P(dim, c->npts, l); }
        }
    }
    const long N = L[0].N;
    double *bf = (double *)malloc(sizeof(double) * N), *x = (double *)calloc(N, sizeof(double)), *r = (double *)malloc(sizeof(double) * N);
    mgo_rhs(dim, c->npts, bf);
    memcpy(r, bf, sizeof(double) * N);                        /* zero guess: r = b */
    double rn = mgo_norm2(r, N), rn0 = rn;
    double rtol = c->rtol > 0 ? c->rtol : 1.e-7, ttol = rtol * rn0;
    if (ttol < 1.e-50) ttol = 1.e-50;
    rnorm[0] = rn;
    int its = 0;
    double t0 = now_s();
    while (its < c->maxiter && rn > ttol && !(rn >= 1.e5 * rn0)) {
        memcpy(L[0].b, r, sizeof(double) * N);                /* z = M^{-1} r */
        memset(L[0].u, 0, sizeof(double) * N);
        pcmg_cycle(c, L, 0);
        for (long q = 0; q < N; q++) x[q] = x[q] + 1.0 * L[0].u[q];      /* Richardson scale 1 */
        if (c->use_csr) mgo_residual_csr(L[0].A, bf, x, r);
        else mgo_st_residual(dim, L[0].n, L[0].n, L[0].As, bf, x, NULL, NULL, r);
        rn = mgo_norm2(r, N);
        its++;
        rnorm[its] = rn;
    }
    double t1 = now_s();
    if (solve_seconds) *solve_seconds = t1 - t0;
    if (bnorm_out) *bnorm_out = rn0;
    if (u_out) memcpy(u_out, x, sizeof(double) * N);
    for (int l = 0; l < levels; l++) {
        free(L[l].u); free(L[l].b); free(L[l].rv); free(L[l].work); free(L[l].dinv);
        mgo_csr_free(L[l].A); mgo_csr_free(L[l].R); mgo_csr_free(L[l].P);
    }
    free(L); free(bf); free(x); free(r);
    return its;
}


/* ------------------------------------------------------------------ */
/* -cycle 1 with one grid: MultigridIcycle, src/solver.c:1991-2060     */
/* ------------------------------------------------------------------ */
/* One KSPRICHARDSON on the level-0 operator, KSP_NORM_UNPRECONDITIONED, rtol 1e-7, residual history (:2011-2019):
 * x += scale * B (b - A x) with ||b - A x|| logged after every iteration.  Restated for one grid per level
 * (-grids 1 -levels 1); the multi-grid-per-level operator of :255-487 is not restated here. */
int mgo_icycle(const mgo_vcycle_cfg *c, double *rnorm, double *u_out, double *bnorm_out) {
    int dim = c->dim, n = mgo_grid_n(c->npts, 0);
    long N = dim == 3 ? (long)n * n * n : (long)n * n;
    double As[7];
    mgo_level_stencil(dim, c->npts, 0, As, NULL);
    double *b = (double *)malloc(sizeof(double) * N), *x = (double *)calloc(N, sizeof(double));
    double *y = (double *)malloc(sizeof(double) * N), *r = (double *)malloc(sizeof(double) * N);
    mgo_csr *A = c->use_csr ? mgo_build_A(dim, c->npts, 0) : NULL;
    double *dinv = NULL, *work = NULL;
    if (A) { dinv = (double *)malloc(sizeof(double) * N); mgo_csr_diag_inv(A, dinv); work = (double *)malloc(sizeof(double) * 2 * N); }
    mgo_rhs(dim, c->npts, b);
    memcpy(r, b, sizeof(double) * N);
    double rn = mgo_norm2(r, N), rn0 = rn;
    double rtol = c->rtol > 0 ? c->rtol : 1.e-7, ttol = rtol * rn0;
    if (ttol < 1.e-50) ttol = 1.e-50;
    rnorm[0] = rn;
    int its = 0;
    while (its < c->maxiter && rn > ttol && !(rn >= 1.e5 * rn0)) {
        if (A) {
            mgo_richardson_csr(A, dinv, b, x, 1, c->scale, its > 0, work);
            mgo_residual_csr(A, b, x, r);
        } else {
            mgo_st_jacobi(dim, n, n, As, c->scale, b, x, NULL, NULL, y, its == 0);
            double *t = x; x = y; y = t;
            mgo_st_residual(dim, n, n, As, b, x, NULL, NULL, r);
        }
        rn = mgo_norm2(r, N);
        its++;
        rnorm[its] = rn;
    }
    if (bnorm_out) *bnorm_out = rn0;
    if (u_out) memcpy(u_out, x, sizeof(double) * N);
    free(b); free(x); free(y); free(r); free(dinv); free(work); mgo_csr_free(A);
    return its;
}
