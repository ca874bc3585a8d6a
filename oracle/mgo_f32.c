/*
 * mgo_f32.c -- CPU ORACLE, fp32 leg (test infrastructure only, see mgo.h).
 *
 * Restates the mixed-precision cycle of BASELINE.json config 5 ("fp32 smoother sweeps with fp64
 * residual/correction").  The reference has no counterpart: this file
 * defines the arithmetic, in the same canonical order as the fp64 leg (mgo.c) evaluated in IEEE
 * binary32 without FMA, so that the HIP kernels (T = float) can be checked bit for bit.
 *
 *   outer (fp64):  r = b - A u ; ||r|| ; stop test of src/solver.c:1530 on the fp64 norms
 *   inner (fp32):  one V(v0,v0) cycle, v1 coarsest sweeps, for A e = (float) r from e = 0
 *   correction:    u = u + (double) e
 */
#include "mgo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static inline float st_row32(int n, const float *As, const float *x, int k, int i, int j) {
    long nn = (long)n * n, c = ((long)k * n + i) * n + j;
    float sum = 0.0f;
    if (k - 1 >= 0) sum += As[0] * x[c - nn];
    if (i - 1 >= 0) sum += As[1] * x[c - n];
    if (j - 1 >= 0) sum += As[2] * x[c - 1];
    sum += As[3] * x[c];
    if (j + 1 < n) sum += As[4] * x[c + 1];
    if (i + 1 < n) sum += As[5] * x[c + n];
    if (k + 1 < n) sum += As[6] * x[c + nn];
    return sum;
}

void mgo_st_jacobi_f32(int n, const float *As, float dinv, float scale, const float *b, const float *u,
                       float *unew, int zero_guess) {
#pragma omp parallel for collapse(2) schedule(static) if ((long)n * n * n > 32768)
    for (int k = 0; k < n; k++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                long c = ((long)k * n + i) * n + j;
                if (zero_guess) { float z = b[c] * dinv; unew[c] = scale * z; continue; }
                float t = st_row32(n, As, u, k, i, j);
                float r = b[c] - t;
                float z = r * dinv;
                unew[c] = u[c] + scale * z;
            }
}

void mgo_st_residual_f32(int n, const float *As, const float *b, const float *u, float *r) {
#pragma omp parallel for collapse(2) schedule(static) if ((long)n * n * n > 32768)
    for (int k = 0; k < n; k++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                long c = ((long)k * n + i) * n + j;
                r[c] = b[c] - st_row32(n, As, u, k, i, j);
            }
}

void mgo_st_restrict_f32(int nf, const float *rf, float *bc) {
    int nc = (nf - 1) / 2;
    long nnf = (long)nf * nf;
    static const float w2[9] = {0.0625f, 0.125f, 0.0625f, 0.125f, 0.25f, 0.125f, 0.0625f, 0.125f, 0.0625f};
    static const float w1[3] = {0.25f, 0.5f, 0.25f};
#pragma omp parallel for collapse(2) schedule(static) if ((long)nc * nc * nc > 32768)
    for (int k1 = 0; k1 < nc; k1++)
        for (int i1 = 0; i1 < nc; i1++)
            for (int j1 = 0; j1 < nc; j1++) {
                float sum = 0.0f;
                for (int dk = 0; dk < 3; dk++)
                    for (int di = 0; di < 3; di++)
                        for (int dj = 0; dj < 3; dj++)
                            sum += (w1[dk] * w2[di * 3 + dj]) * rf[(2 * k1 + dk) * nnf + (long)(2 * i1 + di) * nf + 2 * j1 + dj];
                bc[((long)k1 * nc + i1) * nc + j1] = sum;
            }
}

void mgo_st_prolong_add_f32(int nf, const float *uc, float *uf) {
    int nc = (nf - 1) / 2;
    long ncc = (long)nc * nc;
#pragma omp parallel for collapse(2) schedule(static) if ((long)nf * nf * nf > 32768)
    for (int k = 0; k < nf; k++)
        for (int i = 0; i < nf; i++)
            for (int j = 0; j < nf; j++) {
                int kc0 = (k & 1) ? (k - 1) / 2 : k / 2 - 1, kc1 = (k & 1) ? kc0 : k / 2;
                int ic0 = (i & 1) ? (i - 1) / 2 : i / 2 - 1, ic1 = (i & 1) ? ic0 : i / 2;
                int jc0 = (j & 1) ? (j - 1) / 2 : j / 2 - 1, jc1 = (j & 1) ? jc0 : j / 2;
                float sum = 0.0f;
                for (int kc = kc0; kc <= kc1; kc++) {
                    if (kc < 0 || kc >= nc) continue;
                    float wk = (k & 1) ? 1.0f : 0.5f;
                    for (int ic = ic0; ic <= ic1; ic++) {
                        if (ic < 0 || ic >= nc) continue;
                        float wi = (i & 1) ? 1.0f : 0.5f;
                        for (int jc = jc0; jc <= jc1; jc++) {
                            if (jc < 0 || jc >= nc) continue;
                            float wj = (j & 1) ? 1.0f : 0.5f;
                            sum += (wk * (wi * wj)) * uc[kc * ncc + (long)ic * nc + jc];
                        }
                    }
                }
                long c = ((long)k * nf + i) * nf + j;
                uf[c] = uf[c] + sum;
            }
}

typedef struct { int n; long N; float As[7], dinv; float *u, *b, *rv, *tmp; } l32;

static void smooth32(l32 *L, int its, float scale, int guess_nonzero) {
    float *cur = L->u, *nxt = L->tmp;
    for (int it = 0; it < its; it++) {
        mgo_st_jacobi_f32(L->n, L->As, L->dinv, scale, L->b, cur, nxt, it == 0 && !guess_nonzero);
        float *t = cur; cur = nxt; nxt = t;
    }
    if (its == 0 && !guess_nonzero) memset(L->u, 0, sizeof(float) * L->N);
    if (cur != L->u) memcpy(L->u, cur, sizeof(float) * L->N);
}

/* returns outer iterations; rnorm: fp64 ||b - A u|| per outer iteration */
int mgo_vcycle_mixed(const mgo_vcycle_cfg *c, double *rnorm, double *u_out, double *bnorm_out, double *solve_seconds) {
    const int levels = c->levels, dim = 3;
    if (c->dim != 3 || c->ksp_type != 0) return -1;
    int n0 = mgo_grid_n(c->npts, 0);
    long N0 = (long)n0 * n0 * n0;
    double As0[7];
    mgo_level_stencil(dim, c->npts, 0, As0, NULL);
    double *u = (double *)calloc(N0, sizeof(double)), *b = (double *)calloc(N0, sizeof(double)), *r = (double *)calloc(N0, sizeof(double));
    l32 *L = (l32 *)calloc(levels, sizeof(l32));
    for (int l = 0; l < levels; l++) {
        double As[7];
        L[l].n = mgo_grid_n(c->npts, l);
        L[l].N = (long)L[l].n * L[l].n * L[l].n;
        mgo_level_stencil(dim, c->npts, l, As, NULL);
        for (int q = 0; q < 7; q++) L[l].As[q] = (float)As[q];
        L[l].dinv = (float)(1.0 / As[3]);
        L[l].u = (float *)calloc(L[l].N, sizeof(float)); L[l].b = (float *)calloc(L[l].N, sizeof(float));
        L[l].rv = (float *)calloc(L[l].N, sizeof(float)); L[l].tmp = (float *)calloc(L[l].N, sizeof(float));
    }
    const float scale = (float)c->scale;
    mgo_rhs(dim, c->npts, b);
    double bnorm = mgo_norm2(b, N0);
    mgo_st_residual(dim, n0, n0, As0, b, u, NULL, NULL, r);
    double rchk = mgo_norm2(r, N0);
    rnorm[0] = rchk;
    int iter = 0;
    double rtol = c->rtol > 0 ? c->rtol : 1.e-7;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        if (c->fixed_cycles > 0) { if (iter >= c->fixed_cycles) break; }
        else if (!(iter < c->maxiter && 100000000 * bnorm > rchk && rchk > rtol * bnorm)) break;
        for (long q = 0; q < N0; q++) L[0].b[q] = (float)r[q];
        smooth32(&L[0], c->v0, scale, 0);
        for (int l = 1; l < levels; l++) {
            mgo_st_residual_f32(L[l - 1].n, L[l - 1].As, L[l - 1].b, L[l - 1].u, L[l - 1].rv);
            mgo_st_restrict_f32(L[l - 1].n, L[l - 1].rv, L[l].b);
            smooth32(&L[l], l == levels - 1 ? c->v1 : c->v0, scale, 0);
        }
        for (int l = levels - 2; l >= 0; l--) {
            mgo_st_prolong_add_f32(L[l].n, L[l + 1].u, L[l].u);
            smooth32(&L[l], c->v0, scale, 1);
        }
        for (long q = 0; q < N0; q++) u[q] = u[q] + (double)L[0].u[q];
        mgo_st_residual(dim, n0, n0, As0, b, u, NULL, NULL, r);
        rchk = mgo_norm2(r, N0);
        iter++;
        rnorm[iter] = rchk;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (solve_seconds) *solve_seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    if (bnorm_out) *bnorm_out = bnorm;
    if (u_out) memcpy(u_out, u, sizeof(double) * N0);
    for (int l = 0; l < levels; l++) { free(L[l].u); free(L[l].b); free(L[l].rv); free(L[l].tmp); }
    free(L); free(u); free(b); free(r);
    return iter;
}
