/*
 * mgpoisson.c -- the product's own command-line driver (C99), counterpart of the reference's
 * src/poisson.c:27-138 for sizes its O(N) host maps cannot reach (3-D, >= 4097^2) and for N GPUs.
 *
 *   mgpoisson [-dim 2|3] [-npts N] [-levels L] [-iter M] [-v v0,v1] [-ksp_type richardson|chebyshev]
 *             [-ksp_richardson_scale s] [-ksp_chebyshev_eigenvalues emin,emax] [-precision fp64|mixed]
 *             [-device d] [-options_file poisson.in] [-write_fields 0|1]
 *
 * Same option spelling as the reference where it has one (-npts -levels -iter -v; poisson.in syntax: '#'
 * comments, "-key value" lines); -grids is implied (= -levels: one grid per level), -cycle is 0; -mesh 0|1|2 (1, 2: 2-D).
 * Output mirrors what the reference prints: the PrintInfo block (src/poisson.c:165-214), error[0..2]
 * (src/solver.c:1333), "Relative residual" (:1354), "Solver walltime" (:1572), and the five files of
 * Postprocessing (src/solver.c:160-164,1331-1353): eData.dat, rData.dat, and -- the O(N) text dumps --
 * uData.dat, XgridData.dat, YgridData.dat in the reference's formats ("%.16e    " / "%lf    ", one grid row per
 * line; the grid files repeat coord[0][j], coord[1][i] with the UNSHIFTED index, as the reference does).
 * The O(N) dumps are written by default up to 1025^2 unknowns (2-D); -write_fields 1/0 forces them on/off
 * (3-D: rows of nx values, plane after plane; a "3-D extension", the reference is 2-D only).
 */
#include "mgsolve.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { char key[64], val[128]; } kv;
static kv g_kv[128];
static int g_nkv = 0;

static void put(const char *k, const char *v) {
    for (int q = 0; q < g_nkv; q++) if (!strcmp(g_kv[q].key, k)) { snprintf(g_kv[q].val, sizeof(g_kv[q].val), "%s", v); return; }
    if (g_nkv < 128) { snprintf(g_kv[g_nkv].key, 64, "%s", k); snprintf(g_kv[g_nkv].val, 128, "%s", v); g_nkv++; }
}
static const char *get(const char *k) {
    for (int q = 0; q < g_nkv; q++) if (!strcmp(g_kv[q].key, k)) return g_kv[q].val;
    return NULL;
}
static int is_key(const char *t) { return t[0] == '-' && ((t[1] >= 'a' && t[1] <= 'z') || (t[1] >= 'A' && t[1] <= 'Z')); }
static void tokens(char **t, int n) {
    for (int q = 0; q < n; q++) {
        if (!is_key(t[q])) continue;
        if (q + 1 < n && !is_key(t[q + 1])) { put(t[q], t[q + 1]); q++; } else put(t[q], "");
    }
}
static void read_file(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) return;
    char line[512];
    static char store[256][128];
    char *tok[256];
    int n = 0;
    while (fgets(line, sizeof(line), f)) {
        char *h = strchr(line, '#');
        if (h) *h = 0;
        for (char *t = strtok(line, " \t\r\n"); t && n < 256; t = strtok(NULL, " \t\r\n")) { snprintf(store[n], 128, "%s", t); tok[n] = store[n]; n++; }
    }
    fclose(f);
    tokens(tok, n);
}

int main(int argc, char **argv) {
    read_file("poisson.in");                          /* default options file of the reference (src/poisson.c:29) */
    for (int q = 1; q + 1 < argc; q++) if (!strcmp(argv[q], "-options_file")) read_file(argv[q + 1]);
    tokens(argv + 1, argc - 1);                       /* the command line overrides the files */

    mg_config c;
    mg_config_default(&c);
    const char *v;
    if ((v = get("-dim"))) c.dim = atoi(v);
    if ((v = get("-npts"))) c.npts = atoi(v);
    if ((v = get("-levels"))) c.levels = atoi(v);
    if ((v = get("-iter"))) c.maxiter = atoi(v);
    if ((v = get("-v"))) { int a = c.v[0], b = c.v[1]; if (sscanf(v, "%d,%d", &a, &b) >= 1) { c.v[0] = a; c.v[1] = b; } }
    if ((v = get("-ksp_type"))) c.ksp_type = !strcmp(v, "chebyshev") ? MG_KSP_CHEBYSHEV : MG_KSP_RICHARDSON;
    if ((v = get("-ksp_richardson_scale"))) c.scale = atof(v);
    if ((v = get("-ksp_chebyshev_eigenvalues"))) sscanf(v, "%lf,%lf", &c.emin, &c.emax);
    if ((v = get("-precision"))) c.precision = !strcmp(v, "mixed") ? MG_PREC_MIXED : MG_PREC_FP64;
    if ((v = get("-device"))) c.device = atoi(v);
    if ((v = get("-mg_fuse"))) c.fuse = atoi(v);                    /* tuning / testing: mg_config.fuse, .pair_min_n, .graph */
    if ((v = get("-mg_pair_min_n"))) c.pair_min_n = atoi(v);
    if ((v = get("-mg_graph"))) c.graph = atoi(v);
    if ((v = get("-pc_type")) && strcmp(v, "jacobi")) { fprintf(stderr, "mgpoisson: only -pc_type jacobi is built\n"); return 2; }
    if ((v = get("-cycle")) && atoi(v) != 0) { fprintf(stderr, "mgpoisson: only -cycle 0 (V-cycle) is built\n"); return 2; }
    if ((v = get("-mesh"))) c.mesh = atoi(v);
    /* the V-cycle has one grid per level (src/poisson.c:61-71 guards the other combinations): -grids, when given, must agree */
    if ((v = get("-grids")) && atoi(v) != c.levels) {
        fprintf(stderr, "mgpoisson: -grids %d differs from -levels %d: only one grid per level (the V-cycle case) is built\n", atoi(v), c.levels);
        return 2;
    }
    int map = 2;                                       /* poisson.in:11 */
    if ((v = get("-map"))) map = atoi(v);
    if (map < 0 || map > 2) { fprintf(stderr, "mgpoisson: -map must be 0, 1 or 2 (src/poisson.c:190-192)\n"); return 2; }

    mg_solver *s = NULL;
    if (mg_solver_create(&s, &c, NULL)) { fprintf(stderr, "mgpoisson: %s\n", mg_last_error()); return 1; }
    if (mg_solver_set_rhs_problem(s) || mg_solver_solve(s)) { fprintf(stderr, "mgpoisson: %s\n", mg_last_error()); return 1; }
    const int it = mg_solver_iterations(s);
    const double *rn = mg_solver_rnorm(s);
    double err[3];
    if (mg_solver_error_norms(s, err)) { fprintf(stderr, "mgpoisson: %s\n", mg_last_error()); return 1; }

    printf("rank = [0]; Solver walltime:               %lf\n", mg_solver_solve_seconds(s));
    for (int q = 0; q < 3; q++) printf("\nerror[%d] = %.16e\n", q, err[q]);
    printf("Relative residual = %.16e ", rn[it] / rn[0]);
    FILE *f = fopen("eData.dat", "w");
    if (f) { for (int q = 0; q < 3; q++) fprintf(f, "%.16e\n", err[q]); fclose(f); }
    f = fopen("rData.dat", "w");
    if (f) { for (int q = 0; q <= it; q++) fprintf(f, "%.16e ", rn[q] / rn[0]); fprintf(f, "\n"); fclose(f); }
    const int n = c.npts - 2;
    int dump = (c.dim == 2 && n <= 1023);
    if ((v = get("-write_fields"))) dump = atoi(v) != 0;
    if (dump) {
        const size_t N = (c.dim == 3) ? (size_t)n * n * n : (size_t)n * n;
        double *u = (double *)malloc(N * sizeof(double)), *x = (double *)malloc((size_t)c.npts * sizeof(double));
        if (!u || !x || mg_solver_get_solution(s, u)) { fprintf(stderr, "mgpoisson: cannot fetch the solution: %s\n", mg_last_error()); return 1; }
        double *y = (double *)malloc((size_t)c.npts * sizeof(double));
        if (!y) { fprintf(stderr, "mgpoisson: out of memory\n"); return 1; }
        x[0] = 0.0;                                       /* Coords, uniform branch: repeated addition (src/mesh.c:150-152) */
        for (int q = 1; q < c.npts - 1; q++) x[q] = x[q - 1] + 1.0 / (c.npts - 1);
        x[c.npts - 1] = 1.0;
        for (int q = 0; q < c.npts; q++) y[q] = x[q];
        for (int q = 1; q < c.npts - 1 && c.mesh; q++) {  /* y of the stretched meshes (src/mesh.c:165-169) */
            const double eta = q / (double)(c.npts - 1);
            y[q] = (c.mesh == 1) ? 1.0 - 1.0 * (cos(3.14159265358979323846 * 0.5 * eta)) : 0.0 + 1.0 * ((exp(2 * eta) - 1) / (exp(2) - 1));
        }
        FILE *fu = fopen("uData.dat", "w"), *fx = fopen("XgridData.dat", "w"), *fy = fopen("YgridData.dat", "w");
        if (fu && fx && fy) {
            const size_t rows = N / (size_t)n;
            for (size_t r = 0; r < rows; r++) {
                const int i = (int)(r % (size_t)n);
                for (int j = 0; j < n; j++) {
                    fprintf(fx, "%lf    ", x[j]);
                    fprintf(fy, "%lf    ", y[i]);
                    fprintf(fu, "%.16e    ", u[r * (size_t)n + j]);
                }
                fprintf(fx, "\n"); fprintf(fy, "\n"); fprintf(fu, "\n");
            }
        }
        if (fu) fclose(fu);
        if (fx) fclose(fx);
        if (fy) fclose(fy);
        free(u); free(x); free(y);
    }
    printf("=============================================================\n");
    if (c.dim == 2) printf("Size:\t\t\t\t%d x %d\n", c.npts, c.npts); else printf("Size:\t\t\t\t%d x %d x %d\n", c.npts, c.npts, c.npts);
    if (c.mesh == 0) printf("Mesh Type:\t\t\tUniform\n");          /* the reference prints no line for -mesh 2 (src/poisson.c:179-180) */
    if (c.mesh == 1) printf("Mesh Type:\t\t\tNon Uniform\n");
    printf("Number of grids:\t\t%d\n", c.levels);
    printf("Number of levels:\t\t%d\n", c.levels);
    printf("Number of grids per level:\t");
    for (int l = 0; l < c.levels; l++) printf("1\t");
    printf("\nNumber of unknowns per level:\t");
    for (int l = 0; l < c.levels; l++) { double n = mg_solver_level_n(s, l); printf("%.0f\t", c.dim == 3 ? n * n * n : n * n); }
    /* one grid per level: the three styles give the same (lexicographic) map; the line names the one that was asked for */
    printf("\nMapping style :\t\t\t%s\n", map == 0 ? "Grid after grid" : map == 1 ? "Through the grids" : "Local grid after grid");
    printf("Cycle :\t\t\t\tV-Cycle\n");
    printf("Number of smoothing steps :\t%d(fine) %d(coarsest)\n", c.v[0], c.v[1]);
    printf("Number of processes:\t\t1\n");
    printf("Number of iterations:\t\t%d\n", it);
    printf("DOF-updates per second:\t\t%.6e\n", mg_solver_dof_updates_per_cycle(s) * it / mg_solver_solve_seconds(s));
    printf("=============================================================\n");
    mg_solver_destroy(s);
    return 0;
}
