// tests/mock_mgk.cpp -- TEST INFRASTRUCTURE ONLY.  Host-memory stand-ins for the kernel ABI (include/mgk.h) so that the product's
// host C (csrc/petsc_shim.c, mg_solver.c, mg_comm.c, driver/mgpoisson.c) -- option parsing, MatSetValue -> CSR, stencil
// recognition, PCMG set-up / tear-down, slab and ghost-plane bookkeeping, graph capture, deferred norms -- can run under
// -fsanitize=address,undefined in `pytest -m "not gpu"` (tests/test_host_sanitized.py).  "Device" memory is calloc'ed host
// memory, streams are tags, a captured graph is a list of closures.  Every operation touches exactly the extents the HIP
// kernel touches (ghost rows / planes, far planes included), in the canonical arithmetic of DESIGN.md section 2, so a wrong
// geometry, pointer or plane count in the host code is an ASan report, and results can be compared with the oracle.
// NOT a CPU fallback: it is never built into, linked with or loaded by anything under multigrid_petsc_amd/.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <sched.h>
#include <time.h>
#include "mgk.h"

struct mgk_ctx {
    int device;
    char cs, ms;                       // stream tags
    double *defer;
    bool capturing;
    std::vector<std::function<void()>> rec;
    std::vector<double> partials;
};
struct mock_graph { std::vector<std::function<void()>> ops; };
static thread_local char g_err[256] = "ok";
static int fail(int code, const char *what) { snprintf(g_err, sizeof(g_err), "%s (mock, code %d)", what, code); return code; }
template <class F> static int run(mgk_ctx *c, F f) { if (c->capturing) c->rec.push_back(f); else f(); return 0; }

extern "C" {
const char *mgk_last_error(void) { return g_err; }
int mgk_device_count(void) { return 1; }
int mgk_set_device(int) { return 0; }
int mgk_geom_init(mgk_geom *g, int dim, int nx, int ny, int nz) {
    if (!g || (dim != 2 && dim != 3) || nx < 1 || ny < 1 || nz < 1 || (nx & 1) == 0) return fail(MGK_EINVAL, "mgk_geom_init");
    if (dim == 2) nz = 1;
    g->dim = dim; g->nx = nx; g->ny = ny; g->nz = nz;
    g->pitch = ((MGK_XOFF + nx + 1 + 15) / 16) * 16;
    g->plane = (long)g->pitch * (ny + 2);
    if (dim == 3) { g->org = g->plane + g->pitch + MGK_XOFF; g->total = g->plane * (nz + 2) + g->pitch; }
    else { g->org = g->pitch + MGK_XOFF; g->total = g->plane + g->pitch; }
    return 0;
}
int mgk_geom_init_f32(mgk_geom *g, int dim, int nx, int ny, int nz) {
    if (!g || dim != 3 || nx < 1 || ny < 1 || nz < 1 || (nx & 1) == 0) return fail(MGK_EINVAL, "mgk_geom_init_f32");
    g->dim = dim; g->nx = nx; g->ny = ny; g->nz = nz;
    g->pitch = ((32 + nx + 1 + 31) / 32) * 32;
    g->plane = (long)g->pitch * (ny + 2);
    g->org = g->plane + g->pitch + 32;
    g->total = g->plane * (nz + 2) + g->pitch;
    return 0;
}
int mgk_ctx_create(mgk_ctx **out, int device) {
    if (!out) return fail(MGK_EINVAL, "mgk_ctx_create");
    mgk_ctx *c = new mgk_ctx();
    c->device = device; c->defer = nullptr; c->capturing = false; c->partials.assign(16384, 0.0);
    *out = c;
    return 0;
}
void mgk_ctx_destroy(mgk_ctx *c) { delete c; }
int mgk_ctx_set_chunk_planes(mgk_ctx *c, int planes) { return (c && planes >= 0) ? 0 : fail(MGK_EINVAL, "mgk_ctx_set_chunk_planes"); }
void *mgk_stream_compute(mgk_ctx *c) { return &c->cs; }
void *mgk_stream_comm(mgk_ctx *c) { return &c->ms; }
int mgk_malloc(mgk_ctx *, void **p, size_t bytes) { *p = calloc(1, bytes ? bytes : 8); return *p ? 0 : fail(MGK_EINVAL, "mgk_malloc"); }
int mgk_free(mgk_ctx *, void *p) { free(p); return 0; }
int mgk_host_alloc(mgk_ctx *, void **p, size_t bytes) { *p = calloc(1, bytes ? bytes : 8); return *p ? 0 : fail(MGK_EINVAL, "mgk_host_alloc"); }
int mgk_host_free(mgk_ctx *, void *p) { free(p); return 0; }
int mgk_memset0(mgk_ctx *c, void *p, size_t bytes, void *) { return run(c, [=] { memset(p, 0, bytes); }); }
int mgk_h2d(mgk_ctx *, void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
int mgk_d2h(mgk_ctx *, void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
int mgk_d2h_async(mgk_ctx *, void *d, const void *s, size_t n, void *) { memcpy(d, s, n); return 0; }
int mgk_h2d_async(mgk_ctx *, void *d, const void *s, size_t n, void *) { memcpy(d, s, n); return 0; }
int mgk_d2d(mgk_ctx *c, void *d, const void *s, size_t n, void *) { return run(c, [=] { memmove(d, s, n); }); }
int mgk_sync(mgk_ctx *, void *) { return 0; }
int mgk_paced_copy(mgk_ctx *c, void *d, const void *s, size_t n, double us, int blocks, void *) {
    if (!c || !d || !s || (n & 15) || us < 0 || blocks < 1) return fail(MGK_EINVAL, "mgk_paced_copy");
    return run(c, [=] { memmove(d, s, n); });
}
int mgk_delay_us(mgk_ctx *, double us, void *) { return us < 0 ? fail(MGK_EINVAL, "mgk_delay_us") : 0; }
int mgk_timer_create(mgk_ctx *, void **t) { *t = malloc(8); return 0; }
int mgk_timer_start(mgk_ctx *, void *, void *) { return 0; }
int mgk_timer_stop(mgk_ctx *, void *, void *) { return 0; }
int mgk_timer_elapsed_ms(mgk_ctx *, void *, double *ms) { *ms = 1.0; return 0; }
void mgk_timer_destroy(mgk_ctx *, void *t) { free(t); }
int mgk_stream_wait(mgk_ctx *, void *, void *) { return 0; }
int mgk_capture_begin(mgk_ctx *c) { if (c->capturing) return fail(MGK_EINVAL, "nested capture"); c->capturing = true; c->rec.clear(); return 0; }
int mgk_capture_end(mgk_ctx *c, void **ge) {
    if (!c->capturing) return fail(MGK_EINVAL, "capture_end without begin");
    mock_graph *g = new mock_graph();
    g->ops.swap(c->rec);
    c->capturing = false;
    *ge = g;
    return 0;
}
int mgk_graph_launch(mgk_ctx *c, void *ge) {
    if (c->capturing) return fail(MGK_EINVAL, "graph launch inside a capture");
    for (auto &f : ((mock_graph *)ge)->ops) f();
    return 0;
}
void mgk_graph_destroy(mgk_ctx *, void *ge) { delete (mock_graph *)ge; }
int mgk_defer_result(mgk_ctx *c, double *slot) { c->defer = slot; return 0; }
void mgk_set_tuning(int, int) {}
int mgk_tail_max_n(int dim) { return dim == 3 ? 15 : 63; }
void mgk_debug_tail_stamps(long long *) {}
}   // extern "C"

// ---------------------------------------------------------------------------------------------
// field helpers (f points at the START of the allocation, like the ABI's arguments)
// ---------------------------------------------------------------------------------------------
template <class T> static inline T &at(T *f, const mgk_geom &g, long k, long i, long j) {
    return f[g.org + (g.dim == 3 ? k * g.plane : 0) + i * (long)g.pitch + j];
}
template <class T> static void geom3(mgk_geom *g, int nx, int ny, int nz) {       // a 3-D field of T with the layout of the ABI
    if (sizeof(T) == 8) mgk_geom_init(g, 3, nx, ny, nz); else mgk_geom_init_f32(g, 3, nx, ny, nz);
}
enum { M_JACOBI, M_RESIDUAL, M_APPLY, M_CHEBY };
// rows / planes [zbeg, zend) of the marching axis (3-D: z planes, 2-D: grid rows); ctab/dtab: 2-D per-row coefficients
template <class T>
static void st_op(int mode, const mgk_geom &g, const double *coef, double dinv_, double scale_, double ckm1_, double ck_, double cz_,
                  const T *b, const T *u, const T *aux, T *out, int zbeg, int zend, const double *ctab = nullptr, const double *dtab = nullptr) {
    const T scale = (T)scale_, ckm1 = (T)ckm1_, ck = (T)ck_, cz = (T)cz_;
    T c[7] = {0, 0, 0, 0, 0, 0, 0};
    if (coef) for (int q = 0; q < (g.dim == 3 ? 7 : 5); q++) c[q] = (T)coef[q];
    const int k0 = g.dim == 3 ? zbeg : 0, k1 = g.dim == 3 ? zend : 1, i0 = g.dim == 3 ? 0 : zbeg, i1 = g.dim == 3 ? g.ny : zend;
    for (int k = k0; k < k1; k++)
        for (int i = i0; i < i1; i++) {
            T dinv = (T)dinv_;
            if (ctab) { for (int q = 0; q < 5; q++) c[q] = (T)ctab[5 * (long)i + q]; if (dtab) dinv = (T)dtab[i]; }
            for (int j = 0; j < g.nx; j++) {
                T t;
                if (g.dim == 3) {
                    t = c[0] * at(u, g, k - 1, i, j);
                    t = t + c[1] * at(u, g, k, i - 1, j);
                    t = t + c[2] * at(u, g, k, i, j - 1);
                    t = t + c[3] * at(u, g, k, i, j);
                    t = t + c[4] * at(u, g, k, i, j + 1);
                    t = t + c[5] * at(u, g, k, i + 1, j);
                    t = t + c[6] * at(u, g, k + 1, i, j);
                } else {
                    t = c[0] * at(u, g, 0, i - 1, j);
                    t = t + c[1] * at(u, g, 0, i, j - 1);
                    t = t + c[2] * at(u, g, 0, i, j);
                    t = t + c[3] * at(u, g, 0, i, j + 1);
                    t = t + c[4] * at(u, g, 0, i + 1, j);
                }
                const T res = (b ? at(b, g, k, i, j) : (T)0) - t;
                T o;
                if (mode == M_JACOBI) { const T zz = res * dinv; o = at(u, g, k, i, j) + scale * zz; }
                else if (mode == M_CHEBY) { const T zz = res * dinv; o = (ckm1 * at(aux, g, k, i, j) + ck * at(u, g, k, i, j)) + cz * zz; }
                else if (mode == M_APPLY) o = t;
                else o = res;
                at(out, g, k, i, j) = o;
            }
        }
}
template <class T> static double sumsq_field(const mgk_geom &g, const T *x, int zbeg, int zend) {
    long double s = 0;
    const int k0 = g.dim == 3 ? zbeg : 0, k1 = g.dim == 3 ? zend : 1, i0 = g.dim == 3 ? 0 : zbeg, i1 = g.dim == 3 ? g.ny : zend;
    for (int k = k0; k < k1; k++) for (int i = i0; i < i1; i++) for (int j = 0; j < g.nx; j++) { const double v = (double)at(x, g, k, i, j); s += v * v; }
    return (double)s;
}
static void deliver(mgk_ctx *c, double v, double *host) {       // a single-value reduction: to the deferred slot or to the host
    if (c->defer) { double *slot = c->defer; run(c, [=] { *slot = v; }); *host = 0.0; } else *host = v;
}
// full weighting of the coarse planes [kc0, kc1); dkmax < 3: the last plane of the range is left partial (dk < dkmax)
template <class T>
static void restrict_fw(const mgk_geom &gf, const mgk_geom &gc, const T *rf, T *bc, int kc0, int kc1, int dkmax_last = 3, bool accumulate_dk2 = false) {
    const T w2[3][3] = {{(T)0.0625, (T)0.125, (T)0.0625}, {(T)0.125, (T)0.25, (T)0.125}, {(T)0.0625, (T)0.125, (T)0.0625}};
    const T w1[3] = {(T)0.25, (T)0.5, (T)0.25};
    for (int kc = kc0; kc < kc1; kc++)
        for (int ic = 0; ic < gc.ny; ic++)
            for (int jc = 0; jc < gc.nx; jc++) {
                T sum = accumulate_dk2 ? at(bc, gc, kc, ic, jc) : (T)0;
                if (gf.dim == 3) {
                    const int d0 = accumulate_dk2 ? 2 : 0, d1 = (kc == kc1 - 1 && !accumulate_dk2) ? dkmax_last : 3;
                    for (int dk = d0; dk < d1; dk++)
                        for (int di = 0; di < 3; di++)
                            for (int dj = 0; dj < 3; dj++) sum += (w1[dk] * w2[di][dj]) * at(rf, gf, 2 * kc + dk, 2 * ic + di, 2 * jc + dj);
                } else {
                    for (int di = 0; di < 3; di++)
                        for (int dj = 0; dj < 3; dj++) sum += w2[di][dj] * at(rf, gf, 0, 2 * ic + di, 2 * jc + dj);
                }
                at(bc, gc, kc, ic, jc) = sum;
            }
}
// P uc at one fine point (ghost indices allowed: -1 and n are odd points whose single parent is the coarse ghost)
template <class T> static T prolong_at(const mgk_geom &gf, const mgk_geom &gc, const T *uc, int k, int i, int x) {
    const int iodd = i & 1, kodd = k & 1, xodd = x & 1, d3 = gf.dim == 3;
    const int ic0 = iodd ? (i - 1) / 2 : i / 2 - 1, nic = iodd ? 1 : 2;
    const int kc0 = d3 ? (kodd ? (k - 1) / 2 : k / 2 - 1) : 0, nkc = d3 ? (kodd ? 1 : 2) : 1;
    const int jc0 = xodd ? (x - 1) / 2 : x / 2 - 1, njc = xodd ? 1 : 2;
    const T wi = iodd ? (T)1 : (T)0.5, wk = d3 ? (kodd ? (T)1 : (T)0.5) : (T)1, wj = xodd ? (T)1 : (T)0.5;
    const T w = d3 ? wk * (wi * wj) : wi * wj;
    T s = (T)0;
    for (int qk = 0; qk < nkc; qk++) for (int qi = 0; qi < nic; qi++) for (int qj = 0; qj < njc; qj++) s += w * at(uc, gc, kc0 + qk, ic0 + qi, jc0 + qj);
    (void)gc;
    return s;
}
static bool xfer_ok(const mgk_geom *gf, const mgk_geom *gc) {
    return gf && gc && gf->dim == gc->dim && gf->nx == 2 * gc->nx + 1 && gf->ny == 2 * gc->ny + 1 &&
           (gf->dim == 2 || gf->nz == 2 * gc->nz + 1 || gf->nz == 2 * gc->nz);
}
template <class T> static std::vector<T> corrected(const mgk_geom &gf, const mgk_geom &gc, const T *uc, const T *u) {
    std::vector<T> t(u, u + gf.total);
    const int k0 = gf.dim == 3 ? -1 : 0, k1 = gf.dim == 3 ? gf.nz + 1 : 1;
    for (int k = k0; k < k1; k++)
        for (int i = -1; i <= gf.ny; i++)
            for (int j = 0; j < gf.nx; j++) at(t.data(), gf, k, i, j) = at(t.data(), gf, k, i, j) + prolong_at(gf, gc, uc, gf.dim == 3 ? k : 1, i, j);
    return t;
}
#define NMARCH(g) ((g)->dim == 3 ? (g)->nz : (g)->ny)
#define CHKRANGE(g, z0, z1, what) if ((z0) < 0 || (z1) > NMARCH(g) || (z0) >= (z1)) return fail(MGK_EINVAL, what)

template <class T> static int jacobi_range(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const T *b, const T *u, T *o, int z0, int z1) {
    if (!c || !g || !coef || !b || !u || !o || u == o) return fail(MGK_EINVAL, "mgk_jacobi: bad arguments");
    CHKRANGE(g, z0, z1, "mgk_jacobi_range: range");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 7);
    return run(c, [=] { st_op<T>(M_JACOBI, G, k.data(), dinv, scale, 0, 0, 0, b, u, (const T *)nullptr, o, z0, z1); });
}
template <class T> static int residual_range(mgk_ctx *c, const mgk_geom *g, const double *coef, const T *b, const T *u, T *r, int z0, int z1) {
    if (!c || !g || !coef || !b || !u || !r || u == r) return fail(MGK_EINVAL, "mgk_residual: bad arguments");
    CHKRANGE(g, z0, z1, "mgk_residual_range: range");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 7);
    return run(c, [=] { st_op<T>(M_RESIDUAL, G, k.data(), 1, 1, 0, 0, 0, b, u, (const T *)nullptr, r, z0, z1); });
}
template <class T> static int jacobi_zero(mgk_ctx *c, const mgk_geom *g, double dinv, double scale, const T *b, T *o, const double *dtab) {
    if (!c || !g || !b || !o) return fail(MGK_EINVAL, "mgk_jacobi_zero: bad arguments");
    const mgk_geom G = *g;
    return run(c, [=] {
        for (int k = 0; k < G.nz; k++) for (int i = 0; i < G.ny; i++) for (int j = 0; j < G.nx; j++) {
            const T zx = at(b, G, k, i, j) * (T)(dtab ? dtab[i] : dinv);
            at(o, G, k, i, j) = (T)scale * zx;
        }
    });
}
template <class T> static int restrict_api(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const T *rf, T *bc) {
    if (!c || !rf || !bc || !xfer_ok(gf, gc)) return fail(MGK_EINVAL, "mgk_restrict_fw: bad arguments");
    const mgk_geom F = *gf, Cg = *gc;
    return run(c, [=] { restrict_fw<T>(F, Cg, rf, bc, 0, Cg.dim == 3 ? Cg.nz : 1); });
}
template <class T> static int prolong_add_api(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const T *uc, T *uf) {
    if (!c || !uc || !uf || !xfer_ok(gf, gc)) return fail(MGK_EINVAL, "mgk_prolong_add: bad arguments");
    const mgk_geom F = *gf, Cg = *gc;
    return run(c, [=] {
        for (int k = 0; k < (F.dim == 3 ? F.nz : 1); k++) for (int i = 0; i < F.ny; i++) for (int j = 0; j < F.nx; j++)
            at(uf, F, k, i, j) = at(uf, F, k, i, j) + prolong_at(F, Cg, uc, F.dim == 3 ? k : 1, i, j);
    });
}
template <class T> static int prolong_jacobi_api(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                                 const T *b, const T *uc, const T *u, T *o, int z0, int z1) {
    if (!c || !coef || !b || !uc || !u || !o || u == o || !xfer_ok(gf, gc)) return fail(MGK_EINVAL, "mgk_prolong_jacobi: bad arguments");
    CHKRANGE(gf, z0, z1, "mgk_prolong_jacobi_range: range");
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(coef, coef + 7);
    return run(c, [=] { std::vector<T> t = corrected<T>(F, Cg, uc, u); st_op<T>(M_JACOBI, F, k.data(), dinv, scale, 0, 0, 0, b, t.data(), (const T *)nullptr, o, z0, z1); });
}
// two sweeps; slab: planes -2 / nz+1 of u come from the far field's ghost planes, b on the ghost planes from b's ghost planes
template <class T> static int jacobi2_api(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                                          const T *b, const T *u, T *o, const T *far, int has_lo, int has_hi, int z0, int z1) {
    if (!c || !g || !coef || !b || !u || !o || u == o || g->dim != 3) return fail(MGK_EINVAL, "mgk_jacobi2: bad arguments");
    if (z0 < 0 || z1 > g->nz || z0 >= z1) return fail(MGK_EINVAL, "mgk_jacobi2: range");
    if ((has_lo || has_hi) && (!far || !gfar || gfar->nz != 2 || gfar->nx != g->nx || gfar->ny != g->ny || gfar->pitch != g->pitch))
        return fail(MGK_EINVAL, "mgk_jacobi2_slab: far field");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        // extended slab: planes -2 .. nz+1 of u, -1 .. nz of b, as planes 0 .. nz+3 / 1 .. nz+2 of a field with nz+2 interior planes
        mgk_geom E; geom3<T>(&E, G.nx, G.ny, G.nz + 2);
        std::vector<T> ue(E.total, (T)0), be(E.total, (T)0), w(E.total, (T)0);
        for (int kk = -1; kk <= G.nz; kk++)
            memcpy(&ue[(size_t)((kk + 2) * E.plane)], u + (size_t)((kk + 1) * G.plane), sizeof(T) * (size_t)G.plane);
        if (has_lo) memcpy(&ue[0], far, sizeof(T) * (size_t)G.plane);                                        // far's lo ghost plane
        if (has_hi) memcpy(&ue[(size_t)((G.nz + 3) * E.plane)], far + (size_t)(3 * G.plane), sizeof(T) * (size_t)G.plane);   // far's hi ghost plane
        for (int kk = (has_lo ? -1 : 0); kk <= (has_hi ? G.nz : G.nz - 1); kk++)
            memcpy(&be[(size_t)((kk + 2) * E.plane)], b + (size_t)((kk + 1) * G.plane), sizeof(T) * (size_t)G.plane);
        const int s0 = has_lo ? 0 : 1, s1 = has_hi ? G.nz + 2 : G.nz + 1;      // first sweep on the planes that are real
        st_op<T>(M_JACOBI, E, k.data(), dinv, scale, 0, 0, 0, be.data(), ue.data(), (const T *)nullptr, w.data(), s0, s1);
        std::vector<T> o2(E.total, (T)0);
        st_op<T>(M_JACOBI, E, k.data(), dinv, scale, 0, 0, 0, be.data(), w.data(), (const T *)nullptr, o2.data(), z0 + 1, z1 + 1);
        for (int kk = z0; kk < z1; kk++) for (int i = 0; i < G.ny; i++)
            memcpy(&at(o, G, kk, i, 0), &at(o2.data(), E, kk + 1, i, 0), sizeof(T) * (size_t)G.nx);
    });
}
// b_c = R (b - A u) for the coarse planes [kc0, kc1); far_hi: plane nz+1 of u (inner slab completes its last coarse plane)
template <class T> static int rr_api(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const T *b, const T *u, T *bc, T *uc0,
                                     double dinv_c, double scale_c, int kc0, int kc1, const T *far_hi_plane) {
    if (!c || !coef || !b || !u || !bc || !xfer_ok(gf, gc) || gf->dim != 3) return fail(MGK_EINVAL, "mgk_residual_restrict: bad arguments");
    if (kc0 < 0 || kc1 > gc->nz || kc0 >= kc1) return fail(MGK_EINVAL, "mgk_residual_restrict: coarse range");
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        const bool inner = (F.nz == 2 * Cg.nz);
        mgk_geom E; geom3<T>(&E, F.nx, F.ny, F.nz + 1);                       // one more plane: the residual of plane nz
        std::vector<T> ue(E.total, (T)0), be(E.total, (T)0), r(E.total, (T)0);
        memcpy(ue.data(), u, sizeof(T) * (size_t)(F.plane * (F.nz + 2)));
        memcpy(be.data(), b, sizeof(T) * (size_t)(F.plane * (F.nz + 2)));
        const bool complete = inner && far_hi_plane;
        if (complete) memcpy(&ue[(size_t)((F.nz + 2) * E.plane)], far_hi_plane, sizeof(T) * (size_t)F.plane);
        st_op<T>(M_RESIDUAL, E, k.data(), 1, 1, 0, 0, 0, be.data(), ue.data(), (const T *)nullptr, r.data(), 0, complete ? F.nz + 1 : F.nz);
        mgk_geom Ec = Cg;
        restrict_fw<T>(E, Ec, r.data(), bc, kc0, kc1, (inner && !complete && kc1 == Cg.nz) ? 2 : 3);
        if (uc0) for (int kc = kc0; kc < kc1; kc++) for (int i = 0; i < Cg.ny; i++) for (int j = 0; j < Cg.nx; j++) {
            const T zq = at(bc, Cg, kc, i, j) * (T)dinv_c; at(uc0, Cg, kc, i, j) = (T)scale_c * zq;
        }
    });
}

extern "C" {
int mgk_jacobi_range_f64(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double s, const double *b, const double *u, double *o, int z0, int z1, void *) { return jacobi_range<double>(c, g, k, d, s, b, u, o, z0, z1); }
int mgk_jacobi_range_f32(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double s, const float *b, const float *u, float *o, int z0, int z1, void *) { return jacobi_range<float>(c, g, k, d, s, b, u, o, z0, z1); }
int mgk_jacobi_f64(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double s, const double *b, const double *u, double *o, void *) { return g ? jacobi_range<double>(c, g, k, d, s, b, u, o, 0, NMARCH(g)) : fail(MGK_EINVAL, "g"); }
int mgk_jacobi_f32(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double s, const float *b, const float *u, float *o, void *) { return g ? jacobi_range<float>(c, g, k, d, s, b, u, o, 0, g->nz) : fail(MGK_EINVAL, "g"); }
int mgk_jacobi_zero_f64(mgk_ctx *c, const mgk_geom *g, double d, double s, const double *b, double *o, void *) { return jacobi_zero<double>(c, g, d, s, b, o, nullptr); }
int mgk_jacobi_zero_f32(mgk_ctx *c, const mgk_geom *g, double d, double s, const float *b, float *o, void *) { return jacobi_zero<float>(c, g, d, s, b, o, nullptr); }
int mgk_jacobi_zero_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *dtab, double s, const double *b, double *o, void *) { return (g && g->dim == 2 && dtab) ? jacobi_zero<double>(c, g, 1.0, s, b, o, dtab) : fail(MGK_EINVAL, "rowcoef"); }
int mgk_cheby_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double ckm1, double ck, double cz, const double *b, const double *pk, const double *pkm1, double *o, void *) {
    if (!c || !g || !coef || !b || !pk || !pkm1 || !o || pk == o || pkm1 == o) return fail(MGK_EINVAL, "mgk_cheby_f64");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 7);
    return run(c, [=] { st_op<double>(M_CHEBY, G, k.data(), dinv, 1, ckm1, ck, cz, b, pk, pkm1, o, 0, NMARCH(&G)); });
}
int mgk_residual_range_f64(mgk_ctx *c, const mgk_geom *g, const double *k, const double *b, const double *u, double *r, int z0, int z1, void *) { return residual_range<double>(c, g, k, b, u, r, z0, z1); }
int mgk_residual_range_f32(mgk_ctx *c, const mgk_geom *g, const double *k, const float *b, const float *u, float *r, int z0, int z1, void *) { return residual_range<float>(c, g, k, b, u, r, z0, z1); }
int mgk_residual_f64(mgk_ctx *c, const mgk_geom *g, const double *k, const double *b, const double *u, double *r, void *) { return g ? residual_range<double>(c, g, k, b, u, r, 0, NMARCH(g)) : fail(MGK_EINVAL, "g"); }
int mgk_residual_f32(mgk_ctx *c, const mgk_geom *g, const double *k, const float *b, const float *u, float *r, void *) { return g ? residual_range<float>(c, g, k, b, u, r, 0, g->nz) : fail(MGK_EINVAL, "g"); }
int mgk_apply_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, const double *x, double *y, void *) {
    if (!c || !g || !coef || !x || !y || x == y) return fail(MGK_EINVAL, "mgk_apply_f64");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 7);
    return run(c, [=] { st_op<double>(M_APPLY, G, k.data(), 1, 1, 0, 0, 0, (const double *)nullptr, x, (const double *)nullptr, y, 0, NMARCH(&G)); });
}
int mgk_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, int mode, const double *ctab, const double *dtab, double scale, const double *b, const double *u, double *o, void *) {
    if (!c || !g || g->dim != 2 || !ctab || !u || !o || u == o || (mode != 4 && !b) || (mode == 0 && !dtab)) return fail(MGK_EINVAL, "mgk_rowcoef_f64");
    const mgk_geom G = *g;
    const int m = mode == 0 ? M_JACOBI : mode == 1 ? M_RESIDUAL : mode == 4 ? M_APPLY : -1;
    if (m < 0) return fail(MGK_EINVAL, "mgk_rowcoef_f64: mode");
    return run(c, [=] { st_op<double>(m, G, nullptr, 1.0, scale, 0, 0, 0, m == M_APPLY ? (const double *)nullptr : b, u, (const double *)nullptr, o, 0, G.ny, ctab, dtab); });
}
int mgk_cheby_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *dtab, double ckm1, double ck, double cz, const double *b, const double *pk, const double *pkm1, double *o, void *) {
    if (!c || !g || g->dim != 2 || !ctab || !dtab || !b || !pk || !pkm1 || !o || pk == o || pkm1 == o) return fail(MGK_EINVAL, "mgk_cheby_rowcoef_f64");
    const mgk_geom G = *g;
    return run(c, [=] { st_op<double>(M_CHEBY, G, nullptr, 1.0, 1, ckm1, ck, cz, b, pk, pkm1, o, 0, G.ny, ctab, dtab); });
}
int mgk_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *x, double *out, void *) {
    if (!c || !g || !x || !out) return fail(MGK_EINVAL, "mgk_sumsq_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    deliver(c, sumsq_field<double>(*g, x, 0, NMARCH(g)), out);
    return 0;
}
int mgk_residual_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, const double *b, const double *u, double *out, void *) {
    if (!c || !g || !coef || !b || !u || !out) return fail(MGK_EINVAL, "mgk_residual_sumsq_f64");
    std::vector<double> r(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, NMARCH(g));
    deliver(c, sumsq_field<double>(*g, r.data(), 0, NMARCH(g)), out);
    return 0;
}
int mgk_jacobi_sumsq_range_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *o,
                               int z0, int z1, int part_off, int *nparts, void *) {
    if (!c || !g || !coef || !b || !u || !o || u == o || !nparts || part_off < 0 || part_off >= (int)c->partials.size()) return fail(MGK_EINVAL, "mgk_jacobi_sumsq_range_f64");
    CHKRANGE(g, z0, z1, "mgk_jacobi_sumsq_range_f64: range");
    std::vector<double> r(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), z0, z1);
    st_op<double>(M_JACOBI, *g, coef, dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, o, z0, z1);
    c->partials[part_off] = sumsq_field<double>(*g, r.data(), z0, z1);
    *nparts = 1;
    return 0;
}
int mgk_partials_finish(mgk_ctx *c, int n, double *out, void *) {
    if (!c || n < 1 || n > (int)c->partials.size() || !out) return fail(MGK_EINVAL, "mgk_partials_finish");
    double s = 0; for (int q = 0; q < n; q++) s += c->partials[q];
    deliver(c, s, out);
    return 0;
}
int mgk_jacobi_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *o, double *out, void *s) {
    int n = 0;
    if (!g || !out) return fail(MGK_EINVAL, "mgk_jacobi_sumsq_f64");
    int rc = mgk_jacobi_sumsq_range_f64(c, g, coef, dinv, scale, b, u, o, 0, NMARCH(g), 0, &n, s);
    return rc ? rc : mgk_partials_finish(c, n, out, s);
}
int mgk_jacobi2_f64(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double s, const double *b, const double *u, double *o, void *) { return g ? jacobi2_api<double>(c, g, nullptr, k, d, s, b, u, o, nullptr, 0, 0, 0, g->nz) : fail(MGK_EINVAL, "g"); }
int mgk_jacobi2_f32(mgk_ctx *c, const mgk_geom *g, const double *k, double d, double s, const float *b, const float *u, float *o, void *) { return g ? jacobi2_api<float>(c, g, nullptr, k, d, s, b, u, o, nullptr, 0, 0, 0, g->nz) : fail(MGK_EINVAL, "g"); }
int mgk_jacobi2_slab_f64(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gf, const double *k, double d, double s, const double *b, const double *u, double *o, const double *far, int lo, int hi, int z0, int z1, void *) { return jacobi2_api<double>(c, g, gf, k, d, s, b, u, o, far, lo, hi, z0, z1); }
int mgk_jacobi2_slab_f32(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gf, const double *k, double d, double s, const float *b, const float *u, float *o, const float *far, int lo, int hi, int z0, int z1, void *) { return jacobi2_api<float>(c, g, gf, k, d, s, b, u, o, far, lo, hi, z0, z1); }
int mgk_jacobi2_2d_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *o, void *) {
    if (!c || !g || g->dim != 2 || !coef || !b || !u || !o || u == o) return fail(MGK_EINVAL, "mgk_jacobi2_2d_f64");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        std::vector<double> w(G.total, 0.0);
        st_op<double>(M_JACOBI, G, k.data(), dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, w.data(), 0, G.ny);
        st_op<double>(M_JACOBI, G, k.data(), dinv, scale, 0, 0, 0, b, w.data(), (const double *)nullptr, o, 0, G.ny);
    });
}
int mgk_restrict_fw_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *r, double *bc, void *) { return restrict_api<double>(c, gf, gc, r, bc); }
int mgk_restrict_fw_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const float *r, float *bc, void *) { return restrict_api<float>(c, gf, gc, r, bc); }
int mgk_dense_mult_f64(mgk_ctx *c, int m, int n, const double *B, const double *x, double *y, void *) {
    if (!c || m < 1 || n < 1 || !B || !x || !y || x == y) return fail(MGK_EINVAL, "mgk_dense_mult_f64");
    return run(c, [=] {
        for (int r = 0; r < m; r++) { double acc = 0.0; for (int j = 0; j < n; j++) acc += B[(long)r * n + j] * x[j]; y[r] = acc; }
    });
}
int mgk_apply_add_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, const double *x, double *y, void *) {
    if (!c || !g || !coef || !x || !y || x == y || g->dim != 2 || g->nx != g->ny) return fail(MGK_EINVAL, "mgk_apply_add_f64");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 5);
    return run(c, [=] {
        for (int i = 0; i < G.ny; i++) for (int j = 0; j < G.nx; j++) {
            const long o = G.org + (long)i * G.pitch + j;
            double z = y[o];
            z = z + k[0] * x[o - G.pitch]; z = z + k[1] * x[o - 1]; z = z + k[2] * x[o]; z = z + k[3] * x[o + 1]; z = z + k[4] * x[o + G.pitch];
            y[o] = z;
        }
    });
}
int mgk_window_add_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, int S, const double *wtab, const double *xc, double *yf, void *) {
    if (!c || !gf || !gc || !wtab || !xc || !yf) return fail(MGK_EINVAL, "mgk_window_add_f64: bad arguments");
    if (gf->dim != 2 || gc->dim != 2 || gf->nx != gf->ny || gc->nx != gc->ny || S < 2 || (S & (S - 1)) || (long)gf->nx + 1 != (long)S * (gc->nx + 1))
        return fail(MGK_EINVAL, "mgk_window_add_f64: grids");
    const mgk_geom F = *gf, Cg = *gc;
    return run(c, [=] {
        const int W = 2 * S - 1;
        const double *X = xc + Cg.org;
        for (int i = 0; i < F.ny; i++) for (int j = 0; j < F.nx; j++) {
            const int ni = ((i + 1) % S == 0) ? 1 : 2, ic0 = i / S - (ni - 1);
            const int nj = ((j + 1) % S == 0) ? 1 : 2, jc0 = j / S - (nj - 1);
            double y = yf[F.org + (long)i * F.pitch + j];
            for (int p = 0; p < ni; p++) for (int q = 0; q < nj; q++) {
                const int ic = ic0 + p, jc = jc0 + q;
                y = y + wtab[(i - S * ic) * W + (j - S * jc)] * X[(long)ic * Cg.pitch + jc];
            }
            yf[F.org + (long)i * F.pitch + j] = y;
        }
    });
}
int mgk_prolong_add_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *uc, double *uf, void *) { return prolong_add_api<double>(c, gf, gc, uc, uf); }
int mgk_prolong_add_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const float *uc, float *uf, void *) { return prolong_add_api<float>(c, gf, gc, uc, uf); }
int mgk_prolong_jacobi_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, double d, double s, const double *b, const double *uc, const double *u, double *o, void *) { return gf ? prolong_jacobi_api<double>(c, gf, gc, k, d, s, b, uc, u, o, 0, NMARCH(gf)) : fail(MGK_EINVAL, "g"); }
int mgk_prolong_jacobi_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, double d, double s, const float *b, const float *uc, const float *u, float *o, void *) { return gf ? prolong_jacobi_api<float>(c, gf, gc, k, d, s, b, uc, u, o, 0, gf->nz) : fail(MGK_EINVAL, "g"); }
int mgk_prolong_jacobi_range_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, double d, double s, const double *b, const double *uc, const double *u, double *o, int z0, int z1, void *) { return prolong_jacobi_api<double>(c, gf, gc, k, d, s, b, uc, u, o, z0, z1); }
int mgk_prolong_jacobi_range_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, double d, double s, const float *b, const float *uc, const float *u, float *o, int z0, int z1, void *) { return prolong_jacobi_api<float>(c, gf, gc, k, d, s, b, uc, u, o, z0, z1); }
int mgk_residual_restrict_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const double *b, const double *u, double *bc, void *) { return gc ? rr_api<double>(c, gf, gc, k, b, u, bc, nullptr, 0, 0, 0, gc->nz, nullptr) : fail(MGK_EINVAL, "g"); }
int mgk_residual_restrict_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const float *b, const float *u, float *bc, void *) { return gc ? rr_api<float>(c, gf, gc, k, b, u, bc, nullptr, 0, 0, 0, gc->nz, nullptr) : fail(MGK_EINVAL, "g"); }
int mgk_residual_restrict_range_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const double *b, const double *u, double *bc, int k0, int k1, void *) { return rr_api<double>(c, gf, gc, k, b, u, bc, nullptr, 0, 0, k0, k1, nullptr); }
int mgk_residual_restrict_range_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const float *b, const float *u, float *bc, int k0, int k1, void *) { return rr_api<float>(c, gf, gc, k, b, u, bc, nullptr, 0, 0, k0, k1, nullptr); }
int mgk_residual_restrict_jz_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const double *b, const double *u, double *bc, double *uc0, double d, double s, void *) { return (gc && uc0) ? rr_api<double>(c, gf, gc, k, b, u, bc, uc0, d, s, 0, gc->nz, nullptr) : fail(MGK_EINVAL, "jz"); }
int mgk_residual_restrict_jz_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *k, const float *b, const float *u, float *bc, float *uc0, double d, double s, void *) { return (gc && uc0) ? rr_api<float>(c, gf, gc, k, b, u, bc, uc0, d, s, 0, gc->nz, nullptr) : fail(MGK_EINVAL, "jz"); }
static bool far_ok(const mgk_geom *gf, const mgk_geom *gfar) { return gfar && gfar->dim == 3 && gfar->nz == 2 && gfar->nx == gf->nx && gfar->ny == gf->ny && gfar->pitch == gf->pitch; }
int mgk_residual_restrict_slab_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *k, const double *b, const double *u, const double *far, int hi, double *bc, int k0, int k1, void *) {
    if (hi && (!gf || !gc || !far || !far_ok(gf, gfar) || gf->nz != 2 * gc->nz)) return fail(MGK_EINVAL, "mgk_residual_restrict_slab_f64: far field");
    return rr_api<double>(c, gf, gc, k, b, u, bc, nullptr, 0, 0, k0, k1, hi ? far + 3 * gfar->plane : nullptr);
}
int mgk_residual_restrict_slab_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *k, const float *b, const float *u, const float *far, int hi, float *bc, int k0, int k1, void *) {
    if (hi && (!gf || !gc || !far || !far_ok(gf, gfar) || gf->nz != 2 * gc->nz)) return fail(MGK_EINVAL, "mgk_residual_restrict_slab_f32: far field");
    return rr_api<float>(c, gf, gc, k, b, u, bc, nullptr, 0, 0, k0, k1, hi ? far + 3 * gfar->plane : nullptr);
}
}   // extern "C"
template <class T> static int finish_api(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const T *r, T *bc) {
    if (!c || !r || !bc || !xfer_ok(gf, gc) || gf->dim != 3 || gf->nz != 2 * gc->nz) return fail(MGK_EINVAL, "mgk_restrict_finish");
    const mgk_geom F = *gf, Cg = *gc;
    return run(c, [=] { restrict_fw<T>(F, Cg, r, bc, Cg.nz - 1, Cg.nz, 3, true); });
}
extern "C" {
int mgk_restrict_finish_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *r, double *bc, void *) { return finish_api<double>(c, gf, gc, r, bc); }
int mgk_restrict_finish_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const float *r, float *bc, void *) { return finish_api<float>(c, gf, gc, r, bc); }
int mgk_residual_restrict_2d_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const double *b, const double *u, double *bc, double *uc0, double dinv_c, double scale_c, void *) {
    if (!c || !coef || !b || !u || !bc || !xfer_ok(gf, gc) || gf->dim != 2) return fail(MGK_EINVAL, "mgk_residual_restrict_2d_f64");
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        std::vector<double> r(F.total, 0.0);
        st_op<double>(M_RESIDUAL, F, k.data(), 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, F.ny);
        restrict_fw<double>(F, Cg, r.data(), bc, 0, 1);
        if (uc0) for (int i = 0; i < Cg.ny; i++) for (int j = 0; j < Cg.nx; j++) { const double zq = at(bc, Cg, 0, i, j) * dinv_c; at(uc0, Cg, 0, i, j) = scale_c * zq; }
    });
}
}   // extern "C"
template <class T> static int tail_api(mgk_ctx *c, const mgk_geom *g0, int nlev, const int *n, const double *coef7, const double *dinv, double scale_, int v0, int v1, const T *b, T *u,
                                       const double *const *ctab = nullptr, const double *const *dtab = nullptr, const double *cscale_ = nullptr) {
    const double cscale = cscale_ ? *cscale_ : scale_;
    if (!c || !g0 || !n || (!coef7 && !ctab) || (!dinv && !dtab) || !b || !u || nlev < 1 || nlev > 8 || n[0] != g0->nx || n[0] > mgk_tail_max_n(g0->dim)) return fail(MGK_EINVAL, "mgk_tail_cycle");
    if (ctab && (!dtab || g0->dim != 2 || sizeof(T) != 8)) return fail(MGK_EINVAL, "mgk_tail_cycle: tables are 2-D fp64");
    for (int l = 1; l < nlev; l++) if (n[l - 1] != 2 * n[l] + 1) return fail(MGK_EINVAL, "mgk_tail_cycle: hierarchy");
    const mgk_geom G0 = *g0; std::vector<int> nn(n, n + nlev);
    std::vector<double> k7(7 * nlev, 0.0), di(nlev, 1.0);
    if (coef7) k7.assign(coef7, coef7 + 7 * nlev);
    if (dinv) di.assign(dinv, dinv + nlev);
    std::vector<const double *> ct(nlev, nullptr), dt(nlev, nullptr);
    for (int l = 0; l < nlev && ctab; l++) { ct[l] = ctab[l]; dt[l] = dtab[l]; if (!ct[l] || !dt[l]) return fail(MGK_EINVAL, "mgk_tail_cycle: null table"); }
    return run(c, [=] {
        std::vector<mgk_geom> G(nlev);
        std::vector<std::vector<T>> U(nlev), W(nlev), B(nlev);
        for (int l = 0; l < nlev; l++) {
            if (sizeof(T) == 8) mgk_geom_init(&G[l], G0.dim, nn[l], nn[l], nn[l]); else mgk_geom_init_f32(&G[l], 3, nn[l], nn[l], nn[l]);
            U[l].assign(G[l].total, (T)0); W[l].assign(G[l].total, (T)0); B[l].assign(G[l].total, (T)0);
        }
        memcpy(B[0].data(), b, sizeof(T) * (size_t)G0.total);
        auto smooth = [&](int l, int sweeps, bool zero) {
            const double scale = (l == nlev - 1) ? cscale : scale_;        // (mgk_tail_cycle_cs_f64: another damping factor on the coarsest level)
            for (int it = 0; it < sweeps; it++) {
                if (it == 0 && zero) { for (long q = 0; q < G[l].total; q++) W[l][q] = (T)0;
                    for (int k = 0; k < G[l].nz; k++) for (int i = 0; i < G[l].ny; i++) for (int j = 0; j < G[l].nx; j++) { const T zx = at(B[l].data(), G[l], k, i, j) * (T)(dt[l] ? dt[l][i] : di[l]); at(W[l].data(), G[l], k, i, j) = (T)scale * zx; } }
                else st_op<T>(M_JACOBI, G[l], &k7[7 * l], di[l], scale, 0, 0, 0, B[l].data(), U[l].data(), (const T *)nullptr, W[l].data(), 0, NMARCH(&G[l]), ct[l], dt[l]);
                U[l].swap(W[l]);
            }
        };
        smooth(0, nlev == 1 ? v1 : v0, true);
        for (int l = 1; l < nlev; l++) {
            st_op<T>(M_RESIDUAL, G[l - 1], &k7[7 * (l - 1)], 1, 1, 0, 0, 0, B[l - 1].data(), U[l - 1].data(), (const T *)nullptr, W[l - 1].data(), 0, NMARCH(&G[l - 1]), ct[l - 1], dt[l - 1]);
            restrict_fw<T>(G[l - 1], G[l], W[l - 1].data(), B[l].data(), 0, G[l].dim == 3 ? G[l].nz : 1);
            std::fill(U[l].begin(), U[l].end(), (T)0);
            smooth(l, l == nlev - 1 ? v1 : v0, true);
        }
        for (int l = nlev - 2; l >= 0; l--) {
            for (int k = 0; k < (G[l].dim == 3 ? G[l].nz : 1); k++) for (int i = 0; i < G[l].ny; i++) for (int j = 0; j < G[l].nx; j++)
                at(U[l].data(), G[l], k, i, j) = at(U[l].data(), G[l], k, i, j) + prolong_at(G[l], G[l + 1], U[l + 1].data(), G[l].dim == 3 ? k : 1, i, j);
            smooth(l, v0, false);
        }
        for (int k = 0; k < (G0.dim == 3 ? G0.nz : 1); k++) for (int i = 0; i < G0.ny; i++) memcpy(&at(u, G0, k, i, 0), &at(U[0].data(), G[0], k, i, 0), sizeof(T) * (size_t)G0.nx);
    });
}
extern "C" {
int mgk_tail_cycle_f64(mgk_ctx *c, const mgk_geom *g0, int nl, const int *n, const double *k7, const double *di, double s, int v0, int v1, const double *b, double *u, void *) { return tail_api<double>(c, g0, nl, n, k7, di, s, v0, v1, b, u); }
int mgk_tail_cycle_f32(mgk_ctx *c, const mgk_geom *g0, int nl, const int *n, const double *k7, const double *di, double s, int v0, int v1, const float *b, float *u, void *) { return tail_api<float>(c, g0, nl, n, k7, di, s, v0, v1, b, u); }

}   // extern "C"
// three sweeps from a zero guess in one pass
template <class T> static int j2zero_api(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const T *b, T *o) {
    if (!c || !g || g->dim != 3 || !coef || !b || !o || b == o) return fail(MGK_EINVAL, "mgk_jacobi2_zero");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        std::vector<T> w1(G.total, (T)0), w2(G.total, (T)0);
        for (int kk = 0; kk < G.nz; kk++) for (int i = 0; i < G.ny; i++) for (int j = 0; j < G.nx; j++) { const T zx = at(b, G, kk, i, j) * (T)dinv; at(w1.data(), G, kk, i, j) = (T)scale * zx; }
        st_op<T>(M_JACOBI, G, k.data(), dinv, scale, 0, 0, 0, b, w1.data(), (const T *)nullptr, w2.data(), 0, G.nz);
        st_op<T>(M_JACOBI, G, k.data(), dinv, scale, 0, 0, 0, b, w2.data(), (const T *)nullptr, w1.data(), 0, G.nz);
        for (int kk = 0; kk < G.nz; kk++) for (int i = 0; i < G.ny; i++) memcpy(&at(o, G, kk, i, 0), &at(w1.data(), G, kk, i, 0), sizeof(T) * (size_t)G.nx);
    });
}
extern "C" {
int mgk_jacobi2_zero_ok_f64(const mgk_geom *g) { return (g && g->dim == 3 && g->nx >= 7) ? 1 : 0; }
int mgk_jacobi2_zero_ok_f32(const mgk_geom *g) { return (g && g->dim == 3 && g->nx >= 7) ? 1 : 0; }
int mgk_jacobi2_zero_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, double *o, void *) { return j2zero_api<double>(c, g, coef, dinv, scale, b, o); }
int mgk_jacobi2_zero_f32(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const float *b, float *o, void *) { return j2zero_api<float>(c, g, coef, dinv, scale, b, o); }
// two sweeps + the norm of the input field's residual; the last pre-smoothing sweep fused with residual + restriction
int mgk_jacobi2_sumsq_ok_f64(const mgk_geom *g) { return (g && g->dim == 3 && g->nx >= 7) ? 1 : 0; }     // (the mock takes any 3-D shape: host logic only)
int mgk_jacobi2_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *o, double *out, void *) {
    if (!c || !g || g->dim != 3 || !coef || !b || !u || !o || u == o || !out) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    std::vector<double> r(g->total, 0.0), w(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, g->nz);
    st_op<double>(M_JACOBI, *g, coef, dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, w.data(), 0, g->nz);
    st_op<double>(M_JACOBI, *g, coef, dinv, scale, 0, 0, 0, b, w.data(), (const double *)nullptr, o, 0, g->nz);
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->nz), out);
    return 0;
}
int mgk_sweep_residual_restrict_ok_f64(const mgk_geom *gf, const mgk_geom *gc) { return (xfer_ok(gf, gc) && gf->dim == 3 && gf->nz == 2 * gc->nz + 1 && gf->nx >= 7) ? 1 : 0; }
int mgk_sweep_residual_restrict_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale, const double *b, const double *u, double *o,
                                    double *bc, double *uc0, double dinv_c, double scale_c, void *) {
    if (!c || !coef || !b || !u || !o || u == o || !bc || !mgk_sweep_residual_restrict_ok_f64(gf, gc)) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_f64");
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        st_op<double>(M_JACOBI, F, k.data(), dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, o, 0, F.nz);
        std::vector<double> r(F.total, 0.0);
        st_op<double>(M_RESIDUAL, F, k.data(), 1, 1, 0, 0, 0, b, o, (const double *)nullptr, r.data(), 0, F.nz);
        restrict_fw<double>(F, Cg, r.data(), bc, 0, Cg.nz);
        if (uc0) for (int kc = 0; kc < Cg.nz; kc++) for (int i = 0; i < Cg.ny; i++) for (int j = 0; j < Cg.nx; j++) {
            const double zq = at(bc, Cg, kc, i, j) * dinv_c; at(uc0, Cg, kc, i, j) = scale_c * zq;
        }
    });
}

// 2-D forms
int mgk_jacobi2_2d_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *o, double *out, void *) {
    if (!c || !g || g->dim != 2 || !coef || !b || !u || !o || u == o || !out) return fail(MGK_EINVAL, "mgk_jacobi2_2d_sumsq_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    std::vector<double> r(g->total, 0.0), w(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, g->ny);
    st_op<double>(M_JACOBI, *g, coef, dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, w.data(), 0, g->ny);
    st_op<double>(M_JACOBI, *g, coef, dinv, scale, 0, 0, 0, b, w.data(), (const double *)nullptr, o, 0, g->ny);
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->ny), out);
    return 0;
}
int mgk_sweep_residual_restrict_2d_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale, const double *b, const double *u, double *o,
                                       double *bc, double *uc0, double dinv_c, double scale_c, void *) {
    if (!c || !coef || !b || !u || !o || u == o || !bc || !xfer_ok(gf, gc) || gf->dim != 2) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_2d_f64");
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        st_op<double>(M_JACOBI, F, k.data(), dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, o, 0, F.ny);
        std::vector<double> r(F.total, 0.0);
        st_op<double>(M_RESIDUAL, F, k.data(), 1, 1, 0, 0, 0, b, o, (const double *)nullptr, r.data(), 0, F.ny);
        restrict_fw<double>(F, Cg, r.data(), bc, 0, 1);
        if (uc0) for (int i = 0; i < Cg.ny; i++) for (int j = 0; j < Cg.nx; j++) { const double zq = at(bc, Cg, 0, i, j) * dinv_c; at(uc0, Cg, 0, i, j) = scale_c * zq; }
    });
}
int mgk_jacobi2_2d_sumsq_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *dtab, double scale, const double *b, const double *u, double *o, double *out, void *) {
    if (!c || !g || g->dim != 2 || !ctab || !dtab || !b || !u || !o || u == o || !out) return fail(MGK_EINVAL, "mgk_jacobi2_2d_sumsq_rowcoef_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    std::vector<double> r(g->total, 0.0), w(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, nullptr, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, g->ny, ctab, dtab);
    st_op<double>(M_JACOBI, *g, nullptr, 1.0, scale, 0, 0, 0, b, u, (const double *)nullptr, w.data(), 0, g->ny, ctab, dtab);
    st_op<double>(M_JACOBI, *g, nullptr, 1.0, scale, 0, 0, 0, b, w.data(), (const double *)nullptr, o, 0, g->ny, ctab, dtab);
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->ny), out);
    return 0;
}
int mgk_sweep_residual_restrict_2d_rowcoef_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *ctab, const double *dtab, double scale, const double *b, const double *u,
                                               double *o, double *bc, double *uc0, const double *dtab_c, double scale_c, void *) {
    if (!c || !ctab || !dtab || !b || !u || !o || u == o || !bc || !xfer_ok(gf, gc) || gf->dim != 2 || (uc0 && !dtab_c)) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_2d_rowcoef_f64");
    const mgk_geom F = *gf, Cg = *gc;
    return run(c, [=] {
        st_op<double>(M_JACOBI, F, nullptr, 1.0, scale, 0, 0, 0, b, u, (const double *)nullptr, o, 0, F.ny, ctab, dtab);
        std::vector<double> r(F.total, 0.0);
        st_op<double>(M_RESIDUAL, F, nullptr, 1, 1, 0, 0, 0, b, o, (const double *)nullptr, r.data(), 0, F.ny, ctab, (const double *)nullptr);
        restrict_fw<double>(F, Cg, r.data(), bc, 0, 1);
        if (uc0) for (int i = 0; i < Cg.ny; i++) for (int j = 0; j < Cg.nx; j++) { const double zq = at(bc, Cg, 0, i, j) * dtab_c[i]; at(uc0, Cg, 0, i, j) = scale_c * zq; }
    });
}
// (fuse bit 12) prolongation + two sweeps in one pass; two sweeps + the norm of the residual of the first one's output
static int g_calls_pj2 = 0, g_calls_mid = 0;
int mgk_prolong_jacobi2_ok_f64(const mgk_geom *gf, const mgk_geom *gc) { return (xfer_ok(gf, gc) && gf->dim == 3 && gf->nz == 2 * gc->nz + 1 && gf->nx >= 7) ? 1 : 0; }
int mgk_prolong_jacobi2_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale, const double *b, const double *uc, const double *u, double *o, void *) {
    if (!c || !coef || !b || !uc || !u || !o || u == o || !mgk_prolong_jacobi2_ok_f64(gf, gc)) return fail(MGK_EINVAL, "mgk_prolong_jacobi2_f64");
    __atomic_fetch_add(&g_calls_pj2, 1, __ATOMIC_RELAXED);
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        std::vector<double> t = corrected<double>(F, Cg, uc, u), w(F.total, 0.0);
        st_op<double>(M_JACOBI, F, k.data(), dinv, scale, 0, 0, 0, b, t.data(), (const double *)nullptr, w.data(), 0, F.nz);
        st_op<double>(M_JACOBI, F, k.data(), dinv, scale, 0, 0, 0, b, w.data(), (const double *)nullptr, o, 0, F.nz);
    });
}
int mgk_jacobi2_sumsq_mid_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *o, double *out, void *) {
    if (!c || !g || g->dim != 3 || !coef || !b || !u || !o || u == o || !out) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_mid_f64");
    __atomic_fetch_add(&g_calls_mid, 1, __ATOMIC_RELAXED);
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    std::vector<double> r(g->total, 0.0), w(g->total, 0.0);
    st_op<double>(M_JACOBI, *g, coef, dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, w.data(), 0, g->nz);
    st_op<double>(M_RESIDUAL, *g, coef, 1, 1, 0, 0, 0, b, w.data(), (const double *)nullptr, r.data(), 0, g->nz);
    st_op<double>(M_JACOBI, *g, coef, dinv, scale, 0, 0, 0, b, w.data(), (const double *)nullptr, o, 0, g->nz);
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->nz), out);
    return 0;
}
// (round 3, second session) the same two on a z-slab
static int g_calls_pj2_slab = 0, g_calls_mid_slab = 0;
int mgk_prolong_jacobi2_slab_ok_f64(const mgk_geom *gf, const mgk_geom *gc, int has_hi) {
    return (gf && gc && gf->dim == 3 && gc->dim == 3 && gf->nx == 2 * gc->nx + 1 && gf->ny == 2 * gc->ny + 1 && gf->nz == (has_hi ? 2 * gc->nz : 2 * gc->nz + 1) &&
            gf->nx >= 7 && gf->nz >= 4 && gc->nz >= 2) ? 1 : 0;
}
int mgk_prolong_jacobi2_slab_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const mgk_geom *gcfar, const double *coef, double dinv, double scale,
                                 const double *b, const double *uc, const double *u, double *o, const double *far, const double *cfar, int has_lo, int has_hi, int z0, int z1, void *) {
    if (!c || !coef || !b || !uc || !u || !o || u == o || !mgk_prolong_jacobi2_slab_ok_f64(gf, gc, has_hi)) return fail(MGK_EINVAL, "mgk_prolong_jacobi2_slab_f64");
    if ((has_lo || has_hi) && (!far || !far_ok(gf, gfar))) return fail(MGK_EINVAL, "mgk_prolong_jacobi2_slab_f64: far field");
    if (has_lo && (!cfar || !far_ok(gc, gcfar))) return fail(MGK_EINVAL, "mgk_prolong_jacobi2_slab_f64: coarse far field");
    if (z0 < 0 || (z0 & 1) || z1 > gf->nz || z0 >= z1) return fail(MGK_EINVAL, "mgk_prolong_jacobi2_slab_f64: range");
    __atomic_fetch_add(&g_calls_pj2_slab, 1, __ATOMIC_RELAXED);
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        // extended slab: fine planes -2 .. nz+1 as interior planes 0 .. nz+3 of E (shift 2: parities kept), coarse planes -2 .. nzc as planes -1 .. nzc+1 of Ec
        // (shift 1: fine plane 2 c + 1 stays centred on coarse plane c)
        mgk_geom E, Ec; geom3<double>(&E, F.nx, F.ny, F.nz + 4); geom3<double>(&Ec, Cg.nx, Cg.ny, Cg.nz + 1);
        std::vector<double> ue(E.total, 0.0), be(E.total, 0.0), ce(Ec.total, 0.0), w(E.total, 0.0), o2(E.total, 0.0);
        for (int kk = -1; kk <= F.nz; kk++) memcpy(&ue[(size_t)((kk + 3) * E.plane)], u + (size_t)((kk + 1) * F.plane), sizeof(double) * (size_t)F.plane);
        if (has_lo) memcpy(&ue[(size_t)E.plane], far, sizeof(double) * (size_t)F.plane);                                            // far's lo ghost plane = plane -2
        if (has_hi) memcpy(&ue[(size_t)((F.nz + 4) * E.plane)], far + (size_t)(3 * F.plane), sizeof(double) * (size_t)F.plane);     // far's hi ghost plane = plane nz+1
        for (int kk = (has_lo ? -1 : 0); kk <= (has_hi ? F.nz : F.nz - 1); kk++)
            memcpy(&be[(size_t)((kk + 3) * E.plane)], b + (size_t)((kk + 1) * F.plane), sizeof(double) * (size_t)F.plane);
        for (int kc = -1; kc <= Cg.nz; kc++) memcpy(&ce[(size_t)((kc + 2) * Ec.plane)], uc + (size_t)((kc + 1) * Cg.plane), sizeof(double) * (size_t)Cg.plane);
        if (has_lo) memcpy(&ce[0], cfar, sizeof(double) * (size_t)Cg.plane);                                                        // coarse plane -2 = Ec's lo ghost plane
        const int p0 = has_lo ? 0 : 1, p1 = has_hi ? F.nz + 4 : F.nz + 3;      // planes of E that are real (fine -2 / -1 .. nz / nz+1)
        for (int e = p0; e < p1; e++)
            for (int i = -1; i <= E.ny; i++)
                for (int j = 0; j < E.nx; j++) at(ue.data(), E, e, i, j) = at(ue.data(), E, e, i, j) + prolong_at(E, Ec, ce.data(), e, i, j);
        st_op<double>(M_JACOBI, E, k.data(), dinv, scale, 0, 0, 0, be.data(), ue.data(), (const double *)nullptr, w.data(), has_lo ? 1 : 2, has_hi ? F.nz + 3 : F.nz + 2);
        st_op<double>(M_JACOBI, E, k.data(), dinv, scale, 0, 0, 0, be.data(), w.data(), (const double *)nullptr, o2.data(), z0 + 2, z1 + 2);
        for (int kk = z0; kk < z1; kk++) for (int i = 0; i < F.ny; i++) memcpy(&at(o, F, kk, i, 0), &at(o2.data(), E, kk + 2, i, 0), sizeof(double) * (size_t)F.nx);
    });
}
int mgk_jacobi2_sumsq_mid_slab_f64(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gf, const double *coef, double dinv, double scale, const double *b, const double *u, double *o,
                                   const double *far, int lo, int hi, int z0, int z1, int part_off, int *nparts, void *) {
    if (!c || !g || g->dim != 3 || !coef || !b || !u || !o || u == o || !nparts || part_off < 0 || part_off >= (int)c->partials.size()) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_mid_slab_f64");
    if (z0 < 0 || z1 > g->nz || z0 >= z1) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_mid_slab_f64: range");
    __atomic_fetch_add(&g_calls_mid_slab, 1, __ATOMIC_RELAXED);
    int rc = jacobi2_api<double>(c, g, gf, coef, dinv, scale, b, u, o, far, lo, hi, z0, z1);
    if (rc) return rc;
    // the residual of the FIRST sweep's output on the planes [z0, z1): that sweep on the planes z0-1 .. z1 (u's ghost planes are valid), then b - A w
    mgk_geom E; geom3<double>(&E, g->nx, g->ny, g->nz + 2);
    std::vector<double> ue(E.total, 0.0), be(E.total, 0.0), w(E.total, 0.0), r(E.total, 0.0);
    for (int kk = -1; kk <= g->nz; kk++) memcpy(&ue[(size_t)((kk + 2) * E.plane)], u + (size_t)((kk + 1) * g->plane), sizeof(double) * (size_t)g->plane);
    if (lo) memcpy(&ue[0], far, sizeof(double) * (size_t)g->plane);
    if (hi) memcpy(&ue[(size_t)((g->nz + 3) * E.plane)], far + (size_t)(3 * g->plane), sizeof(double) * (size_t)g->plane);
    for (int kk = (lo ? -1 : 0); kk <= (hi ? g->nz : g->nz - 1); kk++) memcpy(&be[(size_t)((kk + 2) * E.plane)], b + (size_t)((kk + 1) * g->plane), sizeof(double) * (size_t)g->plane);
    st_op<double>(M_JACOBI, E, coef, dinv, scale, 0, 0, 0, be.data(), ue.data(), (const double *)nullptr, w.data(), lo ? 0 : 1, hi ? g->nz + 2 : g->nz + 1);
    st_op<double>(M_RESIDUAL, E, coef, 1, 1, 0, 0, 0, be.data(), w.data(), (const double *)nullptr, r.data(), z0 + 1, z1 + 1);
    c->partials[part_off] = sumsq_field<double>(E, r.data(), z0 + 1, z1 + 1);
    *nparts = 1;
    return 0;
}
// the two on a z-slab (far planes as the halo exchange delivers them; see include/mgk.h)
static int g_calls_j2n_slab = 0, g_calls_srr_slab = 0, g_calls_srr = 0, g_calls_j2n = 0;
static void print_stats() { if (getenv("MOCK_MGK_STATS")) fprintf(stderr, "MOCK_MGK_STATS j2n=%d srr=%d j2n_slab=%d srr_slab=%d pj2=%d mid=%d pj2_slab=%d mid_slab=%d\n", g_calls_j2n, g_calls_srr, g_calls_j2n_slab, g_calls_srr_slab, g_calls_pj2, g_calls_mid, g_calls_pj2_slab, g_calls_mid_slab); }
static struct StatsAtExit { StatsAtExit() { atexit(print_stats); } } g_stats_at_exit;
int mgk_jacobi2_sumsq_slab_f64(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gf, const double *coef, double dinv, double scale, const double *b, const double *u, double *o,
                               const double *far, int lo, int hi, int z0, int z1, int part_off, int *nparts, void *) {
    if (!c || !g || g->dim != 3 || !coef || !b || !u || !o || u == o || !nparts || part_off < 0 || part_off >= (int)c->partials.size()) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_slab_f64");
    if (z0 < 0 || z1 > g->nz || z0 >= z1) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_slab_f64: range");
    __atomic_fetch_add(&g_calls_j2n_slab, 1, __ATOMIC_RELAXED);
    std::vector<double> r(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), z0, z1);      // u's ghost planes are valid
    c->partials[part_off] = sumsq_field<double>(*g, r.data(), z0, z1);
    *nparts = 1;
    return jacobi2_api<double>(c, g, gf, coef, dinv, scale, b, u, o, far, lo, hi, z0, z1);
}
int mgk_sweep_residual_restrict_slab_ok_f64(const mgk_geom *gf, const mgk_geom *gc) {
    return (gf && gc && gf->dim == 3 && gc->dim == 3 && gf->nx == 2 * gc->nx + 1 && gf->ny == 2 * gc->ny + 1 && (gf->nz == 2 * gc->nz || gf->nz == 2 * gc->nz + 1) &&
            gf->nx >= 7 && gf->nz >= 4 && gc->nz >= 1) ? 1 : 0;
}
int mgk_sweep_residual_restrict_slab_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                                         const double *b, const double *u, double *o, const double *far, const double *far2, const double *bfar, int has_lo, int has_hi,
                                         double *bc, int k0, int k1, void *) {
    if (!c || !coef || !b || !u || !o || u == o || !bc || !mgk_sweep_residual_restrict_slab_ok_f64(gf, gc)) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64");
    if ((has_lo || has_hi) && (!far || !far_ok(gf, gfar))) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64: far field");
    if (has_hi && (!far2 || !bfar || gf->nz != 2 * gc->nz)) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64: inner slab");
    if (!has_hi && gf->nz != 2 * gc->nz + 1) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64: last slab");
    if (k0 < 0 || k1 > gc->nz || k0 >= k1) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64: coarse range");
    __atomic_fetch_add(&g_calls_srr_slab, 1, __ATOMIC_RELAXED);
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        // extended slab: interior planes = planes -2 .. nz+1 of the slab (index + 2), hi ghost = plane nz+2
        mgk_geom E; geom3<double>(&E, F.nx, F.ny, F.nz + 4);
        std::vector<double> ue(E.total, 0.0), be(E.total, 0.0), we(E.total, 0.0), re(E.total, 0.0);
        const size_t pl = (size_t)F.plane;
        auto plane_e = [&](std::vector<double> &v, int p) { return &v[(size_t)(p + 3) * pl]; };      // plane p of the slab inside an extended field
        for (int p = -1; p <= F.nz; p++) { memcpy(plane_e(ue, p), u + (size_t)(p + 1) * pl, sizeof(double) * pl); memcpy(plane_e(be, p), b + (size_t)(p + 1) * pl, sizeof(double) * pl); }
        if (has_lo) memcpy(plane_e(ue, -2), far, sizeof(double) * pl);
        if (has_hi) { memcpy(plane_e(ue, F.nz + 1), far + 3 * pl, sizeof(double) * pl); memcpy(plane_e(ue, F.nz + 2), far2 + 3 * pl, sizeof(double) * pl);
                      memcpy(plane_e(be, F.nz + 1), bfar + 3 * pl, sizeof(double) * pl); }
        const int smin = has_lo ? -1 : 0, smax = has_hi ? F.nz + 1 : F.nz - 1, rmax = has_hi ? F.nz : F.nz - 1;
        st_op<double>(M_JACOBI, E, k.data(), dinv, scale, 0, 0, 0, be.data(), ue.data(), (const double *)nullptr, we.data(), smin + 2, smax + 3);
        st_op<double>(M_RESIDUAL, E, k.data(), 1, 1, 0, 0, 0, be.data(), we.data(), (const double *)nullptr, re.data(), 2, rmax + 3);
        restrict_fw<double>(E, Cg, re.data() + 2 * pl, bc, k0, k1);                                   // fine plane 0 of the slab = interior plane 2 of the extended field
        const int zs0 = 2 * k0, zs1 = (k1 == Cg.nz) ? F.nz : 2 * k1;
        for (int kk = zs0; kk < zs1; kk++) for (int i = 0; i < F.ny; i++) memcpy(&at(o, F, kk, i, 0), &at(we.data(), E, kk + 2, i, 0), sizeof(double) * (size_t)F.nx);
    });
}

// the fused forms on a row-table operator (2-D stretched meshes)
int mgk_tail_cycle_cs_f64(mgk_ctx *c, const mgk_geom *g0, int nl, const int *n, const double *k7, const double *di, const double *const *ctab, const double *const *dtab, double s, double cs,
                          int v0, int v1, const double *b, double *u, void *) {
    if ((ctab == nullptr) != (dtab == nullptr)) return fail(MGK_EINVAL, "mgk_tail_cycle_cs_f64");
    return tail_api<double>(c, g0, nl, n, ctab ? nullptr : k7, ctab ? nullptr : di, s, v0, v1, b, u, ctab, dtab, &cs);
}
int mgk_tail_cycle_rowcoef_f64(mgk_ctx *c, const mgk_geom *g0, int nl, const int *n, const double *const *ctab, const double *const *dtab, double s, int v0, int v1, const double *b, double *u, void *) {
    if (!ctab || !dtab) return fail(MGK_EINVAL, "mgk_tail_cycle_rowcoef_f64");
    return tail_api<double>(c, g0, nl, n, nullptr, nullptr, s, v0, v1, b, u, ctab, dtab);
}
int mgk_jacobi2_2d_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *dtab, double scale, const double *b, const double *u, double *o, void *) {
    if (!c || !g || g->dim != 2 || !ctab || !dtab || !b || !u || !o || u == o) return fail(MGK_EINVAL, "mgk_jacobi2_2d_rowcoef_f64");
    const mgk_geom G = *g;
    return run(c, [=] {
        std::vector<double> w(G.total, 0.0);
        st_op<double>(M_JACOBI, G, nullptr, 1.0, scale, 0, 0, 0, b, u, (const double *)nullptr, w.data(), 0, G.ny, ctab, dtab);
        st_op<double>(M_JACOBI, G, nullptr, 1.0, scale, 0, 0, 0, b, w.data(), (const double *)nullptr, o, 0, G.ny, ctab, dtab);
    });
}
int mgk_residual_sumsq_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *b, const double *u, double *out, void *) {
    if (!c || !g || g->dim != 2 || !ctab || !b || !u || !out) return fail(MGK_EINVAL, "mgk_residual_sumsq_rowcoef_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    std::vector<double> r(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, nullptr, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, g->ny, ctab, (const double *)nullptr);
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->ny), out);
    return 0;
}
int mgk_jacobi_sumsq_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *dtab, double scale, const double *b, const double *u, double *o, double *out, void *) {
    if (!c || !g || g->dim != 2 || !ctab || !dtab || !b || !u || !o || u == o || !out) return fail(MGK_EINVAL, "mgk_jacobi_sumsq_rowcoef_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    std::vector<double> r(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, nullptr, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, g->ny, ctab, dtab);
    st_op<double>(M_JACOBI, *g, nullptr, 1.0, scale, 0, 0, 0, b, u, (const double *)nullptr, o, 0, g->ny, ctab, dtab);
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->ny), out);
    return 0;
}
int mgk_jacobi_sumsq_store_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                               const double *b, const double *u, double *o, double *r, double *out, void *) {
    if (!c || !g || g->dim != 2 || (!coef && !ctab) || (ctab && !dtab) || !b || !u || !o || !r || u == o || u == r || b == r || r == o || b == o || !out)
        return fail(MGK_EINVAL, "mgk_jacobi_sumsq_store_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    st_op<double>(M_RESIDUAL, *g, ctab ? nullptr : coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r, 0, g->ny, ctab, dtab);
    st_op<double>(M_JACOBI, *g, ctab ? nullptr : coef, ctab ? 1.0 : dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, o, 0, g->ny, ctab, dtab);
    deliver(c, sumsq_field<double>(*g, r, 0, g->ny), out);
    return 0;
}
int mgk_prolong_jacobi_rowcoef_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *ctab, const double *dtab, double scale, const double *b, const double *uc, const double *u, double *o, void *) {
    if (!c || !ctab || !dtab || !b || !uc || !u || !o || u == o || !xfer_ok(gf, gc) || gf->dim != 2) return fail(MGK_EINVAL, "mgk_prolong_jacobi_rowcoef_f64");
    const mgk_geom F = *gf, Cg = *gc;
    return run(c, [=] { std::vector<double> t = corrected<double>(F, Cg, uc, u); st_op<double>(M_JACOBI, F, nullptr, 1.0, scale, 0, 0, 0, b, t.data(), (const double *)nullptr, o, 0, F.ny, ctab, dtab); });
}
int mgk_residual_restrict_2d_rowcoef_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *ctab, const double *b, const double *u, double *bc, double *uc0, const double *dtab_c, double scale_c, void *) {
    if (!c || !ctab || !b || !u || !bc || !xfer_ok(gf, gc) || gf->dim != 2 || (uc0 && !dtab_c)) return fail(MGK_EINVAL, "mgk_residual_restrict_2d_rowcoef_f64");
    const mgk_geom F = *gf, Cg = *gc;
    return run(c, [=] {
        std::vector<double> r(F.total, 0.0);
        st_op<double>(M_RESIDUAL, F, nullptr, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, F.ny, ctab, (const double *)nullptr);
        restrict_fw<double>(F, Cg, r.data(), bc, 0, 1);
        if (uc0) for (int i = 0; i < Cg.ny; i++) for (int j = 0; j < Cg.nx; j++) { const double zq = at(bc, Cg, 0, i, j) * dtab_c[i]; at(uc0, Cg, 0, i, j) = scale_c * zq; }
    });
}

// round 3: primitives of the peer transport, in host memory: a "handle" is the pointer itself (ranks are threads of one process), flag words
// are atomics, waits spin (yielding) until the other rank-thread has stored the number or the timeout passes
int mgk_ipc_alloc(mgk_ctx *c, size_t bytes, void **ptr, void *handle64) {
    if (!c || !ptr || !handle64 || !bytes) return fail(MGK_EINVAL, "mgk_ipc_alloc");
    *ptr = calloc(1, bytes);
    if (!*ptr) return fail(MGK_EINVAL, "mgk_ipc_alloc: out of memory");
    memset(handle64, 0, MGK_IPC_HANDLE_BYTES);
    memcpy(handle64, ptr, sizeof(void *));
    return 0;
}
int mgk_ipc_open(mgk_ctx *c, const void *handle64, void **ptr) { if (!c || !handle64 || !ptr) return fail(MGK_EINVAL, "mgk_ipc_open"); memcpy(ptr, handle64, sizeof(void *)); return 0; }
int mgk_ipc_close(mgk_ctx *, void *) { return 0; }
int mgk_peer_copy(mgk_ctx *c, void *d, const void *s, size_t n, void *) { if (!c || !d || !s) return fail(MGK_EINVAL, "mgk_peer_copy"); memcpy(d, s, n); return 0; }
int mgk_flags_set(mgk_ctx *c, void *const *flags, int n, unsigned long long value, void *) {
    if (!c || !flags || n < 1 || n > MGK_PEER_MAX) return fail(MGK_EINVAL, "mgk_flags_set");
    for (int q = 0; q < n; q++) __atomic_store_n((unsigned long long *)flags[q], value, __ATOMIC_RELEASE);
    return 0;
}
static bool mock_wait(const unsigned long long *f, unsigned long long value, double timeout_s) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (__atomic_load_n(f, __ATOMIC_ACQUIRE) < value) {
        sched_yield();
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > timeout_s) return false;
    }
    return true;
}
int mgk_flags_wait(mgk_ctx *c, void *const *flags, int n, unsigned long long value, double timeout_s, void *status, void *) {
    if (!c || !flags || n < 1 || n > MGK_PEER_MAX || !status || !(timeout_s > 0.0)) return fail(MGK_EINVAL, "mgk_flags_wait");
    for (int q = 0; q < n; q++) if (!mock_wait((const unsigned long long *)flags[q], value, timeout_s)) __atomic_store_n((unsigned int *)status, 1u, __ATOMIC_RELAXED);
    return 0;
}
int mgk_flag_set(mgk_ctx *c, void *flag, unsigned long long value, void *s) { void *f[1] = {flag}; return mgk_flags_set(c, f, 1, value, s); }
int mgk_flag_wait(mgk_ctx *c, const void *flag, unsigned long long value, double timeout_s, void *status, void *s) { void *f[1] = {const_cast<void *>(flag)}; return mgk_flags_wait(c, f, 1, value, timeout_s, status, s); }
int mgk_peer_allreduce(mgk_ctx *c, void *const *blocks, int nranks, int me, unsigned long long seq, double *vals, int n, double timeout_s, void *status, void *) {
    if (!c || !blocks || nranks < 1 || nranks > MGK_PEER_MAX || me < 0 || me >= nranks || !vals || n < 1 || n > 64 || !status) return fail(MGK_EINVAL, "mgk_peer_allreduce");
    const int buf = (int)(seq & 1ull);
    for (int r = 0; r < nranks; r++) {
        unsigned long long *slot = (unsigned long long *)blocks[r] + 65 * (buf * nranks + me);
        memcpy(slot + 1, vals, sizeof(double) * (size_t)n);
        __atomic_store_n(slot, seq, __ATOMIC_RELEASE);
    }
    const unsigned long long *own = (const unsigned long long *)blocks[me] + 65 * buf * nranks;
    for (int r = 0; r < nranks; r++) if (!mock_wait(own + 65 * r, seq, timeout_s)) __atomic_store_n((unsigned int *)status, 1u, __ATOMIC_RELAXED);
    for (int q = 0; q < n; q++) {
        double s = 0.0;
        for (int r = 0; r < nranks; r++) { double v; memcpy(&v, own + 65 * r + 1 + q, sizeof(double)); s += v; }
        vals[q] = s;
    }
    return 0;
}

// round 3: three sweeps per pass (2-D): compositions of the single sweep
static void j3_sweeps(const mgk_geom &G, const double *coef, double dinv, double scale, const double *ctab, const double *dtab, const double *b,
                      const double *first, double *o, int nsweeps) {
    // `nsweeps` sweeps starting from the field `first` (not modified), the last one into o
    std::vector<double> w1(G.total, 0.0), w2(G.total, 0.0);
    const double *src = first;
    for (int q = 0; q < nsweeps; q++) {
        double *dst = (q == nsweeps - 1) ? o : (q & 1 ? w2.data() : w1.data());
        st_op<double>(M_JACOBI, G, ctab ? nullptr : coef, ctab ? 1.0 : dinv, scale, 0, 0, 0, b, src, (const double *)nullptr, dst, 0, G.ny, ctab, dtab);
        src = dst;
    }
}
int mgk_jacobi3_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *o, void *) {
    if (!c || !g || g->dim != 3 || !coef || !b || !u || !o || u == o || b == o) return fail(MGK_EINVAL, "mgk_jacobi3_f64");
    const mgk_geom G = *g; std::vector<double> k(coef, coef + 7);
    return run(c, [=] {
        std::vector<double> w1(G.total, 0.0), w2(G.total, 0.0);
        st_op<double>(M_JACOBI, G, k.data(), dinv, scale, 0, 0, 0, b, u, (const double *)nullptr, w1.data(), 0, G.nz);
        st_op<double>(M_JACOBI, G, k.data(), dinv, scale, 0, 0, 0, b, w1.data(), (const double *)nullptr, w2.data(), 0, G.nz);
        st_op<double>(M_JACOBI, G, k.data(), dinv, scale, 0, 0, 0, b, w2.data(), (const double *)nullptr, o, 0, G.nz);
    });
}
int mgk_jacobi3_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *o, double *out, void *s) {
    if (!c || !g || g->dim != 3 || !coef || !b || !u || !o || u == o || b == o || !out) return fail(MGK_EINVAL, "mgk_jacobi3_sumsq_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    std::vector<double> r(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, g->nz);
    int rc = mgk_jacobi3_f64(c, g, coef, dinv, scale, b, u, o, s);
    if (rc) return rc;
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->nz), out);
    return 0;
}
int mgk_jacobi3_2d_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab, const double *b, const double *u, double *o, void *) {
    if (!c || !g || g->dim != 2 || (!coef && !ctab) || (ctab && !dtab) || !b || !u || !o || u == o || b == o) return fail(MGK_EINVAL, "mgk_jacobi3_2d_f64");
    const mgk_geom G = *g; std::vector<double> k(7, 0.0); if (coef) k.assign(coef, coef + 7);
    return run(c, [=] { j3_sweeps(G, k.data(), dinv, scale, ctab, dtab, b, u, o, 3); });
}
int mgk_jacobi3_2d_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab, const double *b, const double *u, double *o, double *out, void *) {
    if (!c || !g || g->dim != 2 || (!coef && !ctab) || (ctab && !dtab) || !b || !u || !o || u == o || b == o || !out) return fail(MGK_EINVAL, "mgk_jacobi3_2d_sumsq_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    std::vector<double> r(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, ctab ? nullptr : coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, g->ny, ctab, (const double *)nullptr);
    j3_sweeps(*g, coef, dinv, scale, ctab, dtab, b, u, o, 3);
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->ny), out);
    return 0;
}
int mgk_jacobi3_2d_sumsq_store_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab, const double *b, const double *u, double *o, double *r, double *out, void *) {
    if (!c || !g || g->dim != 2 || (!coef && !ctab) || (ctab && !dtab) || !b || !u || !o || !r || u == o || b == o || r == o || r == u || r == b || !out) return fail(MGK_EINVAL, "mgk_jacobi3_2d_sumsq_store_f64");
    if (c->capturing) return fail(MGK_EINVAL, "reduction to the host inside a capture");
    st_op<double>(M_RESIDUAL, *g, ctab ? nullptr : coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r, 0, g->ny, ctab, (const double *)nullptr);
    j3_sweeps(*g, coef, dinv, scale, ctab, dtab, b, u, o, 3);
    deliver(c, sumsq_field<double>(*g, r, 0, g->ny), out);
    return 0;
}
int mgk_jacobi3_2d_zero_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab, const double *b, double *o, void *) {
    if (!c || !g || g->dim != 2 || (!coef && !ctab) || (ctab && !dtab) || !b || !o || b == o) return fail(MGK_EINVAL, "mgk_jacobi3_2d_zero_f64");
    const mgk_geom G = *g; std::vector<double> k(7, 0.0); if (coef) k.assign(coef, coef + 7);
    return run(c, [=] {
        std::vector<double> z(G.total, 0.0);
        for (int i = 0; i < G.ny; i++) for (int j = 0; j < G.nx; j++) { const double zx = at(b, G, 0, i, j) * (dtab ? dtab[i] : dinv); at(z.data(), G, 0, i, j) = scale * zx; }
        j3_sweeps(G, k.data(), dinv, scale, ctab, dtab, b, z.data(), o, 2);
    });
}
int mgk_prolong_jacobi3_2d_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                               const double *b, const double *uc, const double *u, double *o, void *) {
    if (!c || (!coef && !ctab) || (ctab && !dtab) || !b || !uc || !u || !o || u == o || b == o || !xfer_ok(gf, gc) || gf->dim != 2) return fail(MGK_EINVAL, "mgk_prolong_jacobi3_2d_f64");
    const mgk_geom F = *gf, Cg = *gc; std::vector<double> k(7, 0.0); if (coef) k.assign(coef, coef + 7);
    return run(c, [=] { std::vector<double> t = corrected<double>(F, Cg, uc, u); j3_sweeps(F, k.data(), dinv, scale, ctab, dtab, b, t.data(), o, 3); });
}

int mgk_pack_f64(mgk_ctx *c, const mgk_geom *g, const double *compact, double *padded, void *) {
    if (!c || !g || !compact || !padded) return fail(MGK_EINVAL, "mgk_pack_f64");
    const mgk_geom G = *g;
    return run(c, [=] { for (int k = 0; k < G.nz; k++) for (int i = 0; i < G.ny; i++) memcpy(&at(padded, G, k, i, 0), compact + ((long)k * G.ny + i) * G.nx, sizeof(double) * (size_t)G.nx); });
}
int mgk_unpack_f64(mgk_ctx *c, const mgk_geom *g, const double *padded, double *compact, void *) {
    if (!c || !g || !compact || !padded) return fail(MGK_EINVAL, "mgk_unpack_f64");
    const mgk_geom G = *g;
    return run(c, [=] { for (int k = 0; k < G.nz; k++) for (int i = 0; i < G.ny; i++) memcpy(compact + ((long)k * G.ny + i) * G.nx, &at(padded, G, k, i, 0), sizeof(double) * (size_t)G.nx); });
}
int mgk_pack_f32(mgk_ctx *c, const mgk_geom *g, const double *compact, float *padded, void *) {
    if (!c || !g || !compact || !padded) return fail(MGK_EINVAL, "mgk_pack_f32");
    const mgk_geom G = *g;
    return run(c, [=] { for (int k = 0; k < G.nz; k++) for (int i = 0; i < G.ny; i++) for (int j = 0; j < G.nx; j++) at(padded, G, k, i, j) = (float)compact[((long)k * G.ny + i) * G.nx + j]; });
}
int mgk_unpack_f32(mgk_ctx *c, const mgk_geom *g, const float *padded, double *compact, void *) {
    if (!c || !g || !compact || !padded) return fail(MGK_EINVAL, "mgk_unpack_f32");
    const mgk_geom G = *g;
    return run(c, [=] { for (int k = 0; k < G.nz; k++) for (int i = 0; i < G.ny; i++) for (int j = 0; j < G.nx; j++) compact[((long)k * G.ny + i) * G.nx + j] = (double)at(padded, G, k, i, j); });
}
int mgk_fill_separable_f64(mgk_ctx *c, const mgk_geom *g, const double *cx, const double *sy, const double *sz, double *out, void *) {
    if (!c || !g || !cx || !sy || !out || (g->dim == 3 && !sz)) return fail(MGK_EINVAL, "mgk_fill_separable_f64");
    const mgk_geom G = *g;
    return run(c, [=] { for (int k = 0; k < G.nz; k++) for (int i = 0; i < G.ny; i++) for (int j = 0; j < G.nx; j++) { double v = cx[j] * sy[i]; if (G.dim == 3) v = v * sz[k]; at(out, G, k, i, j) = v; } });
}
int mgk_error_sums_f64(mgk_ctx *c, const mgk_geom *g, const double *u, const double *sx, const double *sy, const double *sz, double *e3, void *) {
    if (!c || !g || !u || !sx || !sy || !e3 || (g->dim == 3 && !sz)) return fail(MGK_EINVAL, "mgk_error_sums_f64");
    double m = 0, s1 = 0, s2 = 0;
    for (int k = 0; k < g->nz; k++) for (int i = 0; i < g->ny; i++) for (int j = 0; j < g->nx; j++) {
        double sol = sx[j] * sy[i]; if (g->dim == 3) sol = sol * sz[k];
        const double d = fabs(at(u, *g, k, i, j) - sol);
        m = fmax(m, d); s1 += d; s2 += d * d;
    }
    e3[0] = m; e3[1] = s1; e3[2] = s2;
    return 0;
}
// (the _Pragma lines: no-ops in the test builds; bench.py's CPU baseline artefact build/refdriver/poisson_cpu is this file compiled -O3 -fopenmp --
// the CSR / BLAS-1 data path of the reference's PETSc on the host cores)
#define FLAT(NAME, ARGS, BODY) int NAME ARGS { if (!c) return fail(MGK_EINVAL, #NAME); return run(c, [=] { _Pragma("omp parallel for schedule(static)") for (long q = 0; q < n; q++) { BODY; } }); }
FLAT(mgk_flat_axpy, (mgk_ctx *c, long n, double a, const double *x, double *y, void *), y[q] = y[q] + a * x[q])
FLAT(mgk_flat_aypx, (mgk_ctx *c, long n, double a, const double *x, double *y, void *), y[q] = x[q] + a * y[q])
FLAT(mgk_flat_axpbypcz, (mgk_ctx *c, long n, double a, double b, double g, const double *x, const double *y, double *z, void *), z[q] = (a * x[q] + b * y[q]) + g * z[q])
FLAT(mgk_flat_fill, (mgk_ctx *c, long n, double a, double *z, void *), z[q] = a)
FLAT(mgk_flat_scale, (mgk_ctx *c, long n, double a, double *z, void *), z[q] = a * z[q])
FLAT(mgk_flat_pointwise_mult, (mgk_ctx *c, long n, const double *x, const double *y, double *z, void *), z[q] = x[q] * y[q])
int mgk_flat_dot(mgk_ctx *c, long n, const double *x, const double *y, double *out, void *) {
    if (!c || !x || !y || !out) return fail(MGK_EINVAL, "mgk_flat_dot");
    long double s = 0;
    _Pragma("omp parallel for schedule(static) reduction(+:s)")
    for (long q = 0; q < n; q++) s += (long double)x[q] * y[q];
    deliver(c, (double)s, out);
    return 0;
}
int mgk_stream_triad_f64(mgk_ctx *c, long n, double *a, const double *b, const double *cc, double s, int, int, void *) {
    if (!c || !a || !b || !cc) return fail(MGK_EINVAL, "mgk_stream_triad_f64");
    for (long q = 0; q < n; q++) a[q] = b[q] + s * cc[q];
    return 0;
}
int mgk_csr_mult_f64(mgk_ctx *c, long nrows, const long *rowptr, const int *col, const double *val, const double *x, double *y, double alpha,
                     const double *addto, int row_n, long row_pitch, long row_org, void *) {
    if (!c || !rowptr || !col || !val || !x || !y) return fail(MGK_EINVAL, "mgk_csr_mult_f64");
    return run(c, [=] {
        _Pragma("omp parallel for schedule(static)")
        for (long r = 0; r < nrows; r++) {
            double sum = 0.0;
            for (long q = rowptr[r]; q < rowptr[r + 1]; q++) sum += val[q] * x[col[q]];
            const long o = row_n > 0 ? row_org + (r / row_n) * row_pitch + (r % row_n) : r;
            y[o] = addto ? addto[o] + alpha * sum : sum;
        }
    });
}
// mixed precision bridges
int mgk_residual_f64_to_f32(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const double *coef, const double *b, const double *u, float *r32, double *out, void *) {
    if (!c || !g || !g32 || !coef || !b || !u || !r32 || !out || g->nx != g32->nx || g->ny != g32->ny || g->nz != g32->nz) return fail(MGK_EINVAL, "mgk_residual_f64_to_f32");
    std::vector<double> r(g->total, 0.0);
    st_op<double>(M_RESIDUAL, *g, coef, 1, 1, 0, 0, 0, b, u, (const double *)nullptr, r.data(), 0, g->nz);
    for (int k = 0; k < g->nz; k++) for (int i = 0; i < g->ny; i++) for (int j = 0; j < g->nx; j++) at(r32, *g32, k, i, j) = (float)at(r.data(), *g, k, i, j);
    deliver(c, sumsq_field<double>(*g, r.data(), 0, g->nz), out);
    return 0;
}
static void jz32(const mgk_geom &H, const float *r32, float *e0, double dinv, double scale) {
    for (int k = 0; k < H.nz; k++) for (int i = 0; i < H.ny; i++) for (int j = 0; j < H.nx; j++) { const float zx = at(r32, H, k, i, j) * (float)dinv; at(e0, H, k, i, j) = (float)scale * zx; }
}
int mgk_residual_f64_to_f32_jz(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const double *coef, const double *b, const double *u, float *r32, float *e0,
                               double dinv, double scale, double *out, void *s) {
    if (!e0 || e0 == r32) return fail(MGK_EINVAL, "mgk_residual_f64_to_f32_jz");
    int rc = mgk_residual_f64_to_f32(c, g, g32, coef, b, u, r32, out, s);
    if (!rc) jz32(*g32, r32, e0, dinv, scale);
    return rc;
}
int mgk_correct_f64_from_f32(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const float *e, double *u, void *) {
    if (!c || !g || !g32 || !e || !u || g->nx != g32->nx || g->ny != g32->ny || g->nz != g32->nz) return fail(MGK_EINVAL, "mgk_correct_f64_from_f32");
    const mgk_geom G = *g, H = *g32;
    return run(c, [=] { for (int k = 0; k < G.nz; k++) for (int i = 0; i < G.ny; i++) for (int j = 0; j < G.nx; j++) at(u, G, k, i, j) = at(u, G, k, i, j) + (double)at(e, H, k, i, j); });
}
int mgk_correct_residual_f64_f32(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const double *coef, const double *b, const double *u, const float *e, double *unew, float *r32, double *out, void *s) {
    if (!c || !g || !g32 || !coef || !b || !u || !e || !unew || !r32 || !out || u == unew) return fail(MGK_EINVAL, "mgk_correct_residual_f64_f32");
    memcpy(unew, u, sizeof(double) * (size_t)g->total);
    int rc = mgk_correct_f64_from_f32(c, g, g32, e, unew, s);
    return rc ? rc : mgk_residual_f64_to_f32(c, g, g32, coef, b, unew, r32, out, s);
}
int mgk_correct_residual_f64_f32_jz(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const double *coef, const double *b, const double *u, const float *e, double *unew,
                                    float *r32, float *e0, double dinv, double scale, double *out, void *s) {
    if (!e0 || e0 == r32 || (const float *)e0 == e) return fail(MGK_EINVAL, "mgk_correct_residual_f64_f32_jz");
    int rc = mgk_correct_residual_f64_f32(c, g, g32, coef, b, u, e, unew, r32, out, s);
    if (!rc) jz32(*g32, r32, e0, dinv, scale);
    return rc;
}
}   // extern "C"
