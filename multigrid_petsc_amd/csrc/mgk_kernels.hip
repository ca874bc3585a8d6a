// mgk_kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels of the multigrid V-cycle hot path
// and their C ABI (include/mgk.h).  HBM-bound fp64 stencil work: no MFMA; 64-wide wavefronts,
// 16-byte-per-lane coalesced HBM access, LDS-staged plane tiles, register marching along the
// slowest axis, wavefront-shuffle reductions.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off   (NO fast-math: results must be
// bit-identical to the canonical arithmetic restated in oracle/mgo.c).
//
// Reference operations replaced (paths relative to /root/reference):
//   k_stencil<MODE 0>  KSPSolve Richardson+Jacobi sweep        src/solver.c:1531,1536,1542
//   k_stencil<MODE 1>  KSPBuildResidual / MatMult+VecAXPY      src/solver.c:1516-1517,1534,1545
//   k_stencil<MODE 2>  KSPChebyshev recurrence step            (KSPSetFromOptions, src/solver.c:1476)
//   k_stencil<MODE 3>  KSPBuildResidual fused with VecNorm     src/solver.c:1545-1546
//   k_restrict*        MatMult(res[l],r,b[l+1])                src/solver.c:1535 (matrix :1071-1092)
//   k_prolong*         MatMult(pro[l],u,rv)+VecAXPY            src/solver.c:1540-1541 (matrix :1131-1152)
//   k_sumsq            VecNorm(NORM_2)                         src/solver.c:1512,1518,1546
#include "mgk_dev.hpp"

thread_local char g_err[512] = "ok";
int fail(int code, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s (code %d%s%s)", what, code,
             code < 10000 ? ": " : "", code < 10000 ? hipGetErrorString((hipError_t)code) : "");
    return code;
}

// ------------------------------------------------------------------------------------------
// error handling / context
// ------------------------------------------------------------------------------------------

extern "C" const char *mgk_last_error(void) { return g_err; }

extern "C" int mgk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int mgk_set_device(int device) { HIPCHK(hipSetDevice(device)); return 0; }

__global__ void k_finish_sum(const double *partials, int n, double *out, int slot);
extern "C" int mgk_ctx_create(mgk_ctx **out, int device) {
    if (!out) return fail(MGK_EINVAL, "mgk_ctx_create: null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MGK_ENOGPU, "mgk_ctx_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(MGK_EINVAL, "mgk_ctx_create: device index out of range");
    HIPCHK(hipSetDevice(device));
    mgk_ctx *c = new mgk_ctx();
    c->device = device;
    HIPCHK(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
    {   // the comm stream gets the highest priority the device offers: when CUs free up, the (few) workgroups of an exchange are
        // dispatched before the next workgroups of the marching kernel that fills the chip
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
        if (hipStreamCreateWithPriority(&c->comm, hipStreamNonBlocking, hi) != hipSuccess)
            HIPCHK(hipStreamCreateWithFlags(&c->comm, hipStreamNonBlocking));
    }
    c->chunk_planes = 0;
    c->max_partials = 16384;
    HIPCHK(hipMalloc(&c->partials, sizeof(double) * 3 * c->max_partials));
    HIPCHK(hipMalloc(&c->result_dev, sizeof(double) * 8));
    HIPCHK(hipHostMalloc(&c->result_host, sizeof(double) * 8, hipHostMallocDefault));
    for (int q = 0; q < 32; q++) HIPCHK(hipEventCreateWithFlags(&c->ev[q], hipEventDisableTiming));
    c->ev_next = 0;
    {   // HIP loads a translation unit's code object at the first use of one of its kernels (~2 ms each: a third of the reference driver's
        // whole `Solver walltime` at 4097^2 when it happened inside its first KSPSolve): both units are loaded here
        hipFuncAttributes fa;
        HIPCHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_finish_sum)));
        int rc = mgk_preload_kernels3();
        if (rc) return rc;
    }
    *out = c;
    return 0;
}

extern "C" void mgk_ctx_destroy(mgk_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(c->partials);
    (void)hipFree(c->result_dev);
    (void)hipHostFree(c->result_host);
    for (int q = 0; q < 32; q++) (void)hipEventDestroy(c->ev[q]);
    (void)hipStreamDestroy(c->compute);
    (void)hipStreamDestroy(c->comm);
    delete c;
}

extern "C" int mgk_ctx_set_chunk_planes(mgk_ctx *c, int planes) {
    if (!c || planes < 0) return fail(MGK_EINVAL, "mgk_ctx_set_chunk_planes: planes >= 0");
    c->chunk_planes = planes;
    return 0;
}
extern "C" void *mgk_stream_compute(mgk_ctx *c) { return (void *)c->compute; }
extern "C" void *mgk_stream_comm(mgk_ctx *c) { return (void *)c->comm; }

extern "C" int mgk_malloc(mgk_ctx *c, void **dptr, size_t bytes) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 8));
    HIPCHK(hipMemsetAsync(*dptr, 0, bytes ? bytes : 8, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}
extern "C" int mgk_free(mgk_ctx *c, void *dptr) { (void)c; HIPCHK(hipFree(dptr)); return 0; }
extern "C" int mgk_memset0(mgk_ctx *c, void *dptr, size_t bytes, void *stream) {
    HIPCHK(hipMemsetAsync(dptr, 0, bytes, S(c, stream)));
    return 0;
}
extern "C" int mgk_h2d(mgk_ctx *c, void *dst, const void *src, size_t bytes) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}
extern "C" int mgk_d2h(mgk_ctx *c, void *dst, const void *src, size_t bytes) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    return 0;
}
extern "C" int mgk_d2d(mgk_ctx *c, void *dst, const void *src, size_t bytes, void *stream) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(c, stream)));
    return 0;
}
extern "C" int mgk_sync(mgk_ctx *c, void *stream) {
    if (stream) HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    else { HIPCHK(hipSetDevice(c->device)); HIPCHK(hipDeviceSynchronize()); }
    return 0;
}

// pinned host memory and stream-ordered copies (device-resident reductions read back without a device-wide sync)
extern "C" int mgk_host_alloc(mgk_ctx *c, void **hptr, size_t bytes) {
    (void)c;
    HIPCHK(hipHostMalloc(hptr, bytes ? bytes : 8, hipHostMallocDefault));
    memset(*hptr, 0, bytes ? bytes : 8);
    return 0;
}
extern "C" int mgk_host_free(mgk_ctx *c, void *hptr) { (void)c; if (hptr) HIPCHK(hipHostFree(hptr)); return 0; }
extern "C" int mgk_d2h_async(mgk_ctx *c, void *dst_pinned, const void *src, size_t bytes, void *stream) {
    HIPCHK(hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, S(c, stream)));
    return 0;
}
extern "C" int mgk_h2d_async(mgk_ctx *c, void *dst, const void *src_pinned, size_t bytes, void *stream) {
    HIPCHK(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, S(c, stream)));
    return 0;
}
// occupies ONE wavefront for `us` microseconds of wall time (s_memrealtime ticks at 100 MHz): the stand-in for the time a
// plane spends on an xGMI link when one rank's share of a multi-GPU run is measured on a single GPU (phantom communicator)
__global__ void __launch_bounds__(64) k_delay(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// `bytes` (a multiple of 16) from src to dst by a handful of small workgroups that then stay resident until `us` microseconds have
// passed since they started: the footprint of a send/recv kernel that moves a plane at link speed (phantom communicator)
__global__ void __launch_bounds__(256) k_paced_copy(const uint4 *src, uint4 *dst, long n16, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n16; q += (long)gridDim.x * blockDim.x) dst[q] = src[q];
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int mgk_paced_copy(mgk_ctx *c, void *dst, const void *src, size_t bytes, double us, int blocks, void *stream) {
    if (!c || !dst || !src || (bytes & 15) || us < 0.0 || us > 1.0e6 || blocks < 1 || blocks > 64) return fail(MGK_EINVAL, "mgk_paced_copy: bad arguments");
    hipLaunchKernelGGL(k_paced_copy, dim3((unsigned)blocks), dim3(256), 0, S(c, stream), (const uint4 *)src, (uint4 *)dst, (long)(bytes / 16),
                       (unsigned long long)(us * 100.0));
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_delay_us(mgk_ctx *c, double us, void *stream) {
    if (!c || us < 0.0 || us > 1.0e6) return fail(MGK_EINVAL, "mgk_delay_us: 0 <= us <= 1e6");
    if (us == 0.0) return 0;
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, S(c, stream), (unsigned long long)(us * 100.0));
    HIPCHK(hipGetLastError());
    return 0;
}

struct mgk_timer { hipEvent_t a, b; };
extern "C" int mgk_timer_create(mgk_ctx *c, void **t) {
    (void)c;
    mgk_timer *x = new mgk_timer();
    HIPCHK(hipEventCreate(&x->a));
    HIPCHK(hipEventCreate(&x->b));
    *t = x;
    return 0;
}
extern "C" int mgk_timer_start(mgk_ctx *c, void *t, void *s) { HIPCHK(hipEventRecord(((mgk_timer *)t)->a, S(c, s))); return 0; }
extern "C" int mgk_timer_stop(mgk_ctx *c, void *t, void *s) { HIPCHK(hipEventRecord(((mgk_timer *)t)->b, S(c, s))); return 0; }
extern "C" int mgk_timer_elapsed_ms(mgk_ctx *c, void *t, double *ms) {
    (void)c;
    mgk_timer *x = (mgk_timer *)t;
    HIPCHK(hipEventSynchronize(x->b));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, x->a, x->b));
    *ms = (double)f;
    return 0;
}
extern "C" void mgk_timer_destroy(mgk_ctx *c, void *t) {
    (void)c;
    mgk_timer *x = (mgk_timer *)t;
    if (!x) return;
    (void)hipEventDestroy(x->a);
    (void)hipEventDestroy(x->b);
    delete x;
}
extern "C" int mgk_stream_wait(mgk_ctx *c, void *waiter, void *signaller) {
    // A wait captures the state of the event at the time of the call, so an event of the ring can be
    // re-recorded later without disturbing earlier waits.
    hipEvent_t ev = c->ev[c->ev_next];
    c->ev_next = (c->ev_next + 1) & 31;
    HIPCHK(hipEventRecord(ev, S(c, signaller)));
    HIPCHK(hipStreamWaitEvent(S(c, waiter), ev, 0));
    return 0;
}

// ---- HIP graphs: the launch-bound coarse part of a V-cycle is captured once and replayed ----
extern "C" int mgk_capture_begin(mgk_ctx *c) {
    HIPCHK(hipStreamBeginCapture(c->compute, hipStreamCaptureModeThreadLocal));
    return 0;
}
extern "C" int mgk_capture_end(mgk_ctx *c, void **graph_exec) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamEndCapture(c->compute, &g));
    hipGraphExec_t e = nullptr;
    hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (rc != hipSuccess) return fail((int)rc, "hipGraphInstantiate");
    *graph_exec = (void *)e;
    return 0;
}
extern "C" int mgk_graph_launch(mgk_ctx *c, void *graph_exec) {
    HIPCHK(hipGraphLaunch((hipGraphExec_t)graph_exec, c->compute));
    return 0;
}
extern "C" void mgk_graph_destroy(mgk_ctx *c, void *graph_exec) {
    (void)c;
    if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

// ------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------
extern "C" int mgk_geom_init(mgk_geom *g, int dim, int nx, int ny, int nz) {
    if (!g || (dim != 2 && dim != 3) || nx < 1 || ny < 1 || nz < 1 || (nx & 1) == 0)
        return fail(MGK_EINVAL, "mgk_geom_init: need dim in {2,3}, odd nx >= 1, ny,nz >= 1");
    if (dim == 2) nz = 1;
    g->dim = dim; g->nx = nx; g->ny = ny; g->nz = nz;
    g->pitch = ((MGK_XOFF + nx + 1 + 15) / 16) * 16;
    g->plane = (long)g->pitch * (ny + 2);
    if (dim == 3) { g->org = g->plane + g->pitch + MGK_XOFF; g->total = g->plane * (nz + 2) + g->pitch; }
    else { g->org = g->pitch + MGK_XOFF; g->total = g->plane + g->pitch; }
    return 0;
}

// ------------------------------------------------------------------------------------------
// the marching stencil kernel
// ------------------------------------------------------------------------------------------
// A block owns an (x,y) tile of TX = 128*WX by TY = WY*RY unknowns and marches along z (3-D) or
// along y (2-D) over `zc` planes.  Each lane owns one aligned pair of x-neighbours (16-byte
// loads: a wave row is 1 KiB contiguous) for RY consecutive rows.  Planes z-1, z, z+1 of the
// own points live in registers (plus z+2 in flight); the plane being processed is staged in a
// double-buffered LDS tile from which the x neighbours and the y neighbours owned by other
// waves are read.  y halos across tile borders go straight from global memory to the consuming
// lane; x halos across tile borders are fetched by the two edge lanes of a row.
// Each u plane is read from HBM once per tile column (+ halo lines that neighbouring tiles also
// touch: L2 / Infinity-Cache hits), b once, the output written once: 24 B per unknown.

// aligned pair of unknowns (row kernels): double2 / float2
template <typename T> struct alignas(2 * sizeof(T)) P2 { T x, y; };
__device__ __forceinline__ P2<double> ldp_stream(const double *p) { double2 t = ld2_stream(p, true); P2<double> r; r.x = t.x; r.y = t.y; return r; }
__device__ __forceinline__ P2<float> ldp_stream(const float *p) { return *reinterpret_cast<const P2<float> *>(p); }
__device__ __forceinline__ void stp_stream(double *p, P2<double> v) { st2_stream(p, make_double2(v.x, v.y)); }
__device__ __forceinline__ void stp_stream(float *p, P2<float> v) { *reinterpret_cast<P2<float> *>(p) = v; }

template <typename T>
struct StArgs {
    const T *u, *b, *aux;
    const T *uc;           // MODE_PJACOBI: coarse correction field (interior origin), its strides and extents
    long crs, cms;
    int nxc, nyc, nzc;
    T *out;
    T *out2;               // MODE_JNORM, optional: the residual b - A u of the INPUT field is stored as well (the drop-in's VecNorm pass)
    float *out32;          // MODE_RES32 / MODE_CRES32: float copy of the fp64 residual (own strides below)
    float *jz32;           // optional: the fp32 cycle's first sweep from its zero guess, scale32 * (r32 * dinv32), same strides
    float dinv32, scale32;
    const float *e32;      // MODE_CRES32: fp32 correction added to u on the fly (same fp32 geometry as out32)
    double *partials;
    int nx, ny, nm;       // nm: number of marching planes (nz in 3-D, ny in 2-D)
    long rs, ms;          // row stride (3-D: pitch), marching stride (3-D: plane, 2-D: pitch)
    long ors, oms;        // strides of out32
    int zc, ntx, nty;
    int zbeg, zend;       // marching range [zbeg, zend) of this launch (whole grid: 0, nm)
    T a0, a1, a2, a3, a4, a5, a6;   // (m-1), S, W, C, E, N, (m+1); 2-D: S and N unused
    const T *ctab;        // 2-D only, optional: coefficients that vary with the marching index (stretched meshes):
    const T *dtab;        //   ctab[5*i .. 5*i+4] = {(i-1), W, C, E, (i+1)} of grid row i, dtab[i] = 1/diag
    T dinv, scale, ckm1, ck, cz;
};

enum { MODE_JACOBI = 0, MODE_RESIDUAL = 1, MODE_CHEBY = 2, MODE_RESNORM = 3, MODE_APPLY = 4, MODE_RES32 = 5, MODE_PJACOBI = 6,
       MODE_JNORM = 7,     // Jacobi sweep that also accumulates ||b - A u||^2 of its INPUT (norm of cycle k + first sweep of cycle k+1)
       MODE_CRES32 = 8 };  // mixed precision outer step in one pass: u' = u + (double)e32 (written), r32 = (float)(b - A u'), sum r^2

// Interpolated coarse correction (row of pro, src/solver.c:1140-1148) at the VX fine points (k, i, x0..x0+VX-1),
// x0 a multiple of VX: parents summed in ascending coarse index (kc, ic, jc) exactly like k_prolong_add.
// Ghost rows/planes (i or k = -1, n) are "odd" points whose single parent is the coarse ghost.
template <typename T>
__device__ __forceinline__ V16<T> prolong_vec(const T *uc, long crs, long cms, int k, int i, int x0) {
    constexpr int VX = 16 / sizeof(T);
    const int iodd = i & 1, kodd = k & 1;
    const int ic0 = iodd ? (i - 1) / 2 : i / 2 - 1, nic = iodd ? 1 : 2;
    const int kc0 = kodd ? (k - 1) / 2 : k / 2 - 1, nkc = kodd ? 1 : 2;
    const T wi = iodd ? (T)1 : (T)0.5, wk = kodd ? (T)1 : (T)0.5;
    const T wh = wk * (wi * (T)0.5), w1 = wk * (wi * (T)1);
    const T *cr = uc + (long)kc0 * cms + (long)ic0 * crs + (x0 / 2 - 1);
    V16<T> s = v16_zero<T>();
    for (int qk = 0; qk < nkc; qk++)
        for (int qi = 0; qi < nic; qi++) {
            const T *c = cr + (long)qk * cms + (long)qi * crs;
            const T c0 = c[0], c1 = c[1];
            s.v[0] += wh * c0;
            s.v[0] += wh * c1;
            s.v[1] += w1 * c1;
            if (VX == 4) {
                const T c2 = c[2];
                s.v[VX - 2] += wh * c1;
                s.v[VX - 2] += wh * c2;
                s.v[VX - 1] += w1 * c2;
            }
        }
    return s;
}
template <typename T>
__device__ __forceinline__ T prolong_one(const T *uc, long crs, long cms, int k, int i, int x) {
    const int iodd = i & 1, kodd = k & 1, xodd = x & 1;
    const int ic0 = iodd ? (i - 1) / 2 : i / 2 - 1, nic = iodd ? 1 : 2;
    const int kc0 = kodd ? (k - 1) / 2 : k / 2 - 1, nkc = kodd ? 1 : 2;
    const int jc0 = xodd ? (x - 1) / 2 : x / 2 - 1, njc = xodd ? 1 : 2;
    const T wi = iodd ? (T)1 : (T)0.5, wk = kodd ? (T)1 : (T)0.5, wj = xodd ? (T)1 : (T)0.5;
    const T w = wk * (wi * wj);
    const T *cr = uc + (long)kc0 * cms + (long)ic0 * crs + jc0;
    T s = (T)0;
    for (int qk = 0; qk < nkc; qk++)
        for (int qi = 0; qi < nic; qi++)
            for (int qj = 0; qj < njc; qj++) s += w * cr[(long)qk * cms + (long)qi * crs + qj];
    return s;
}
// 2-D fused prolongation: the parents of fine row i (x0..x0+VX-1) fetched raw one marching step ahead, summed later in
// the order of prolong_vec (k odd: one parent plane of weight 1, 1*(wi*wj) == wi*wj)
template <typename T>
__device__ __forceinline__ void prolong2d_raw(const T *uc, long crs, int i, int x0, bool ok, T (&pr)[2][3]) {
    constexpr int VX = 16 / sizeof(T);
    const int iodd = i & 1;
    const int ic0 = iodd ? (i - 1) / 2 : i / 2 - 1, nic = iodd ? 1 : 2;
    const T *cr = uc + (long)ic0 * crs + (x0 / 2 - 1);
#pragma unroll
    for (int qi = 0; qi < 2; qi++) {
        const bool use = ok && qi < nic;
        pr[qi][0] = use ? cr[(long)qi * crs] : (T)0;
        pr[qi][1] = use ? cr[(long)qi * crs + 1] : (T)0;
        pr[qi][2] = (use && VX == 4) ? cr[(long)qi * crs + 2] : (T)0;
    }
}
template <typename T>
__device__ __forceinline__ V16<T> prolong2d_sum(const T (&pr)[2][3], int i) {
    constexpr int VX = 16 / sizeof(T);
    const int iodd = i & 1, nic = iodd ? 1 : 2;
    const T wi = iodd ? (T)1 : (T)0.5;
    const T wh = (T)1 * (wi * (T)0.5), w1 = (T)1 * (wi * (T)1);
    V16<T> s = v16_zero<T>();
#pragma unroll
    for (int qi = 0; qi < 2; qi++) {
        if (qi < nic) {
            const T c0 = pr[qi][0], c1 = pr[qi][1];
            s.v[0] += wh * c0;
            s.v[0] += wh * c1;
            s.v[1] += w1 * c1;
            if (VX == 4) {
                const T c2 = pr[qi][2];
                s.v[VX - 2] += wh * c1;
                s.v[VX - 2] += wh * c2;
                s.v[VX - 1] += w1 * c2;
            }
        }
    }
    return s;
}
// MODE_CRES32: the fp32 correction at the VX fine points (k, i, x0..) / at one point, widened to T (ghost cells hold 0)
template <typename T>
__device__ __forceinline__ V16<T> e32_vec(const float *e, long ers, long ems, int k, int i, int x0) {
    constexpr int VX = 16 / sizeof(T);
    const float *p = e + (long)k * ems + (long)i * ers + x0;
    V16<T> s;
#pragma unroll
    for (int q = 0; q < VX; q += 2) {
        const float2 f = *reinterpret_cast<const float2 *>(p + q);
        s.v[q] = (T)f.x; s.v[q + 1] = (T)f.y;
    }
    return s;
}
// raw fp32 values of one lane vector (kept unconverted while the load is in flight)
template <int VX>
__device__ __forceinline__ void e32_raw(const float *p, float (&dst)[VX], bool ok) {
#pragma unroll
    for (int q = 0; q < VX; q += 2) {
        float2 f = make_float2(0.f, 0.f);
        if (ok) f = *reinterpret_cast<const float2 *>(p + q);
        dst[q] = f.x; dst[q + 1] = f.y;
    }
}
template <typename T>
__device__ __forceinline__ T e32_one(const float *e, long ers, long ems, int k, int i, int x) {
    return (T)e[(long)k * ems + (long)i * ers + x];
}
// The same sum with the parents taken from the block's LDS ring of coarse plane tiles (MODE_PJACOBI marching loop):
// cl[slot][row][col], slot = (kc+3)%3, row = ic - ic_base, col = jc - jc_base; `xh2` = (local x of the vector)/2.
template <typename T, int TYC, int CWP>
__device__ __forceinline__ V16<T> prolong_vec_lds(const T (&cl)[3][TYC][CWP], int ic_base, int k, int i, int xh2) {
    constexpr int VX = 16 / sizeof(T);
    const int iodd = i & 1, kodd = k & 1;
    const int ic0 = iodd ? (i - 1) / 2 : i / 2 - 1, nic = iodd ? 1 : 2;
    const int kc0 = kodd ? (k - 1) / 2 : k / 2 - 1, nkc = kodd ? 1 : 2;
    const T wi = iodd ? (T)1 : (T)0.5, wk = kodd ? (T)1 : (T)0.5;
    const T wh = wk * (wi * (T)0.5), w1 = wk * (wi * (T)1);
    V16<T> s = v16_zero<T>();
    int slot = (kc0 + 3) % 3;
    for (int qk = 0; qk < nkc; qk++, slot = (slot == 2 ? 0 : slot + 1))
        for (int qi = 0; qi < nic; qi++) {
            const T *c = &cl[slot][ic0 + qi - ic_base][xh2];
            const T c0 = c[0], c1 = c[1];
            s.v[0] += wh * c0;
            s.v[0] += wh * c1;
            s.v[1] += w1 * c1;
            if (VX == 4) {
                const T c2 = c[2];
                s.v[VX - 2] += wh * c1;
                s.v[VX - 2] += wh * c2;
                s.v[VX - 1] += w1 * c2;
            }
        }
    return s;
}
template <typename T>
__device__ __forceinline__ V16<T> vadd(const V16<T> &a, const V16<T> &b) {
    V16<T> r;
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); e++) r.v[e] = a.v[e] + b.v[e];
    return r;
}

template <typename T, int DIM, int WX, int WY, int RY, int MODE>
__global__ void __launch_bounds__(64 * WX * WY) k_stencil(const StArgs<T> a) {
    constexpr int VX = 16 / sizeof(T);           // unknowns per lane: one 16-byte access
    constexpr int TX = 64 * VX * WX;
    constexpr int TY = (DIM == 3) ? WY * RY : 1;
    constexpr int LW = TX + 2 * VX;
    static_assert(DIM == 3 || (WY == 1 && RY == 1), "2-D marches along y: one row per tile");
    __shared__ __attribute__((aligned(16))) T lds[2][TY][LW];
    using VT = V16<T>;
    // MODE_PJACOBI: ring of three coarse plane tiles (rows ic_base..ic_base+TYC-1, columns jc_base..) feeding the
    // interpolation of the planes the loop brings in; loaded cooperatively one step ahead of its first use
    // (only for blocks of <= 512 threads: a 1024-thread block is capped at 128 VGPRs and would spill)
    constexpr bool PJ = (MODE == MODE_PJACOBI) && (DIM == 3) && (64 * WX * WY <= 512);
    // modes that read u through an additive correction (coarse interpolant / fp32 correction) at every point they touch
    constexpr bool ADDU = (MODE == MODE_PJACOBI || MODE == MODE_CRES32);
    // (2-D: the marching index is the grid row; the interpolation helpers are called with a fixed odd plane index, whose
    //  single parent plane has weight 1: 1*(wi*wj) == wi*wj bit for bit; a.cms is then the coarse ROW stride)
#define ADDV(zz, yy, xx) ((MODE == MODE_CRES32) ? e32_vec<T>(a.e32, a.ors, a.oms, zz, yy, xx) \
                          : (DIM == 3) ? prolong_vec(a.uc, a.crs, a.cms, zz, yy, xx) : prolong_vec(a.uc, a.cms, 0L, 1, zz, xx))
#define ADD1(zz, yy, xx) ((MODE == MODE_CRES32) ? e32_one<T>(a.e32, a.ors, a.oms, zz, yy, xx) \
                          : (DIM == 3) ? prolong_one(a.uc, a.crs, a.cms, zz, yy, xx) : prolong_one(a.uc, a.cms, 0L, 1, zz, xx))
    constexpr int TYC = PJ ? (TY / 2 + 2) : 1, CW = PJ ? (TX / 2 + 2) : 1, CWP = PJ ? (CW + 2) : 1;
    constexpr int NTHR = 64 * WX * WY, NLC = PJ ? ((CW + NTHR - 1) / NTHR) : 1, NL = TYC * NLC;   // thread t loads column t (+k*NTHR) of every tile row
    static_assert(!PJ || (TY % 2 == 0), "fused prolongation needs an even tile height");
    __shared__ T cl[3][TYC][CWP];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wx = w % WX, wy = w / WX;

    // XCD-aware block remap: blocks b and b+8 share an XCD (L2); give every XCD a contiguous
    // range of tiles so that halo lines shared by neighbouring tiles hit in its L2.
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int tx = bid % a.ntx;
    const int t2 = bid / a.ntx;
    const int ty = t2 % a.nty, tz = t2 / a.nty;

    const int xl = wx * 64 * VX + VX * lane;     // local x of the own vector
    const int x0 = tx * TX + xl;                 // global x (multiple of VX)
    const int yb = (DIM == 3) ? ty * TY + wy * RY : 0;
    const int lrow = (DIM == 3) ? wy * RY : 0;   // first own row inside the LDS tile
    const int z0 = a.zbeg + tz * a.zc;
    const int z1 = min(z0 + a.zc, a.zend);
    if (z0 >= z1) return;

    const bool xok = x0 < a.nx;                  // vector in bounds (x0+VX-1 <= nx: right ghost at most)
    const bool lastvec = (x0 + VX > a.nx);       // elements at x >= nx are the right ghost / row padding: must stay 0
    bool rok[RY];
#pragma unroll
    for (int r = 0; r < RY; r++) rok[r] = xok && (DIM == 2 || yb + r < a.ny);

    const bool isW = (lane == 0 && wx == 0), isE = (lane == 63 && wx == WX - 1);
    const int xh = isW ? tx * TX - 1 : tx * TX + TX;
    const bool xhok = (isW || isE) && xh <= a.nx;
    const int xhl = isW ? VX - 1 : TX + VX;      // LDS column of the halo cell

    const long rowoff = (DIM == 3) ? (long)yb * a.rs : 0;
    const T *up_ = a.u + rowoff + x0;            // own vector, row 0, plane 0
    const T *bp_ = a.b + rowoff + x0;
    const T *ap_ = (MODE == MODE_CHEBY) ? a.aux + rowoff + x0 : nullptr;
    const T *hp_ = a.u + rowoff + xh;            // x-halo cell, row 0, plane 0
    // y halos across the tile border (3-D only)
    const bool needS = (DIM == 3) && (wy == 0);
    const bool needN = (DIM == 3) && (wy == WY - 1);
    const bool okS = needS && xok;                              // row yb-1 >= -1 always exists
    const bool okN = needN && xok && (yb + RY <= a.ny);         // ghost row ny is the last one

    VT um[RY], uc[RY], up[RY], uq[RY], bc[RY], bn[RY], ac[RY], an[RY];
    T xp[RY], xq[RY];
    VT hS = v16_zero<T>(), hN = hS, hSn = hS, hNn = hS;

    // ---- prologue: planes z0-1, z0, z0+1; stage plane z0 in LDS ----
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const long ro = (long)r * a.rs;
        um[r] = ldv(up_ + (long)(z0 - 1) * a.ms + ro, rok[r]);
        uc[r] = ldv(up_ + (long)z0 * a.ms + ro, rok[r]);
        up[r] = ldv(up_ + (long)(z0 + 1) * a.ms + ro, rok[r]);
        bc[r] = ldv_stream(bp_ + (long)z0 * a.ms + ro, rok[r]);
        if (MODE == MODE_CHEBY) ac[r] = ldv_stream(ap_ + (long)z0 * a.ms + ro, rok[r]);
        const bool hok = xhok && (DIM == 2 || yb + r < a.ny);
        T xc = hok ? hp_[(long)z0 * a.ms + ro] : (T)0;
        xp[r] = hok ? hp_[(long)(z0 + 1) * a.ms + ro] : (T)0;
        if (ADDU) {                          // u := u + P e (or + e32) on everything this block reads
            if (rok[r]) {
                um[r] = vadd(um[r], ADDV(z0 - 1, yb + r, x0));
                uc[r] = vadd(uc[r], ADDV(z0, yb + r, x0));
                up[r] = vadd(up[r], ADDV(z0 + 1, yb + r, x0));
            }
            if (hok) {
                xc = xc + ADD1(z0, yb + r, xh);
                xp[r] = xp[r] + ADD1(z0 + 1, yb + r, xh);
            }
        }
        *reinterpret_cast<VT *>(&lds[0][lrow + r][xl + VX]) = uc[r];
        if (isW || isE) lds[0][lrow + r][xhl] = xc;
    }
    if (DIM == 3) {
        hS = ldv(up_ + (long)z0 * a.ms - a.rs, okS);
        hN = ldv(up_ + (long)z0 * a.ms + (long)RY * a.rs, okN);
        if (ADDU) {
            if (okS) hS = vadd(hS, ADDV(z0, yb - 1, x0));
            if (okN) hN = vadd(hN, ADDV(z0, yb + RY, x0));
        }
    }

    // MODE_CRES32: fp32 corrections of the plane being brought in, loaded one step ahead like u itself
    float erw[RY][VX], ehSr[VX], ehNr[VX], exr[RY];
    T praw[2][3];       // MODE_PJACOBI in 2-D: raw parents of the row being brought in
    double acc = 0.0;   // MODE_RESNORM / MODE_RES32 / MODE_JNORM
    const int ic_base = PJ ? (ty * TY) / 2 - 1 : 0, jc_base = PJ ? (tx * TX) / 2 - 1 : 0;
    T cnew[NL];
    int kc_new = -2;                             // coarse plane held in cnew (none)
    if (PJ) {                                    // planes floor(z0/2), +1 are needed by the first step
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const int kc = z0 / 2 + pl;
#pragma unroll
            for (int qc = 0; qc < NLC; qc++) {
                const int jcl = threadIdx.x + qc * NTHR, jc = jc_base + jcl;
#pragma unroll
                for (int icl = 0; icl < TYC; icl++) {
                    const int ic = ic_base + icl;
                    const bool ok = (jcl < CW) && (kc <= a.nzc) && (ic <= a.nyc) && (jc <= a.nxc);
                    if (jcl < CW) cl[(kc + 3) % 3][icl][jcl] = ok ? a.uc[(long)kc * a.cms + (long)ic * a.crs + jc] : (T)0;
                }
            }
        }
    }

    for (int z = z0; z < z1; z++) {
        const int buf = (z - z0) & 1;
        const bool more = (z + 1 < z1);
        // ---- issue the loads consumed in the NEXT step ----
        if (more) {
#pragma unroll
            for (int r = 0; r < RY; r++) {
                const long ro = (long)r * a.rs;
                uq[r] = ldv(up_ + (long)(z + 2) * a.ms + ro, rok[r]);
                bn[r] = ldv_stream(bp_ + (long)(z + 1) * a.ms + ro, rok[r]);
                if (MODE == MODE_CHEBY) an[r] = ldv_stream(ap_ + (long)(z + 1) * a.ms + ro, rok[r]);
                const bool hok = xhok && (DIM == 2 || yb + r < a.ny);
                xq[r] = hok ? hp_[(long)(z + 2) * a.ms + ro] : (T)0;
            }
            if (DIM == 3) {
                hSn = ldv(up_ + (long)(z + 1) * a.ms - a.rs, okS);
                hNn = ldv(up_ + (long)(z + 1) * a.ms + (long)RY * a.rs, okN);
            }
            if (MODE == MODE_PJACOBI && DIM == 2) prolong2d_raw(a.uc, a.cms, z + 2, x0, rok[0], praw);
            if (MODE == MODE_CRES32) {
                const float *ep = a.e32 + (long)(z + 2) * a.oms + (long)yb * a.ors + x0;
#pragma unroll
                for (int r = 0; r < RY; r++) {
                    e32_raw<VX>(ep + (long)r * a.ors, erw[r], rok[r]);
                    const bool hok = xhok && (DIM == 2 || yb + r < a.ny);
                    exr[r] = hok ? a.e32[(long)(z + 2) * a.oms + (long)(yb + r) * a.ors + xh] : 0.f;
                }
                e32_raw<VX>(ep - a.oms - a.ors, ehSr, okS);
                e32_raw<VX>(ep - a.oms + (long)RY * a.ors, ehNr, okN);
            }
            if (PJ && (z & 1)) {                 // coarse plane (z+1)/2+1: first needed at step z+1
                kc_new = (z + 1) / 2 + 1;
#pragma unroll
                for (int qc = 0; qc < NLC; qc++) {
                    const int jcl = threadIdx.x + qc * NTHR, jc = jc_base + jcl;
#pragma unroll
                    for (int icl = 0; icl < TYC; icl++) {
                        const int ic = ic_base + icl;
                        const bool ok = (jcl < CW) && (kc_new <= a.nzc) && (ic <= a.nyc) && (jc <= a.nxc);
                        cnew[qc * TYC + icl] = ok ? a.uc[(long)kc_new * a.cms + (long)ic * a.crs + jc] : (T)0;
                    }
                }
            }
        }
        __syncthreads();   // lds[buf] (plane z) complete
        // coefficients of this marching step: launch constants, or (2-D stretched meshes) the table row of grid row z
        T c0 = a.a0, c2 = a.a2, c3 = a.a3, c4 = a.a4, c6 = a.a6, dv = a.dinv;
        if (DIM == 2 && a.ctab) {
            const T *cr_ = a.ctab + 5 * (long)z;
            c0 = cr_[0]; c2 = cr_[1]; c3 = cr_[2]; c4 = cr_[3]; c6 = cr_[4];
            if (a.dtab) dv = a.dtab[z];
        }

        // ---- neighbours of plane z ----
        T Wn[RY], En[RY];
#pragma unroll
        for (int r = 0; r < RY; r++) {
            Wn[r] = lds[buf][lrow + r][xl + VX - 1];
            En[r] = lds[buf][lrow + r][xl + 2 * VX];
        }
        VT Sn = hS, Nn = hN;
        if (DIM == 3 && WY > 1) {
            if (wy > 0) Sn = *reinterpret_cast<const VT *>(&lds[buf][lrow - 1][xl + VX]);
            if (wy < WY - 1) Nn = *reinterpret_cast<const VT *>(&lds[buf][lrow + RY][xl + VX]);
        }

        // ---- compute + store plane z ----
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const VT s2 = (r == 0) ? Sn : uc[r > 0 ? r - 1 : 0];
            const VT n2 = (r == RY - 1) ? Nn : uc[r < RY - 1 ? r + 1 : r];
            VT o, rsq;
#pragma unroll
            for (int e = 0; e < VX; e++) {
                const T wv = (e == 0) ? Wn[r] : uc[r].v[e > 0 ? e - 1 : 0];
                const T ev = (e == VX - 1) ? En[r] : uc[r].v[e < VX - 1 ? e + 1 : e];
                T t = c0 * um[r].v[e];
                if (DIM == 3) t = t + a.a1 * s2.v[e];
                t = t + c2 * wv;
                t = t + c3 * uc[r].v[e];
                t = t + c4 * ev;
                if (DIM == 3) t = t + a.a5 * n2.v[e];
                t = t + c6 * up[r].v[e];
                const T res = bc[r].v[e] - t;
                if (MODE == MODE_JNORM) rsq.v[e] = res;
                if (MODE == MODE_JACOBI || MODE == MODE_PJACOBI || MODE == MODE_JNORM) {
                    const T zz = res * dv;
                    o.v[e] = uc[r].v[e] + a.scale * zz;
                } else if (MODE == MODE_CHEBY) {
                    const T zz = res * dv;
                    o.v[e] = (a.ckm1 * ac[r].v[e] + a.ck * uc[r].v[e]) + a.cz * zz;
                } else if (MODE == MODE_APPLY) {
                    o.v[e] = t;
                } else {
                    o.v[e] = res;
                }
            }
            if (lastvec) {                       // the vector straddles the end of the row: ghost / padding stay 0
#pragma unroll
                for (int e = 0; e < VX; e++) if (x0 + e >= a.nx) { o.v[e] = (T)0; if (MODE == MODE_JNORM) rsq.v[e] = (T)0; }
            }
            if (rok[r]) {
                if (MODE == MODE_RESNORM || MODE == MODE_RES32 || MODE == MODE_CRES32) {
#pragma unroll
                    for (int e = 0; e < VX; e++) acc += (double)o.v[e] * (double)o.v[e];
                }
                if (MODE == MODE_JNORM) {
#pragma unroll
                    for (int e = 0; e < VX; e++) acc += (double)rsq.v[e] * (double)rsq.v[e];
                }
                if (MODE == MODE_RES32 || MODE == MODE_CRES32) {
                    float2 f; f.x = (float)o.v[0]; f.y = (float)o.v[VX - 1];     // T == double here (VX == 2)
                    *reinterpret_cast<float2 *>(a.out32 + (DIM == 3 ? (long)yb * a.ors : 0) + x0 + (long)z * a.oms + (long)r * a.ors) = f;
                    if (a.jz32) {                                               // k_jacobi_zero<float>'s arithmetic on the value just stored
                        float2 zq;
                        const float zx = f.x * a.dinv32, zy = f.y * a.dinv32;
                        zq.x = a.scale32 * zx; zq.y = a.scale32 * zy;
                        if (x0 + 1 == a.nx) zq.y = 0.f;
                        *reinterpret_cast<float2 *>(a.jz32 + (DIM == 3 ? (long)yb * a.ors : 0) + x0 + (long)z * a.oms + (long)r * a.ors) = zq;
                    }
                    if (MODE == MODE_CRES32) stv_stream(a.out + rowoff + x0 + (long)z * a.ms + (long)r * a.rs, uc[r]);   // the corrected u
                } else if (MODE != MODE_RESNORM) {
                    stv_stream(a.out + rowoff + x0 + (long)z * a.ms + (long)r * a.rs, o);
                    if (MODE == MODE_JNORM) { if (a.out2) stv_stream(a.out2 + rowoff + x0 + (long)z * a.ms + (long)r * a.rs, rsq); }
                }
            }
        }

        // ---- stage plane z+1 and rotate ----
        if (more) {
#pragma unroll
            for (int r = 0; r < RY; r++) {
                *reinterpret_cast<VT *>(&lds[buf ^ 1][lrow + r][xl + VX]) = up[r];
                if (isW || isE) lds[buf ^ 1][lrow + r][xhl] = xp[r];
                um[r] = uc[r]; uc[r] = up[r]; up[r] = uq[r];
                bc[r] = bn[r]; xp[r] = xq[r];
                if (MODE == MODE_CHEBY) ac[r] = an[r];
                if (MODE == MODE_CRES32) {
#pragma unroll
                    for (int e = 0; e < VX; e++) up[r].v[e] = up[r].v[e] + (T)erw[r][e];
                    xp[r] = xp[r] + (T)exr[r];
                } else if (MODE == MODE_PJACOBI && DIM == 2) {
                    if (rok[r]) up[r] = vadd(up[r], prolong2d_sum(praw, z + 2));
                    if (xhok) xp[r] = xp[r] + ADD1(z + 2, yb + r, xh);
                } else if (ADDU) {
                    if (rok[r]) up[r] = vadd(up[r], PJ ? prolong_vec_lds(cl, ic_base, z + 2, yb + r, xl / 2) : ADDV(z + 2, yb + r, x0));
                    if (xhok && (DIM == 2 || yb + r < a.ny)) xp[r] = xp[r] + ADD1(z + 2, yb + r, xh);
                }
            }
            hS = hSn; hN = hNn;
            if (MODE == MODE_CRES32) {
#pragma unroll
                for (int e = 0; e < VX; e++) { hS.v[e] = hS.v[e] + (T)ehSr[e]; hN.v[e] = hN.v[e] + (T)ehNr[e]; }
            } else if (ADDU) {
                if (okS) hS = vadd(hS, PJ ? prolong_vec_lds(cl, ic_base, z + 1, yb - 1, xl / 2) : ADDV(z + 1, yb - 1, x0));
                if (okN) hN = vadd(hN, PJ ? prolong_vec_lds(cl, ic_base, z + 1, yb + RY, xl / 2) : ADDV(z + 1, yb + RY, x0));
                if (PJ && kc_new >= -1) {        // publish the coarse plane fetched at the top of this step
                    const int slot = (kc_new + 3) % 3;
#pragma unroll
                    for (int qc = 0; qc < NLC; qc++) {
                        const int jcl = threadIdx.x + qc * NTHR;
#pragma unroll
                        for (int icl = 0; icl < TYC; icl++) if (jcl < CW) cl[slot][icl][jcl] = cnew[qc * TYC + icl];
                    }
                    kc_new = -2;
                }
            }
        }
    }

    if (MODE == MODE_RESNORM || MODE == MODE_RES32 || MODE == MODE_JNORM || MODE == MODE_CRES32) {
        __shared__ double red[16];
        double s = block_sum(acc, red);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = s;
    }
#undef ADDV
#undef ADD1
}

// sum `n` partials in a fixed order (one block) -> out[slot]
__global__ void __launch_bounds__(256) k_finish_sum(const double *partials, int n, double *out, int slot) {
    __shared__ double red[16];
    double s = 0.0;
    for (int q = threadIdx.x; q < n; q += 256) s += partials[q];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[slot] = s;
}
__global__ void __launch_bounds__(256) k_finish_max(const double *partials, int n, double *out, int slot) {
    __shared__ double red[16];
    double s = 0.0;
    for (int q = threadIdx.x; q < n; q += 256) s = fmax(s, partials[q]);
    s = block_max(s, red);
    if (threadIdx.x == 0) out[slot] = s;
}

// ------------------------------------------------------------------------------------------
// launch of the marching kernel
// ------------------------------------------------------------------------------------------
// per THREAD: a test or tool that forces a kernel form does not change what solver threads (loopback ranks) launch
thread_local int g_variant = -1, g_zchunk = -1;
extern "C" void mgk_set_tuning(int variant, int zchunk) { g_variant = variant; g_zchunk = zchunk; }

template <typename T, int DIM, int WX, int WY, int RY, int MODE>
static int launch_st(mgk_ctx *c, StArgs<T> &a, int nrows, hipStream_t s, int *nblocks_out) {
    constexpr int VX = 16 / sizeof(T);
    constexpr int TX = 64 * VX * WX, TY = (DIM == 3) ? WY * RY : 1;
    constexpr bool REDUCE = (MODE == MODE_RESNORM || MODE == MODE_RES32 || MODE == MODE_JNORM || MODE == MODE_CRES32);
    a.ntx = (a.nx + 1 + TX - 1) / TX;
    a.nty = (DIM == 3) ? (nrows + TY - 1) / TY : 1;
    long tiles = (long)a.ntx * a.nty;
    if (a.zend <= a.zbeg) { a.zbeg = 0; a.zend = a.nm; }
    const int nmr = a.zend - a.zbeg;      // planes marched by this launch
    int zc = g_zchunk;
    if (zc <= 0) {
        // Few, long streams: ~256-512 blocks, each marching a long run of planes over a full-row tile,
        // keep HBM pages open (measured at 1023^3: 256 blocks x 1023 planes 5.7 TB/s vs 4096 blocks 5.2).
        // The store-free residual-norm mode is latency bound instead and wants many short blocks;
        // so does the 2-D kernel (a tile is one row segment: little work per marching step).
        long nch = (MODE == MODE_RESNORM) ? (4096 + tiles - 1) / tiles
                 : (DIM == 2)             ? (1024 + tiles - 1) / tiles
                 : (tiles >= 256)         ? ((64 * WX * WY <= 256) ? 2 : 1)     // small blocks: two per CU (fp32 PJ 3.45 -> 2.96 ms)
                                          : (512 + tiles - 1) / tiles;
        zc = (int)((nmr + nch - 1) / nch);
        // 2-D: a marching step is one row segment, so a small level only fills the chip when it is cut into many short
        // chunks (measured at 255^2..1023^2: 16-row chunks left 32-128 blocks, 18-20 us per sweep; 4097^2 cycle 1.57 -> 1.12 ms)
        // (3-D likewise: 2-plane chunks on the cache-resident coarse levels, 255^3 cycle 1.09 -> 0.94 ms)
        const int zmin = (MODE == MODE_RESNORM && DIM == 3) ? 16 : 2;
        if (zc < zmin) zc = zmin;
        if (DIM == 3 && c->chunk_planes > 0 && zc > c->chunk_planes) zc = c->chunk_planes;     // slab ranks: never longer than the hint
    }
    if (zc > nmr) zc = nmr;
    a.zc = zc;
    long ntz = (nmr + zc - 1) / zc;
    long nblk = tiles * ntz;
    if (nblk > 0x7fffffffL) return fail(MGK_EINVAL, "stencil launch: too many blocks");
    if (REDUCE && nblk > c->max_partials) {
        // fewer, longer chunks so that the partial buffer suffices
        ntz = c->max_partials / tiles;
        if (ntz < 1) return fail(MGK_EINVAL, "stencil launch: partial buffer too small");
        zc = (int)((nmr + ntz - 1) / ntz);
        a.zc = zc;
        ntz = (nmr + zc - 1) / zc;
        nblk = tiles * ntz;
    }
    if (nblocks_out) *nblocks_out = (int)nblk;
    hipLaunchKernelGGL((k_stencil<T, DIM, WX, WY, RY, MODE>), dim3((unsigned)nblk), dim3(64 * WX * WY), 0, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// fp64 variants (tile = 128*WX x WY*RY)
template <int MODE>
static int dispatch_st(mgk_ctx *c, const mgk_geom *g, StArgs<double> &a, hipStream_t s, int *nblocks) {
    a.nx = g->nx;
    if (g->dim == 3) {
        a.ny = g->ny; a.nm = g->nz; a.rs = g->pitch; a.ms = g->plane;
        int v = g_variant >= 30 ? -1 : g_variant;
        if (v < 0) v = (MODE == MODE_RESNORM && g->nx >= 255) ? 3
                     : (g->nx >= 1023) ? ((MODE == MODE_PJACOBI || MODE == MODE_CRES32 || MODE == MODE_CHEBY) ? 9 : 12)   // on-the-fly corrections and the Chebyshev step's third operand: 512-thread blocks (a 1024-thread block is capped at 128 VGPRs and spills: Chebyshev step 9.88 -> 5.96 ms)
                     : (g->nx >= 511) ? (MODE == MODE_PJACOBI ? 13 : 6)             // fused prolongation at 511^3: 0.91 -> 0.73 ms
                     : (g->nx >= 255) ? 2 : (g->nx >= 127 ? 1 : 0);
        switch (v) {
            case 0: return launch_st<double, 3, 1, 2, 2, MODE>(c, a, g->ny, s, nblocks);   // 128 x 4, 128 thr
            case 1: return launch_st<double, 3, 1, 4, 2, MODE>(c, a, g->ny, s, nblocks);   // 128 x 8, 256 thr
            case 2: return launch_st<double, 3, 2, 2, 2, MODE>(c, a, g->ny, s, nblocks);   // 256 x 4, 256 thr
            case 3: return launch_st<double, 3, 2, 2, 4, MODE>(c, a, g->ny, s, nblocks);   // 256 x 8, 256 thr
            case 6: return launch_st<double, 3, 4, 2, 2, MODE>(c, a, g->ny, s, nblocks);   // 512 x 4, 512 thr
            case 9: return launch_st<double, 3, 8, 1, 4, MODE>(c, a, g->ny, s, nblocks);   // 1024 x 4, 512 thr
            case 12: return launch_st<double, 3, 8, 2, 2, MODE>(c, a, g->ny, s, nblocks);  // 1024 x 4, 1024 thr
            case 13: return launch_st<double, 3, 4, 1, 4, MODE>(c, a, g->ny, s, nblocks);  // 512 x 4, 256 thr
            default: return fail(MGK_EINVAL, "unknown 3-D stencil variant");
        }
    } else {
        a.ny = 1; a.nm = g->ny; a.rs = 0; a.ms = g->pitch;
        int v = g_variant >= 30 ? -1 : g_variant;
        if (v < 0) v = (g->nx >= 511) ? 2 : (g->nx >= 255 ? 1 : 0);
        switch (v) {
            case 0: return launch_st<double, 2, 1, 1, 1, MODE>(c, a, 1, s, nblocks);
            case 1: return launch_st<double, 2, 2, 1, 1, MODE>(c, a, 1, s, nblocks);
            case 2: return launch_st<double, 2, 4, 1, 1, MODE>(c, a, 1, s, nblocks);
            default: return fail(MGK_EINVAL, "unknown 2-D stencil variant");
        }
    }
}
// fp32 variants (tile = 256*WX x WY*RY; 3-D only: the mixed-precision inner cycle)
template <int MODE>
static int dispatch_st(mgk_ctx *c, const mgk_geom *g, StArgs<float> &a, hipStream_t s, int *nblocks) {
    a.nx = g->nx;
    if (g->dim != 3) return fail(MGK_EINVAL, "fp32 stencil kernels are built for 3-D only");
    a.ny = g->ny; a.nm = g->nz; a.rs = g->pitch; a.ms = g->plane;
    int v = g_variant >= 30 ? -1 : g_variant;
    if (v < 0) v = (g->nx >= 1023) ? (MODE == MODE_PJACOBI ? 3 : 2) : (g->nx >= 511) ? 1 : 0;
    switch (v) {
        case 0: return launch_st<float, 3, 1, 2, 2, MODE>(c, a, g->ny, s, nblocks);   // 256 x 4, 128 thr
        case 1: return launch_st<float, 3, 2, 2, 2, MODE>(c, a, g->ny, s, nblocks);   // 512 x 4, 256 thr
        case 2: return launch_st<float, 3, 4, 2, 2, MODE>(c, a, g->ny, s, nblocks);   // 1024 x 4, 512 thr
        case 3: return launch_st<float, 3, 4, 1, 4, MODE>(c, a, g->ny, s, nblocks);   // 1024 x 4, 256 thr
        default: return fail(MGK_EINVAL, "unknown fp32 stencil variant");
    }
}

template <typename T>
static void set_coef(StArgs<T> &a, const mgk_geom *g, const double *coef) {
    if (g->dim == 3) {
        a.a0 = (T)coef[0]; a.a1 = (T)coef[1]; a.a2 = (T)coef[2]; a.a3 = (T)coef[3]; a.a4 = (T)coef[4]; a.a5 = (T)coef[5]; a.a6 = (T)coef[6];
    } else {
        a.a0 = (T)coef[0]; a.a1 = (T)0; a.a2 = (T)coef[1]; a.a3 = (T)coef[2]; a.a4 = (T)coef[3]; a.a5 = (T)0; a.a6 = (T)coef[4];
    }
}

// register / shuffle form of the plain sweeps (k_jrow, defined with the other row kernels further down)
template <typename T> static bool jrow_ok(const mgk_geom *g);
template <typename T, bool NORM>
static int launch_jrow(mgk_ctx *c, const mgk_geom *g, const StArgs<T> &a, int zbeg, int zend, hipStream_t s, double *partials, int max_partials, int *nparts);

extern "C" int mgk_jacobi_range_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                    const double *b, const double *u, double *unew, int zbeg, int zend, void *stream) {
    if (!c || !g || !coef || !b || !u || !unew || u == unew) return fail(MGK_EINVAL, "mgk_jacobi_f64: bad arguments");
    const int nm = (g->dim == 3) ? g->nz : g->ny;
    if (zbeg < 0 || zend > nm || zbeg >= zend) return fail(MGK_EINVAL, "mgk_jacobi_range_f64: empty or out-of-range plane range");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org;
    set_coef(a, g, coef); a.dinv = dinv; a.scale = scale;
    a.zbeg = zbeg; a.zend = zend;
    if (jrow_ok<double>(g)) return launch_jrow<double, false>(c, g, a, zbeg, zend, S(c, stream), nullptr, 0, nullptr);
    return dispatch_st<MODE_JACOBI>(c, g, a, S(c, stream), nullptr);
}
extern "C" int mgk_jacobi_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                              const double *b, const double *u, double *unew, void *stream) {
    if (!g) return fail(MGK_EINVAL, "mgk_jacobi_f64: bad arguments");
    return mgk_jacobi_range_f64(c, g, coef, dinv, scale, b, u, unew, 0, (g->dim == 3) ? g->nz : g->ny, stream);
}

extern "C" int mgk_cheby_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv,
                             double c_km1, double c_k, double c_z,
                             const double *b, const double *pk, const double *pkm1, double *pkp1, void *stream) {
    if (!c || !g || !coef || !b || !pk || !pkm1 || !pkp1 || pk == pkp1 || pkm1 == pkp1)
        return fail(MGK_EINVAL, "mgk_cheby_f64: bad arguments");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = pk + g->org; a.b = b + g->org; a.aux = pkm1 + g->org; a.out = pkp1 + g->org;
    set_coef(a, g, coef); a.dinv = dinv; a.ckm1 = c_km1; a.ck = c_k; a.cz = c_z;
    return dispatch_st<MODE_CHEBY>(c, g, a, S(c, stream), nullptr);
}

extern "C" int mgk_residual_range_f64(mgk_ctx *c, const mgk_geom *g, const double *coef,
                                      const double *b, const double *u, double *r, int zbeg, int zend, void *stream) {
    if (!c || !g || !coef || !b || !u || !r || u == r) return fail(MGK_EINVAL, "mgk_residual_f64: bad arguments");
    const int nm = (g->dim == 3) ? g->nz : g->ny;
    if (zbeg < 0 || zend > nm || zbeg >= zend) return fail(MGK_EINVAL, "mgk_residual_range_f64: empty or out-of-range plane range");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = r + g->org;
    set_coef(a, g, coef);
    a.zbeg = zbeg; a.zend = zend;
    return dispatch_st<MODE_RESIDUAL>(c, g, a, S(c, stream), nullptr);
}
extern "C" int mgk_residual_f64(mgk_ctx *c, const mgk_geom *g, const double *coef,
                                const double *b, const double *u, double *r, void *stream) {
    if (!g) return fail(MGK_EINVAL, "mgk_residual_f64: bad arguments");
    return mgk_residual_range_f64(c, g, coef, b, u, r, 0, (g->dim == 3) ? g->nz : g->ny, stream);
}

extern "C" int mgk_apply_f64(mgk_ctx *c, const mgk_geom *g, const double *coef,
                             const double *x, double *y, void *stream) {
    if (!c || !g || !coef || !x || !y || x == y) return fail(MGK_EINVAL, "mgk_apply_f64: bad arguments");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = x + g->org; a.b = x + g->org; a.out = y + g->org;      // b is loaded but unused in this mode
    set_coef(a, g, coef);
    return dispatch_st<MODE_APPLY>(c, g, a, S(c, stream), nullptr);
}

// mgk_defer_result(ctx, slot): until reset with slot = NULL, every single-value reduction (sum of squares, dot) writes its
// result to the device double `slot` in stream order and returns 0.0 to the host WITHOUT synchronising; the caller reads the
// slots later in one copy.  Used by the fixed-count cycling loop, which needs no norm on the host between cycles.
extern "C" int mgk_defer_result(mgk_ctx *c, double *slot_dev) {
    if (!c) return fail(MGK_EINVAL, "mgk_defer_result: null context");
    c->defer_slot = slot_dev;
    return 0;
}
int finish_to_host(mgk_ctx *c, int nparts, int nslots, hipStream_t s, double *host_out) {
    if (c->defer_slot && nslots == 1) {
        hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(256), 0, s, c->partials, nparts, c->defer_slot, 0);
        HIPCHK(hipGetLastError());
        host_out[0] = 0.0;
        return 0;
    }
    for (int q = 0; q < nslots; q++)
        hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(256), 0, s, c->partials + (long)q * c->max_partials, nparts, c->result_dev, q);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->result_host, c->result_dev, sizeof(double) * nslots, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int q = 0; q < nslots; q++) host_out[q] = c->result_host[q];
    return 0;
}

extern "C" int mgk_residual_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef,
                                      const double *b, const double *u, double *sumsq_host, void *stream) {
    if (!c || !g || !coef || !b || !u || !sumsq_host) return fail(MGK_EINVAL, "mgk_residual_sumsq_f64: bad arguments");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.partials = c->partials;
    set_coef(a, g, coef);
    int nblk = 0;
    int rc = dispatch_st<MODE_RESNORM>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}

// Jacobi sweep of u that also returns ||b - A u||^2: the residual norm that closes cycle k (src/solver.c:1545-1546)
// and the first pre-smoothing sweep of cycle k+1 (:1531) read the same u and b and form the same residual.
extern "C" int mgk_jacobi_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                    const double *b, const double *u, double *unew, double *sumsq_host, void *stream) {
    if (!c || !g || !coef || !b || !u || !unew || u == unew || !sumsq_host) return fail(MGK_EINVAL, "mgk_jacobi_sumsq_f64: bad arguments");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org; a.partials = c->partials;
    set_coef(a, g, coef); a.dinv = dinv; a.scale = scale;
    int nblk = 0;
    int rc = jrow_ok<double>(g) ? launch_jrow<double, true>(c, g, a, 0, g->nz, S(c, stream), c->partials, c->max_partials, &nblk)
                                : dispatch_st<MODE_JNORM>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}

// The drop-in's pass for KSPBuildResidual + VecNorm + the first sweep of the KSPSolve that follows (src/solver.c:1545-1546,1531): one read of
// u and b yields r = b - A u (stored), sum r^2, and unew = u + scale*(r*dinv) -- 32 B per unknown instead of 24 + 8 + 24.  2-D (the PETSc
// surface is 2-D), constant coefficients or row tables.
extern "C" int mgk_jacobi_sumsq_store_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                                          const double *b, const double *u, double *unew, double *r, double *sumsq_host, void *stream) {
    if (!c || !g || g->dim != 2 || (!coef && !ctab) || (ctab && !dtab) || !b || !u || !unew || !r || u == unew || u == r || b == r || r == unew || b == unew || !sumsq_host)
        return fail(MGK_EINVAL, "mgk_jacobi_sumsq_store_f64: bad arguments (2-D)");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org; a.out2 = r + g->org; a.partials = c->partials;
    if (coef) set_coef(a, g, coef);
    a.dinv = ctab ? 1.0 : dinv; a.scale = scale; a.ctab = ctab; a.dtab = dtab;
    int nblk = 0;
    int rc = dispatch_st<MODE_JNORM>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}

// The same on the marching planes [zbeg, zend) only, block partials deposited from slot `part_off` on and NOT reduced:
// a slab rank sweeps its interior planes while the ghost planes are still travelling, then the two boundary planes, and
// closes with one mgk_partials_finish over all the slots (fixed order: deterministic).
extern "C" int mgk_jacobi_sumsq_range_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                          const double *b, const double *u, double *unew, int zbeg, int zend,
                                          int part_off, int *nparts, void *stream) {
    if (!c || !g || !coef || !b || !u || !unew || u == unew || !nparts) return fail(MGK_EINVAL, "mgk_jacobi_sumsq_range_f64: bad arguments");
    const int nm = (g->dim == 3) ? g->nz : g->ny;
    if (zbeg < 0 || zend > nm || zbeg >= zend || part_off < 0 || part_off >= c->max_partials)
        return fail(MGK_EINVAL, "mgk_jacobi_sumsq_range_f64: empty or out-of-range plane range / partial offset");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org; a.partials = c->partials + part_off;
    set_coef(a, g, coef); a.dinv = dinv; a.scale = scale;
    a.zbeg = zbeg; a.zend = zend;
    // launch_st keeps the block count within max_partials; the offset must leave that much room
    mgk_ctx lim = *c;
    lim.max_partials = c->max_partials - part_off;
    int nblk = 0;
    int rc = jrow_ok<double>(g) ? launch_jrow<double, true>(c, g, a, zbeg, zend, S(c, stream), c->partials + part_off, lim.max_partials, &nblk)
                                : dispatch_st<MODE_JNORM>(&lim, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    *nparts = nblk;
    return 0;
}
extern "C" int mgk_partials_finish(mgk_ctx *c, int nparts, double *sumsq_host, void *stream) {
    if (!c || nparts < 1 || nparts > c->max_partials || !sumsq_host) return fail(MGK_EINVAL, "mgk_partials_finish: bad arguments");
    return finish_to_host(c, nparts, 1, S(c, stream), sumsq_host);
}

// ------------------------------------------------------------------------------------------
// two Jacobi sweeps in one pass (temporal blocking): out = J(J(u)), 24 B + halo re-reads instead of 48 B per unknown
// ------------------------------------------------------------------------------------------
// Block = one full-row tile of TY = 4 rows marching along z, 64*WX lanes, lane t owns the column pair x = 2t, 2t+1 of ALL
// rows of the tile.  Stage 1 (first sweep) is evaluated on the tile plus one halo row either side (6 rows) for plane
// z+1 from u rows yb-2 .. yb+5 held in registers (planes z, z+1, z+2, and z+3 in flight): y neighbours are the lane's own
// registers, x neighbours come by wave shuffle (the two edge lanes of a wave: from a small LDS edge buffer), so stage 1
// needs no barrier.  Its output u' goes to an LDS ring of three planes; stage 2 (second sweep) reads the seven u'
// neighbours of plane z from the ring and stores the result.  The halo rows of u' are recomputed by the neighbouring
// tile (1.5x stage-1 arithmetic, 2x u row loads that mostly hit L2); u' never goes to memory.  Each stage performs
// exactly the arithmetic of k_stencil<MODE_JACOBI>, so the result equals two separate sweeps bit for bit.
// one Jacobi update of a lane vector: o = c + scale * ((b - A.(dn, s, w, c, e, n, up)) * dinv), terms in the canonical order.
// The float overload works on the 4-wide vector type so that the compiler issues packed fp32 instructions (v_pk_mul_f32 /
// v_pk_add_f32: two lanes of the lane vector per instruction); every lane sees the same IEEE operations in the same order.

template <typename T>
__device__ __forceinline__ V16<T> jac7(T a0, T a1, T a2, T a3, T a4, T a5, T a6, T dinv, T scale, const V16<T> &dn, const V16<T> &sv,
                                       const V16<T> &c, const V16<T> &nv, const V16<T> &upv, T Wv, T Ev, const V16<T> &b) {
    constexpr int VX = 16 / sizeof(T);
    V16<T> o;
#pragma unroll
    for (int e = 0; e < VX; e++) {
        const T wv = (e == 0) ? Wv : c.v[e - 1 < 0 ? 0 : e - 1];
        const T ev = (e == VX - 1) ? Ev : c.v[e + 1 > VX - 1 ? VX - 1 : e + 1];
        T s = a0 * dn.v[e];
        s = s + a1 * sv.v[e];
        s = s + a2 * wv;
        s = s + a3 * c.v[e];
        s = s + a4 * ev;
        s = s + a5 * nv.v[e];
        s = s + a6 * upv.v[e];
        const T res = b.v[e] - s;
        const T zz = res * dinv;
        o.v[e] = c.v[e] + scale * zz;
    }
    return o;
}
__device__ __forceinline__ V16<float> jac7(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float dinv, float scale,
                                           const V16<float> &dn, const V16<float> &sv, const V16<float> &c, const V16<float> &nv,
                                           const V16<float> &upv, float Wv, float Ev, const V16<float> &b) {
    auto ld = [](const V16<float> &x) { f4v r; r.x = x.v[0]; r.y = x.v[1]; r.z = x.v[2]; r.w = x.v[3]; return r; };
    const f4v C = ld(c);
    f4v W; W.x = Wv; W.y = c.v[0]; W.z = c.v[1]; W.w = c.v[2];
    f4v E; E.x = c.v[1]; E.y = c.v[2]; E.z = c.v[3]; E.w = Ev;
    f4v s = a0 * ld(dn);
    s = s + a1 * ld(sv);
    s = s + a2 * W;
    s = s + a3 * C;
    s = s + a4 * E;
    s = s + a5 * ld(nv);
    s = s + a6 * ld(upv);
    const f4v res = ld(b) - s;
    const f4v zz = res * dinv;
    const f4v o4 = C + scale * zz;
    V16<float> o; o.v[0] = o4.x; o.v[1] = o4.y; o.v[2] = o4.z; o.v[3] = o4.w;
    return o;
}

// row loads of the register / shuffle kernels: wave-uniform row base + a 32-bit lane offset, unconditional
template <typename T>
__device__ __forceinline__ V16<T> ldrow(const T *row_uniform, unsigned lane_bytes) {
    return *reinterpret_cast<const V16<T> *>(reinterpret_cast<const char *>(row_uniform) + lane_bytes);
}
__device__ __forceinline__ V16<double> ldrow_stream(const double *row_uniform, unsigned lane_bytes) {
    return ldv_stream(reinterpret_cast<const double *>(reinterpret_cast<const char *>(row_uniform) + lane_bytes), true);
}
__device__ __forceinline__ V16<float> ldrow_stream(const float *row_uniform, unsigned lane_bytes) {
    return ldv_stream(reinterpret_cast<const float *>(reinterpret_cast<const char *>(row_uniform) + lane_bytes), true);
}

template <typename T>
struct J2Args {
    const T *u, *b;
    T *out;
    int nx, ny, nz;
    long rs, ms;
    int nty, zc;
    T a0, a1, a2, a3, a4, a5, a6, dinv, scale;
    // z-slab of a multi-GPU run: the second sweep of planes 0 / nz-1 needs the first sweep of the neighbour's last / first
    // plane, hence TWO of its planes of u (plane -1 / nz is the field's own ghost plane, plane -2 / nz+1 comes in these
    // separate plane buffers, same pitch, pointing at their interior origin) and b on the ghost planes.
    const T *far_lo, *far_hi;
    int has_lo, has_hi;
    int zbeg, zend;          // output planes of this launch (whole grid / slab: 0, nz)
    double *partials;        // k_jacobi2r<.., NORM = true>: one partial sum of squares per block (|| b - A u ||^2 of the INPUT field)
};

// ZG: THREE sweeps from a zero initial guess.  The first one is pointwise, u1 = scale * (b * dinv) (k_jacobi_zero), so the pass reads
// b alone -- at the rows and planes where it would read u -- and forms u1 on the way: 8 + 8 B per unknown (fp32: 4 + 4) instead of
// 8 written by the zero-guess sweep + 24 for the two-sweep pass.  a.u is not read.
template <typename T, int WX, int FORM, bool ZG = false>
__global__ void __launch_bounds__(64 * WX) k_jacobi2(const J2Args<T> a) {
    constexpr bool UNC = (FORM & 1) != 0, DPP = (FORM & 2) != 0;      // as in k_jacobi2r
    constexpr int VX = 16 / sizeof(T), TY = 4, R1 = TY + 4, R2 = TY + 2, TX = 64 * VX * WX, LW = TX + 2 * VX;
    __shared__ __attribute__((aligned(16))) T ring[3][R2][LW];
    __shared__ T edgeW[2][R2][WX], edgeE[2][R2][WX];          // first / last element of every wave's row segment
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int yb = TY * ty;
    const int z0 = a.zbeg + tz * a.zc, z1 = min(z0 + a.zc, a.zend);
    if (z0 >= z1) return;
    const int xl = VX * tid, x0 = xl;
    const bool xok = x0 < a.nx;
    const bool lastvec = (x0 + VX > a.nx);

    bool uok[R1], s1ok[R2];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) { const int y = yb - 2 + rr; uok[rr] = xok && y >= -1 && y <= a.ny; }
#pragma unroll
    for (int q = 0; q < R2; q++) { const int y = yb - 1 + q; s1ok[q] = y >= 0 && y < a.ny; }
    const long uoff = (long)(yb - 2) * a.rs + x0;             // row rr of plane p: uplane(p) + rr*rs
    const T *bp_ = a.b + (long)(yb - 1) * a.rs + x0;          // row q  of plane p: bp_ + p*ms + q*rs
    const int pmin = a.has_lo ? -2 : -1, pmax = a.has_hi ? a.nz + 1 : a.nz;     // u planes that exist
    const int smin = a.has_lo ? -1 : 0, smax = a.has_hi ? a.nz : a.nz - 1;      // planes on which the first sweep is real
    const T *usrc = ZG ? a.b : a.u;
    auto uplane = [&](int p) -> const T * {
        return (p == -2 ? a.far_lo : p == a.nz + 1 ? a.far_hi : usrc + (long)p * a.ms) + uoff;
    };
    auto first = [&](VT v) -> VT {                            // ZG: the zero-guess sweep of what was loaded (zeros stay zeros)
        if (ZG) {
#pragma unroll
            for (int e = 0; e < VX; e++) { const T zx = v.v[e] * a.dinv; v.v[e] = a.scale * zx; }
        }
        return v;
    };

    // zero the ring (u'(-1) and the padding columns are 0) -- one pass, strided over the block
    for (int i = tid; i < 3 * R2 * LW; i += 64 * WX) (&ring[0][0][0])[i] = (T)0;
    const unsigned lb = (unsigned)(x0 * (int)sizeof(T));
    long uro[R1], bro[R2];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) uro[rr] = (long)max(-1, min(yb - 2 + rr, a.ny)) * a.rs;
#pragma unroll
    for (int q = 0; q < R2; q++) bro[q] = (long)max(0, min(yb - 1 + q, a.ny)) * a.rs;
    auto LDU = [&](int p, int rr, bool pv) -> VT {
        if (UNC) {
            const int pp = max(pmin, min(p, pmax));
            const T *pl = (pp == -2) ? a.far_lo : (pp == a.nz + 1) ? a.far_hi : usrc + (long)pp * a.ms;
            return first(*reinterpret_cast<const VT *>(reinterpret_cast<const char *>(pl + uro[rr]) + lb));
        }
        return first(ldv(uplane(p) + (long)rr * a.rs, uok[rr] && pv));
    };
    auto LDB = [&](int p, int q, bool pv, bool stream) -> VT {
        if (UNC) {
            const T *pl = a.b + (long)max(smin, min(p, smax)) * a.ms + bro[q];
            const T *ad = reinterpret_cast<const T *>(reinterpret_cast<const char *>(pl) + lb);
            return stream ? ldv_stream(ad, true) : ldv(ad, true);
        }
        const T *bp = bp_ + (long)p * a.ms + (long)q * a.rs;
        const bool ok = xok && s1ok[q] && pv;
        return stream ? ldv_stream(bp, ok) : ldv(bp, ok);
    };

    VT ua[R1], ub[R1], uc[R1], ud[R1], b1[R2], bn[R2], b0[TY];
    const int t0 = z0 - 2;                                    // first step: stage 1 of plane z0 - 1
#pragma unroll
    for (int rr = 0; rr < R1; rr++) {
        const long ro = (long)rr * a.rs;
        (void)ro;
        ua[rr] = LDU(t0, rr, t0 >= pmin);
        ub[rr] = LDU(t0 + 1, rr, t0 + 1 >= pmin);
        uc[rr] = LDU(t0 + 2, rr, true);
        ud[rr] = v16_zero<T>();
    }
#pragma unroll
    for (int q = 0; q < R2; q++) {
        const int p = t0 + 1;
        b1[q] = LDB(p, q, p >= smin && p <= smax, false);
        bn[q] = v16_zero<T>();
    }
#pragma unroll
    for (int j = 0; j < TY; j++) b0[j] = v16_zero<T>();
    // edges of the first centre plane (ub = plane t0 + 1)
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int q = 0; q < R2; q++) {
            if (lane == 0) edgeW[(t0 + 1) & 1][q][w] = ub[q + 1].v[0];
            else edgeE[(t0 + 1) & 1][q][w] = ub[q + 1].v[VX - 1];
        }
    }
    __syncthreads();

    for (int t = t0; t < z1; t++) {
        const int p = t + 1;                                  // plane of stage 1
        // ---- loads consumed in the next step ----
        {
            const bool pu = (t + 3 <= pmax), pb = (t + 2 <= smax) && (t + 1 < z1);
#pragma unroll
            for (int rr = 0; rr < R1; rr++) ud[rr] = LDU(t + 3, rr, pu);
#pragma unroll
            for (int q = 0; q < R2; q++) bn[q] = LDB(t + 2, q, pb, q >= 2 && q < R2 - 2);     // rows shared with neighbours: cached
        }
        // ---- stage 1: u'(p) on rows yb-1 .. yb+TY ----
        {
            const bool pin = (p >= smin && p <= smax);
            const int slot = (p + 3) % 3, eb = p & 1;
#pragma unroll
            for (int q = 0; q < R2; q++) {
                const int rr = q + 1;
                T Wv = lane_up<DPP>(ub[rr].v[VX - 1]), Ev = lane_dn<DPP>(ub[rr].v[0]);
                if (lane == 0) Wv = (w > 0) ? edgeE[eb][q][w - 1] : (T)0;
                if (lane == 63) Ev = (w < WX - 1) ? edgeW[eb][q][w + 1] : (T)0;
                VT o = jac7(a.a0, a.a1, a.a2, a.a3, a.a4, a.a5, a.a6, a.dinv, a.scale, ua[rr], ub[rr - 1], ub[rr], ub[rr + 1], uc[rr], Wv, Ev, b1[q]);
#pragma unroll
                for (int e = 0; e < VX; e++)
                    if (!pin || !s1ok[q] || !xok || (lastvec && x0 + e >= a.nx)) o.v[e] = (T)0;
                *reinterpret_cast<VT *>(&ring[slot][q][xl + VX]) = o;
            }
            // edges of the next centre plane (uc = plane p + 1), other buffer
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int q = 0; q < R2; q++) {
                    if (lane == 0) edgeW[eb ^ 1][q][w] = uc[q + 1].v[0];
                    else edgeE[eb ^ 1][q][w] = uc[q + 1].v[VX - 1];
                }
            }
        }
        __syncthreads();                                      // u'(p) complete
        // ---- stage 2: out(t) on rows yb .. yb+TY-1 from u'(t-1), u'(t), u'(t+1) ----
        if (t >= z0) {
            const int sm = (t + 2) % 3, sc = t % 3, sp = (t + 1) % 3;
#pragma unroll
            for (int j = 0; j < TY; j++) {
                const int q = j + 1;
                const VT c = *reinterpret_cast<const VT *>(&ring[sc][q][xl + VX]);
                const VT dn = *reinterpret_cast<const VT *>(&ring[sm][q][xl + VX]);
                const VT upv = *reinterpret_cast<const VT *>(&ring[sp][q][xl + VX]);
                const VT sv = *reinterpret_cast<const VT *>(&ring[sc][q - 1][xl + VX]);
                const VT nv = *reinterpret_cast<const VT *>(&ring[sc][q + 1][xl + VX]);
                const T Wv = ring[sc][q][xl + VX - 1], Ev = ring[sc][q][xl + 2 * VX];
                VT o = jac7(a.a0, a.a1, a.a2, a.a3, a.a4, a.a5, a.a6, a.dinv, a.scale, dn, sv, c, nv, upv, Wv, Ev, b0[j]);
                if (lastvec) {
#pragma unroll
                    for (int e = 0; e < VX; e++) if (x0 + e >= a.nx) o.v[e] = (T)0;
                }
                if (xok && yb + j < a.ny) stv_stream(a.out + (long)t * a.ms + (long)(yb + j) * a.rs + x0, o);
            }
        }
        __syncthreads();                                      // ring slot (t-1)%3 is free for u'(t+2)
        // ---- rotate ----
#pragma unroll
        for (int j = 0; j < TY; j++) b0[j] = b1[j + 1];
#pragma unroll
        for (int q = 0; q < R2; q++) b1[q] = bn[q];
#pragma unroll
        for (int rr = 0; rr < R1; rr++) { ua[rr] = ub[rr]; ub[rr] = uc[rr]; uc[rr] = ud[rr]; }
    }
}

// k_jacobi2 on full-row shapes with the instruction overhead taken out (fp32 1023^3: the LDS-ring form issued 725 vector instructions per
// step of which 160 register copies of the plane rotation, 86 selects, 57 SGPR reloads and 19 64-bit address adds; two waves per SIMD were
// issue bound): buffer-descriptor loads / stores with 32-bit scalar row offsets, wave-edge neighbours as the DPP `old` operand (zero pads
// at the ends of the row), rows / planes outside the grid skipped by wave-uniform branches, and the marching loop unrolled by four with the
// roles of the four register planes permuted instead of copied.  Same arithmetic (jac7), same results.
template <typename T, int WX, bool ZG>
__global__ void __launch_bounds__(64 * WX) k_jacobi2b(const J2Args<T> a) {
    constexpr int VX = 16 / sizeof(T), TY = 4, R1 = TY + 4, R2 = TY + 2, TX = 64 * VX * WX, LW = TX + 2 * VX;
    __shared__ __attribute__((aligned(16))) T ring[3][R2][LW];
    __shared__ T edgeW[2][R2][WX + 1], edgeE[2][R2][WX + 1];      // [.][.][0] of edgeE and [.][.][WX] of edgeW: zero pads
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int yb = TY * ty;
    const int z0 = a.zbeg + tz * a.zc, z1 = min(z0 + a.zc, a.zend);
    if (z0 >= z1) return;
    const int xl = VX * tid;
    const bool lastlane = (tid == 64 * WX - 1);                   // its last element is the ghost column x = nx: stays 0
    const unsigned lb = (unsigned)(xl * (int)sizeof(T));
    const int pmin = a.has_lo ? -2 : -1, pmax = a.has_hi ? a.nz + 1 : a.nz;     // u planes that exist
    const int smin = a.has_lo ? -1 : 0, smax = a.has_hi ? a.nz : a.nz - 1;      // planes on which the first sweep is real
    const int qlo = max(0, 1 - yb), qhi = min(R2 - 1, a.ny - yb);               // first-sweep rows q (y = yb-1+q) inside the grid
    const unsigned rowb = (unsigned)(a.rs * (long)sizeof(T));
    const unsigned plane_bytes = (unsigned)(a.ny + 2) * rowb;
    unsigned urb[R1], brb[R2];                                    // byte offsets of the rows from row -1 of a plane (clamped to the ghost rows)
#pragma unroll
    for (int rr = 0; rr < R1; rr++) urb[rr] = (unsigned)(max(-1, min(yb - 2 + rr, a.ny)) + 1) * rowb;
#pragma unroll
    for (int q = 0; q < R2; q++) brb[q] = (unsigned)(max(0, min(yb - 1 + q, a.ny)) + 1) * rowb;
    const T *usrc = ZG ? a.b : a.u;
    auto URS = [&](int p) {
        const int pp = max(pmin, min(p, pmax));
        const T *pl = (pp == -2) ? a.far_lo : (pp == a.nz + 1) ? a.far_hi : usrc + (long)pp * a.ms;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(pl - a.rs), 0, (int)plane_bytes, 0x00020000);
    };
    auto BRS = [&](int p) { return __builtin_amdgcn_make_buffer_rsrc((void *)(a.b - a.rs + (long)max(smin, min(p, smax)) * a.ms), 0, (int)plane_bytes, 0x00020000); };
    auto first = [&](VT v) -> VT {                                // ZG: the zero-guess sweep of what was loaded (zeros stay zeros)
        if (ZG) {
#pragma unroll
            for (int e = 0; e < VX; e++) { const T zx = v.v[e] * a.dinv; v.v[e] = a.scale * zx; }
        }
        return v;
    };
    for (int i = tid; i < 3 * R2 * LW; i += 64 * WX) (&ring[0][0][0])[i] = (T)0;
    for (int i = tid; i < 2 * R2 * (WX + 1); i += 64 * WX) { (&edgeW[0][0][0])[i] = (T)0; (&edgeE[0][0][0])[i] = (T)0; }

    VT ua[R1], ub[R1], uc[R1], ud[R1], b1[R2], bn[R2], b0[TY];
    const int t0 = z0 - 2;                                        // first step: stage 1 of plane z0 - 1
    {
        const auto r0 = URS(t0), r1 = URS(t0 + 1), r2 = URS(t0 + 2);
        const auto rb = BRS(t0 + 1);
#pragma unroll
        for (int rr = 0; rr < R1; rr++) { ua[rr] = first(bufld<T>(r0, lb, urb[rr])); ub[rr] = first(bufld<T>(r1, lb, urb[rr])); uc[rr] = first(bufld<T>(r2, lb, urb[rr])); ud[rr] = v16_zero<T>(); }
#pragma unroll
        for (int q = 0; q < R2; q++) { b1[q] = bufld<T>(rb, lb, brb[q]); bn[q] = v16_zero<T>(); }
    }
#pragma unroll
    for (int j = 0; j < TY; j++) b0[j] = v16_zero<T>();
    __syncthreads();                                              // the zero fill is complete before the first edge values land
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int q = 0; q < R2; q++) {
            if (lane == 0) edgeW[(t0 + 1) & 1][q][w] = ub[q + 1].v[0];
            else edgeE[(t0 + 1) & 1][q][w + 1] = ub[q + 1].v[VX - 1];
        }
    }
    __syncthreads();

    // one marching step; A: plane t, B: t+1, C: t+2 of u, D receives plane t+3
    auto step = [&](VT (&A)[R1], VT (&B)[R1], VT (&C)[R1], VT (&D)[R1], const int t) __attribute__((always_inline)) {
        const int p = t + 1;                                      // plane of stage 1
        {
            const auto r3 = URS(t + 3);
            const auto rb = BRS(t + 2);
#pragma unroll
            for (int rr = 0; rr < R1; rr++) D[rr] = bufld<T>(r3, lb, urb[rr]);
#pragma unroll
            for (int q = 0; q < R2; q++) bn[q] = (q >= 2 && q < R2 - 2) ? bufld_nt<T>(rb, lb, brb[q]) : bufld<T>(rb, lb, brb[q]);     // rows shared with neighbours: cached
        }
        // ---- stage 1: u'(p) on rows yb-1 .. yb+TY ----
        {
            const bool pin = (p >= smin && p <= smax);
            const int slot = (p + 3) % 3, eb = p & 1;
            const int qlo_t = pin ? qlo : R2;                     // (loop-variant bounds: planes outside the grid have no valid row)
            const unsigned span_t = (unsigned)(qhi - qlo_t);
#pragma unroll
            for (int q = 0; q < R2; q++) {
                const int rr = q + 1;
                if ((unsigned)(q - qlo_t) <= span_t && qhi >= qlo_t) {          // wave-uniform: the row is inside the grid
                    const T Wv = lane_up_old(B[rr].v[VX - 1], edgeE[eb][q][w]);
                    const T Ev = lane_dn_old(B[rr].v[0], edgeW[eb][q][w + 1]);
                    VT o = jac7(a.a0, a.a1, a.a2, a.a3, a.a4, a.a5, a.a6, a.dinv, a.scale, A[rr], B[rr - 1], B[rr], B[rr + 1], C[rr], Wv, Ev, b1[q]);
                    if (lastlane) o.v[VX - 1] = (T)0;
                    *reinterpret_cast<VT *>(&ring[slot][q][xl + VX]) = o;
                } else if (!pin) {
                    *reinterpret_cast<VT *>(&ring[slot][q][xl + VX]) = v16_zero<T>();   // plane outside the grid: zeros (rows outside keep their initial zeros)
                }
            }
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int q = 0; q < R2; q++) {
                    if (lane == 0) edgeW[eb ^ 1][q][w] = C[q + 1].v[0];
                    else edgeE[eb ^ 1][q][w + 1] = C[q + 1].v[VX - 1];
                }
            }
        }
        __syncthreads();                                          // u'(p) complete
        // ---- stage 2: out(t) on rows yb .. yb+TY-1 from u'(t-1), u'(t), u'(t+1) ----
        if (t >= z0) {
            const int sm = (t + 2) % 3, sc = t % 3, sp = (t + 1) % 3;
            const auto ro = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out - a.rs + (long)t * a.ms), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < TY; j++) {
                const int q = j + 1;
                if (yb + j < a.ny) {                                            // wave-uniform
                    const VT c = *reinterpret_cast<const VT *>(&ring[sc][q][xl + VX]);
                    const VT dn = *reinterpret_cast<const VT *>(&ring[sm][q][xl + VX]);
                    const VT upv = *reinterpret_cast<const VT *>(&ring[sp][q][xl + VX]);
                    const VT sv = *reinterpret_cast<const VT *>(&ring[sc][q - 1][xl + VX]);
                    const VT nv = *reinterpret_cast<const VT *>(&ring[sc][q + 1][xl + VX]);
                    const T Wv = ring[sc][q][xl + VX - 1], Ev = ring[sc][q][xl + 2 * VX];
                    VT o = jac7(a.a0, a.a1, a.a2, a.a3, a.a4, a.a5, a.a6, a.dinv, a.scale, dn, sv, c, nv, upv, Wv, Ev, b0[j]);
                    if (lastlane) o.v[VX - 1] = (T)0;
                    bufst_nt<T>(o, ro, lb, (unsigned)(yb + j + 1) * rowb);
                }
            }
        }
        __syncthreads();                                          // ring slot (t-1)%3 is free for u'(t+2)
#pragma unroll
        for (int j = 0; j < TY; j++) b0[j] = b1[j + 1];
#pragma unroll
        for (int q = 0; q < R2; q++) b1[q] = bn[q];
#pragma unroll
        for (int rr = 0; rr < R1; rr++) D[rr] = first(D[rr]);     // (ZG: the plane that arrived becomes the zero-guess sweep of b; loads are awaited here)
    };
    for (int t = t0; t < z1; t += 4) {
        step(ua, ub, uc, ud, t);
        if (t + 1 < z1) step(ub, uc, ud, ua, t + 1);
        if (t + 2 < z1) step(uc, ud, ua, ub, t + 2);
        if (t + 3 < z1) step(ud, ua, ub, uc, t + 3);
    }
}

// Variant of k_jacobi2 with ONE barrier per marching step: the first-sweep values a lane needs again at its own columns
// (planes t-1 and t+1 of the second sweep) stay in its registers; LDS holds only the centre plane u'(t) that the x / y
// neighbours of the second sweep are read from, double buffered (written in step t-1, read in step t while u'(t+1) goes
// to the other buffer).  The plane of u brought in for the next step is loaded straight into the registers of the plane
// the first sweep has just finished with.
// NORM: the first sweep forms b - A u of the input field on the way; its sum of squares over the points this block owns goes to
// partials[blockIdx.x] -- the residual norm that closes cycle k (src/solver.c:1545-1546) out of the pass that makes the first TWO
// pre-smoothing sweeps of cycle k+1 (:1531).
// NORM == 2: the sum of squares of the residual the SECOND sweep forms, i.e. of b - A J(u): the norm that closes a cycle whose last
// post-smoothing sweep is this pass's first sweep (the iterate J(u) itself is never stored: the caller keeps u and owes one sweep).
template <typename T, int WX, int FORM, int NORM = 0>
__global__ void __launch_bounds__(64 * WX) k_jacobi2r(const J2Args<T> a) {
    // FORM bit 0: unconditional loads with rows / planes clamped on the scalar unit (full-row shapes only: no lane outside the
    // grid; whatever a clamped load brings in only reaches first-sweep values that are forced to 0 anyway); bit 1: DPP lane shifts
    constexpr bool UNC = (FORM & 1) != 0, DPP = (FORM & 2) != 0;
    constexpr int VX = 16 / sizeof(T), TY = 4, R1 = TY + 4, R2 = TY + 2, TX = 64 * VX * WX, LW = TX + 2 * VX;
    __shared__ __attribute__((aligned(16))) T cen[2][R2][LW];
    __shared__ T edgeW[2][R2][WX], edgeE[2][R2][WX];
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int yb = TY * ty;
    const int z0 = a.zbeg + tz * a.zc, z1 = min(z0 + a.zc, a.zend);
    if (z0 >= z1) { if (NORM && tid == 0) a.partials[blockIdx.x] = 0.0; return; }
    const int xl = VX * tid, x0 = xl;
    const bool xok = x0 < a.nx;
    const bool lastvec = (x0 + VX > a.nx);

    bool uok[R1], s1ok[R2];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) { const int y = yb - 2 + rr; uok[rr] = xok && y >= -1 && y <= a.ny; }
#pragma unroll
    for (int q = 0; q < R2; q++) { const int y = yb - 1 + q; s1ok[q] = y >= 0 && y < a.ny; }
    const long uoff = (long)(yb - 2) * a.rs + x0;
    const T *bp_ = a.b + (long)(yb - 1) * a.rs + x0;
    const int pmin = a.has_lo ? -2 : -1, pmax = a.has_hi ? a.nz + 1 : a.nz;
    const int smin = a.has_lo ? -1 : 0, smax = a.has_hi ? a.nz : a.nz - 1;
    auto uplane = [&](int p) -> const T * {
        return (p == -2 ? a.far_lo : p == a.nz + 1 ? a.far_hi : a.u + (long)p * a.ms) + uoff;
    };
    for (int i = tid; i < 2 * R2 * LW; i += 64 * WX) (&cen[0][0][0])[i] = (T)0;
    const unsigned lb = (unsigned)(x0 * (int)sizeof(T));
    long uro[R1], bro[R2];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) uro[rr] = (long)max(-1, min(yb - 2 + rr, a.ny)) * a.rs;
#pragma unroll
    for (int q = 0; q < R2; q++) bro[q] = (long)max(0, min(yb - 1 + q, a.ny)) * a.rs;
    auto LDU = [&](int p, int rr, bool pv) -> VT {
        if (UNC) {
            const int pp = max(pmin, min(p, pmax));
            const T *pl = (pp == -2) ? a.far_lo : (pp == a.nz + 1) ? a.far_hi : a.u + (long)pp * a.ms;
            return *reinterpret_cast<const VT *>(reinterpret_cast<const char *>(pl + uro[rr]) + lb);
        }
        return ldv(uplane(p) + (long)rr * a.rs, uok[rr] && pv);
    };
    auto LDB = [&](int p, int q, bool pv, bool stream) -> VT {
        if (UNC) {
            const T *pl = a.b + (long)max(smin, min(p, smax)) * a.ms + bro[q];
            const T *ad = reinterpret_cast<const T *>(reinterpret_cast<const char *>(pl) + lb);
            return stream ? ldv_stream(ad, true) : ldv(ad, true);
        }
        const T *bp = bp_ + (long)p * a.ms + (long)q * a.rs;
        const bool ok = xok && s1ok[q] && pv;
        return stream ? ldv_stream(bp, ok) : ldv(bp, ok);
    };

    VT ua[R1], ub[R1], uc[R1], b1[R2], bn[R2], b0[TY], wm[TY], wc[TY], wp[TY];
    double nacc = 0.0;
    const int t0 = z0 - 2;
#pragma unroll
    for (int rr = 0; rr < R1; rr++) {
        const long ro = (long)rr * a.rs;
        (void)ro;
        ua[rr] = LDU(t0, rr, t0 >= pmin);
        ub[rr] = LDU(t0 + 1, rr, t0 + 1 >= pmin);
        uc[rr] = LDU(t0 + 2, rr, true);
    }
#pragma unroll
    for (int q = 0; q < R2; q++) {
        const int p = t0 + 1;
        b1[q] = LDB(p, q, p >= smin && p <= smax, false);
        bn[q] = v16_zero<T>();
    }
#pragma unroll
    for (int j = 0; j < TY; j++) { b0[j] = v16_zero<T>(); wm[j] = b0[j]; wc[j] = b0[j]; wp[j] = b0[j]; }
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int q = 0; q < R2; q++) {
            if (lane == 0) edgeW[(t0 + 1) & 1][q][w] = ub[q + 1].v[0];
            else edgeE[(t0 + 1) & 1][q][w] = ub[q + 1].v[VX - 1];
        }
    }
    __syncthreads();

    for (int t = t0; t < z1; t++) {
        const int p = t + 1;
        {   // b of the next step (u of the next step is issued after the first sweep, into the registers it frees)
            const bool pb = (t + 2 <= smax) && (t + 1 < z1);
#pragma unroll
            for (int q = 0; q < R2; q++) bn[q] = LDB(t + 2, q, pb, q >= 2 && q < R2 - 2);
        }
        // ---- second sweep of plane t: neighbours in x / y from cen[t & 1] (written in step t-1), z neighbours wm / wp ... ----
        // (wp = u'(t+1) is produced by the first sweep below, so the first sweep goes first)
        {
            const bool pin = (p >= smin && p <= smax);
            const int eb = p & 1, cb = p & 1;
#pragma unroll
            for (int q = 0; q < R2; q++) {
                const int rr = q + 1;
                T Wv = lane_up<DPP>(ub[rr].v[VX - 1]), Ev = lane_dn<DPP>(ub[rr].v[0]);
                if (lane == 0) Wv = (w > 0) ? edgeE[eb][q][w - 1] : (T)0;
                if (lane == 63) Ev = (w < WX - 1) ? edgeW[eb][q][w + 1] : (T)0;
                VT o;
#pragma unroll
                for (int e = 0; e < VX; e++) {
                    const T wv = (e == 0) ? Wv : ub[rr].v[e - 1 < 0 ? 0 : e - 1];
                    const T ev = (e == VX - 1) ? Ev : ub[rr].v[e + 1 > VX - 1 ? VX - 1 : e + 1];
                    T s = a.a0 * ua[rr].v[e];
                    s = s + a.a1 * ub[rr - 1].v[e];
                    s = s + a.a2 * wv;
                    s = s + a.a3 * ub[rr].v[e];
                    s = s + a.a4 * ev;
                    s = s + a.a5 * ub[rr + 1].v[e];
                    s = s + a.a6 * uc[rr].v[e];
                    const T res = b1[q].v[e] - s;
                    const T zz = res * a.dinv;
                    o.v[e] = ub[rr].v[e] + a.scale * zz;
                    if (!pin || !s1ok[q] || !xok || (lastvec && x0 + e >= a.nx)) o.v[e] = (T)0;
                    if (NORM == 1 && q >= 1 && q <= TY) {         // rows this tile owns, planes this chunk owns, inside the grid
                        const bool own = (p >= z0 && p < z1) && s1ok[q] && xok && !(lastvec && x0 + e >= a.nx);
                        nacc += own ? (double)res * (double)res : 0.0;
                    }
                }
                *reinterpret_cast<VT *>(&cen[cb][q][xl + VX]) = o;
                if (q >= 1 && q <= TY) wp[q - 1] = o;
            }
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int q = 0; q < R2; q++) {
                    if (lane == 0) edgeW[eb ^ 1][q][w] = uc[q + 1].v[0];
                    else edgeE[eb ^ 1][q][w] = uc[q + 1].v[VX - 1];
                }
            }
        }
        // plane t+3 of u into the registers of plane t (the first sweep is done with them)
        {
            const bool pu = (t + 3 <= pmax);
#pragma unroll
            for (int rr = 0; rr < R1; rr++) ua[rr] = LDU(t + 3, rr, pu);
        }
        if (t >= z0) {
            const int cb = t & 1;
#pragma unroll
            for (int j = 0; j < TY; j++) {
                const int q = j + 1;
                const VT sv = *reinterpret_cast<const VT *>(&cen[cb][q - 1][xl + VX]);
                const VT nv = *reinterpret_cast<const VT *>(&cen[cb][q + 1][xl + VX]);
                const T Wv = cen[cb][q][xl + VX - 1], Ev = cen[cb][q][xl + 2 * VX];
                VT o;
#pragma unroll
                for (int e = 0; e < VX; e++) {
                    const T wv = (e == 0) ? Wv : wc[j].v[e - 1 < 0 ? 0 : e - 1];
                    const T ev = (e == VX - 1) ? Ev : wc[j].v[e + 1 > VX - 1 ? VX - 1 : e + 1];
                    T s = a.a0 * wm[j].v[e];
                    s = s + a.a1 * sv.v[e];
                    s = s + a.a2 * wv;
                    s = s + a.a3 * wc[j].v[e];
                    s = s + a.a4 * ev;
                    s = s + a.a5 * nv.v[e];
                    s = s + a.a6 * wp[j].v[e];
                    const T res = b0[j].v[e] - s;
                    const T zz = res * a.dinv;
                    o.v[e] = wc[j].v[e] + a.scale * zz;
                    if (lastvec && x0 + e >= a.nx) o.v[e] = (T)0;
                    if (NORM == 2) nacc += (xok && yb + j < a.ny && !(lastvec && x0 + e >= a.nx)) ? (double)res * (double)res : 0.0;
                }
                if (xok && yb + j < a.ny) stv_stream(a.out + (long)t * a.ms + (long)(yb + j) * a.rs + x0, o);
            }
        }
        __syncthreads();            // cen[(t+1)&1] = u'(t+1) complete for the next step; cen[t&1] free for u'(t+2)
#pragma unroll
        for (int j = 0; j < TY; j++) { b0[j] = b1[j + 1]; wm[j] = wc[j]; wc[j] = wp[j]; }
#pragma unroll
        for (int q = 0; q < R2; q++) b1[q] = bn[q];
#pragma unroll
        for (int rr = 0; rr < R1; rr++) { VT tmpv = ua[rr]; ua[rr] = ub[rr]; ub[rr] = uc[rr]; uc[rr] = tmpv; }
    }
    if (NORM) {
        double *red = reinterpret_cast<double *>(&cen[0][0][0]);           // free after the last barrier of the loop
        __syncthreads();
        const double sblk = block_sum(nacc, red);
        if (tid == 0) a.partials[blockIdx.x] = sblk;
    }
}



// ------------------------------------------------------------------------------------------
// The prolongation fused into a TWO-sweep pass (mg_config.fuse bit 12): out = J(J(u + P uc)) (src/solver.c:1540-1542, first two iterations
// of the post-smoothing KSPSolve): 8 B (u) + 8 B (b) + 1 B (uc) read, 8 B written per fine unknown.  The skeleton of k_jacobi2r (one
// barrier per plane, first-sweep planes of the lane's columns in registers, centre plane in LDS); the parents live in LDS (three coarse
// planes of the 5 rows a tile needs, 62 KB: 158.5 of the CU's 160 KB in all) and every plane of u gets its correction one step after it
// was requested; loads / stores through buffer descriptors, DPP old-operand wave edges.  242 VGPRs, no spill.  fp64, rows of 1024, whole
// grid.  1023^3: 4.94 ms (prolongation sweep 4.49 + sweep 4.37 as two passes).
// ------------------------------------------------------------------------------------------
struct PJ2Args {
    const double *u, *b, *uc;
    double *out;
    int nx, ny, nz, nxc, nyc, nzc;
    long rs, ms, crs, cms;
    int nty, zc;
    double a0, a1, a2, a3, a4, a5, a6, dinv, scale;
    // z-slab of a multi-GPU run (k_pj2r3<WX, true>): planes -1 / nz of u and b and planes -1 / nzc of uc are the fields' ghost planes (the
    // neighbours' boundary planes, u BEFORE the correction); plane -2 of u (far_lo), plane nz+1 of u (far_hi) and plane -2 of uc (cfar_lo)
    // come in separate plane buffers (pointers at their interior origin); [zbeg, zend): the output planes of this launch, zbeg even
    const double *far_lo, *far_hi, *cfar_lo;
    int has_lo, has_hi, zbeg, zend;
};
typedef unsigned int mgk_u2v __attribute__((ext_vector_type(2)));
template <int WX>
__global__ void __launch_bounds__(64 * WX) k_pj2r(const PJ2Args a) {
    typedef double T;
    constexpr int VX = 2, TY = 4, R1 = TY + 4, R2 = TY + 2, TX = 64 * VX * WX, LW = TX + 2 * VX, NCR = 5, LC = 64 * WX + 4;
    __shared__ __attribute__((aligned(16))) T cen[2][R2][LW];
    __shared__ T edgeW[2][R2][WX + 1], edgeE[2][R2][WX + 1];      // [.][.][0] of edgeE and [.][.][WX] of edgeW: zero pads
    __shared__ T Cb[3][NCR][LC];                                   // coarse plane k in slot k mod 3: rows 2 ty - 2 .. 2 ty + 2, column c at [c + 1]
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int yb = TY * ty;
    const int z0 = tz * a.zc, z1 = min(z0 + a.zc, a.nz);          // zc is even: z0 is even
    if (z0 >= z1) return;
    const int xl = VX * tid;
    const bool lastlane = (tid == 64 * WX - 1);
    const unsigned lb = (unsigned)(xl * (int)sizeof(T));
    const int qlo = max(0, 1 - yb), qhi = min(R2 - 1, a.ny - yb);
    const unsigned rowb = (unsigned)(a.rs * (long)sizeof(T)), plane_bytes = (unsigned)(a.ny + 2) * rowb;
    const unsigned crowb = (unsigned)(a.crs * (long)sizeof(T)), cplane_bytes = (unsigned)(a.nyc + 2) * crowb;
    unsigned urb[R1], brb[R2], crb[NCR];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) urb[rr] = (unsigned)(max(-1, min(yb - 2 + rr, a.ny)) + 1) * rowb;
#pragma unroll
    for (int q = 0; q < R2; q++) brb[q] = (unsigned)(max(0, min(yb - 1 + q, a.ny)) + 1) * rowb;
#pragma unroll
    for (int j = 0; j < NCR; j++) crb[j] = (unsigned)(max(-1, min(2 * ty - 2 + j, a.nyc)) + 1) * crowb;
    auto URS = [&](int p) { return __builtin_amdgcn_make_buffer_rsrc((void *)(a.u - a.rs + (long)max(-1, min(p, a.nz)) * a.ms), 0, (int)plane_bytes, 0x00020000); };
    auto BRS = [&](int p) { return __builtin_amdgcn_make_buffer_rsrc((void *)(a.b - a.rs + (long)max(0, min(p, a.nz - 1)) * a.ms), 0, (int)plane_bytes, 0x00020000); };
    auto CRS = [&](int k) { return __builtin_amdgcn_make_buffer_rsrc((void *)(a.uc - a.crs + (long)max(-1, min(k, a.nzc)) * a.cms), 0, (int)cplane_bytes, 0x00020000); };
    const unsigned clb = (unsigned)(min(tid, a.nxc) * (int)sizeof(T));      // coarse column tid (columns beyond the ghost column alias it: zero)
    auto ldc = [&](T (&cl)[NCR], int k) {
        const auto r = CRS(k);
#pragma unroll
        for (int j = 0; j < NCR; j++) { mgk_u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, clb, crb[j], 0); cl[j] = __builtin_bit_cast(T, v); }
    };
    auto stc = [&](const T (&cl)[NCR], int k) {
        const int sl = ((k % 3) + 3) % 3;
#pragma unroll
        for (int j = 0; j < NCR; j++) Cb[sl][j][tid + 1] = cl[j];
    };
    // u + P uc on the 8 rows of plane z held in P_: row rr <-> fine row yb-2+rr (rr even: even fine row, parent rows rr/2 and rr/2+1 of the
    // tile's five; rr odd: one parent row (rr+1)/2); an even plane has the parent planes (z-1)>>1 and that + 1, an odd one (z-1)>>1 alone;
    // the even column x0 has the parent columns tid-1, tid, the odd one tid.  Terms in the order of the prolongation's row (plane, row,
    // column ascending), weights w = wk * (wi * wj).  Ghost parents are zero, so ghost rows / planes / the ghost column get +0.
    auto correct = [&](VT (&P_)[R1], int z) {
        const bool two = ((z & 1) == 0);
        const int kl = (z - 1) >> 1;
        const int sA = ((kl % 3) + 3) % 3, sB = (((kl + 1) % 3) + 3) % 3;
        const T wk = two ? 0.5 : 1.0;
#pragma unroll
        for (int rr = 0; rr < R1; rr++) {
            const bool tworows = ((rr & 1) == 0);
            const int j0 = tworows ? rr / 2 : (rr + 1) / 2;
            const T wi = tworows ? 0.5 : 1.0;
            const T wh = wk * (wi * 0.5), w1 = wk * (wi * 1.0);
            T s0 = 0.0, s1 = 0.0;
            {   // parent plane kl (always)
                const T c0 = Cb[sA][j0][tid], c1 = Cb[sA][j0][tid + 1];
                s0 += wh * c0; s0 += wh * c1; s1 += w1 * c1;
                if (tworows) { const T d0 = Cb[sA][j0 + 1][tid], d1 = Cb[sA][j0 + 1][tid + 1]; s0 += wh * d0; s0 += wh * d1; s1 += w1 * d1; }
            }
            if (two) {   // wave-uniform: an even plane has a second parent plane
                const T c0 = Cb[sB][j0][tid], c1 = Cb[sB][j0][tid + 1];
                s0 += wh * c0; s0 += wh * c1; s1 += w1 * c1;
                if (tworows) { const T d0 = Cb[sB][j0 + 1][tid], d1 = Cb[sB][j0 + 1][tid + 1]; s0 += wh * d0; s0 += wh * d1; s1 += w1 * d1; }
            }
            P_[rr].v[0] = P_[rr].v[0] + s0; P_[rr].v[1] = P_[rr].v[1] + s1;
        }
    };
    for (int i = tid; i < 2 * R2 * LW; i += 64 * WX) (&cen[0][0][0])[i] = (T)0;
    for (int i = tid; i < 2 * R2 * (WX + 1); i += 64 * WX) { (&edgeW[0][0][0])[i] = (T)0; (&edgeE[0][0][0])[i] = (T)0; }
    for (int i = tid; i < 3 * NCR * LC; i += 64 * WX) (&Cb[0][0][0])[i] = (T)0;
    __syncthreads();

    VT ua[R1], ub[R1], uc[R1], b1[R2], bn[R2], b0[TY], wm[TY], wc[TY], wp[TY];
    const int t0 = z0 - 2;
    {
        const int m = z0 >> 1;
        T c0[NCR], c1[NCR], c2[NCR];
        ldc(c0, m - 2); ldc(c1, m - 1); ldc(c2, m);
        const auto r0 = URS(t0), r1 = URS(t0 + 1), r2 = URS(t0 + 2);
        const auto rb = BRS(t0 + 1);
#pragma unroll
        for (int rr = 0; rr < R1; rr++) { ua[rr] = bufld<T>(r0, lb, urb[rr]); ub[rr] = bufld<T>(r1, lb, urb[rr]); uc[rr] = bufld<T>(r2, lb, urb[rr]); }
#pragma unroll
        for (int q = 0; q < R2; q++) { b1[q] = bufld<T>(rb, lb, brb[q]); bn[q] = v16_zero<T>(); }
        stc(c0, m - 2); stc(c1, m - 1); stc(c2, m);
    }
#pragma unroll
    for (int j = 0; j < TY; j++) { b0[j] = v16_zero<T>(); wm[j] = b0[j]; wc[j] = b0[j]; wp[j] = b0[j]; }
    __syncthreads();
    correct(ua, t0); correct(ub, t0 + 1); correct(uc, t0 + 2);
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int q = 0; q < R2; q++) {
            if (lane == 0) edgeW[(t0 + 1) & 1][q][w] = ub[q + 1].v[0];
            else edgeE[(t0 + 1) & 1][q][w + 1] = ub[q + 1].v[VX - 1];
        }
    }
    __syncthreads();

    for (int t = t0; t < z1; t++) {
        const int p = t + 1;
        T cnew[NCR];
        const bool newc = ((t & 1) != 0);                         // odd step t = 2k-3: coarse plane k = (t+3)/2 is requested, in LDS from the next step on
        if (newc) ldc(cnew, (t + 3) >> 1);
        {
            const auto rb = BRS(t + 2);
#pragma unroll
            for (int q = 0; q < R2; q++) bn[q] = (q >= 2 && q < R2 - 2) ? bufld_nt<T>(rb, lb, brb[q]) : bufld<T>(rb, lb, brb[q]);
        }
        if (t > t0) correct(uc, t + 2);                           // the plane requested in the last step (its edges below come after this)
        // ---- first sweep of plane p on rows yb-1 .. yb+TY ----
        {
            const bool pin = (p >= 0 && p < a.nz);
            const int eb = p & 1, cb = p & 1;
            const int qlo_t = pin ? qlo : R2;
            const unsigned span_t = (unsigned)(qhi - qlo_t);
#pragma unroll
            for (int q = 0; q < R2; q++) {
                const int rr = q + 1;
                VT o = v16_zero<T>();
                if ((unsigned)(q - qlo_t) <= span_t && qhi >= qlo_t) {
                    const T Wv = lane_up_old(ub[rr].v[VX - 1], edgeE[eb][q][w]);
                    const T Ev = lane_dn_old(ub[rr].v[0], edgeW[eb][q][w + 1]);
#pragma unroll
                    for (int e = 0; e < VX; e++) {
                        const T wv = (e == 0) ? Wv : ub[rr].v[0];
                        const T ev = (e == VX - 1) ? Ev : ub[rr].v[VX - 1];
                        T s = a.a0 * ua[rr].v[e];
                        s = s + a.a1 * ub[rr - 1].v[e];
                        s = s + a.a2 * wv;
                        s = s + a.a3 * ub[rr].v[e];
                        s = s + a.a4 * ev;
                        s = s + a.a5 * ub[rr + 1].v[e];
                        s = s + a.a6 * uc[rr].v[e];
                        const T res = b1[q].v[e] - s;
                        const T zz = res * a.dinv;
                        o.v[e] = ub[rr].v[e] + a.scale * zz;
                    }
                    if (lastlane) o.v[VX - 1] = (T)0;
                }
                *reinterpret_cast<VT *>(&cen[cb][q][xl + VX]) = o;
                if (q >= 1 && q <= TY) wp[q - 1] = o;
            }
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int q = 0; q < R2; q++) {
                    if (lane == 0) edgeW[eb ^ 1][q][w] = uc[q + 1].v[0];
                    else edgeE[eb ^ 1][q][w + 1] = uc[q + 1].v[VX - 1];
                }
            }
        }
        {
            const auto r3 = URS(t + 3);
#pragma unroll
            for (int rr = 0; rr < R1; rr++) ua[rr] = bufld<T>(r3, lb, urb[rr]);
        }
        // ---- second sweep of plane t ----
        if (t >= z0) {
            const int cb = t & 1;
            const auto ro = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out - a.rs + (long)t * a.ms), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < TY; j++) {
                const int q = j + 1;
                if (yb + j < a.ny) {
                    const VT sv = *reinterpret_cast<const VT *>(&cen[cb][q - 1][xl + VX]);
                    const VT nv = *reinterpret_cast<const VT *>(&cen[cb][q + 1][xl + VX]);
                    const T Wv = cen[cb][q][xl + VX - 1], Ev = cen[cb][q][xl + 2 * VX];
                    VT o;
#pragma unroll
                    for (int e = 0; e < VX; e++) {
                        const T wv = (e == 0) ? Wv : wc[j].v[0];
                        const T ev = (e == VX - 1) ? Ev : wc[j].v[VX - 1];
                        T s = a.a0 * wm[j].v[e];
                        s = s + a.a1 * sv.v[e];
                        s = s + a.a2 * wv;
                        s = s + a.a3 * wc[j].v[e];
                        s = s + a.a4 * ev;
                        s = s + a.a5 * nv.v[e];
                        s = s + a.a6 * wp[j].v[e];
                        const T res = b0[j].v[e] - s;
                        const T zz = res * a.dinv;
                        o.v[e] = wc[j].v[e] + a.scale * zz;
                    }
                    if (lastlane) o.v[VX - 1] = (T)0;
                    bufst_nt<T>(o, ro, lb, (unsigned)(yb + j + 1) * rowb);
                }
            }
        }
        if (newc) stc(cnew, (t + 3) >> 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TY; j++) { b0[j] = b1[j + 1]; wm[j] = wc[j]; wc[j] = wp[j]; }
#pragma unroll
        for (int q = 0; q < R2; q++) b1[q] = bn[q];
#pragma unroll
        for (int rr = 0; rr < R1; rr++) { VT tmpv = ua[rr]; ua[rr] = ub[rr]; ub[rr] = uc[rr]; uc[rr] = tmpv; }
    }
}
template <int WX, bool SLAB>
__global__ void __launch_bounds__(64 * WX) k_pj2r3(const PJ2Args a) {
    typedef double T;
    constexpr int VX = 2, TY = 4, R1 = TY + 4, R2 = TY + 2, TX = 64 * VX * WX, LW = TX + 2 * VX, NCR = 5, LC = 64 * WX + 4;
    __shared__ __attribute__((aligned(16))) T cen[2][R2][LW];
    __shared__ T edgeW[2][R2][WX + 1], edgeE[2][R2][WX + 1];      // [.][.][0] of edgeE and [.][.][WX] of edgeW: zero pads
    __shared__ T Cb[3][NCR][LC];                                   // coarse plane k in slot k mod 3: rows 2 ty - 2 .. 2 ty + 2, column c at [c + 1]
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int yb = TY * ty;
    const int z0 = (SLAB ? a.zbeg : 0) + tz * a.zc, z1 = min(z0 + a.zc, SLAB ? a.zend : a.nz);          // zc (and zbeg) even: z0 is even
    const int pmin = (SLAB && a.has_lo) ? -2 : -1, pmax = (SLAB && a.has_hi) ? a.nz + 1 : a.nz;     // u planes that exist
    const int smin = (SLAB && a.has_lo) ? -1 : 0, smax = (SLAB && a.has_hi) ? a.nz : a.nz - 1;      // planes on which the first sweep is real
    const int cmin = (SLAB && a.has_lo) ? -2 : -1;                                                  // coarse planes that exist: cmin .. nzc
    if (z0 >= z1) return;
    const int xl = VX * tid;
    const bool lastlane = (tid == 64 * WX - 1);
    const unsigned lb = (unsigned)(xl * (int)sizeof(T));
    const int qlo = max(0, 1 - yb), qhi = min(R2 - 1, a.ny - yb);
    const unsigned rowb = (unsigned)(a.rs * (long)sizeof(T)), plane_bytes = (unsigned)(a.ny + 2) * rowb;
    const unsigned crowb = (unsigned)(a.crs * (long)sizeof(T)), cplane_bytes = (unsigned)(a.nyc + 2) * crowb;
    unsigned urb[R1], brb[R2], crb[NCR];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) urb[rr] = (unsigned)(max(-1, min(yb - 2 + rr, a.ny)) + 1) * rowb;
#pragma unroll
    for (int q = 0; q < R2; q++) brb[q] = (unsigned)(max(0, min(yb - 1 + q, a.ny)) + 1) * rowb;
#pragma unroll
    for (int j = 0; j < NCR; j++) crb[j] = (unsigned)(max(-1, min(2 * ty - 2 + j, a.nyc)) + 1) * crowb;
    auto URS = [&](int p) {
        const int pp = max(pmin, min(p, pmax));
        const double *pl = (SLAB && pp == -2) ? a.far_lo : (SLAB && pp == a.nz + 1) ? a.far_hi : a.u + (long)pp * a.ms;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(pl - a.rs), 0, (int)plane_bytes, 0x00020000);
    };
    auto BRS = [&](int p) { return __builtin_amdgcn_make_buffer_rsrc((void *)(a.b - a.rs + (long)max(smin, min(p, smax)) * a.ms), 0, (int)plane_bytes, 0x00020000); };
    auto CRS = [&](int k) {
        const int kk = max(cmin, min(k, a.nzc));
        const double *pl = (SLAB && kk == -2) ? a.cfar_lo : a.uc + (long)kk * a.cms;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(pl - a.crs), 0, (int)cplane_bytes, 0x00020000);
    };
    const unsigned clb = (unsigned)(min(tid, a.nxc) * (int)sizeof(T));      // coarse column tid (columns beyond the ghost column alias it: zero)
    auto ldc = [&](T (&cl)[NCR], int k) {
        const auto r = CRS(k);
#pragma unroll
        for (int j = 0; j < NCR; j++) { mgk_u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, clb, crb[j], 0); cl[j] = __builtin_bit_cast(T, v); }
    };
    auto stc = [&](const T (&cl)[NCR], int k) {
        const int sl = ((k % 3) + 3) % 3;
#pragma unroll
        for (int j = 0; j < NCR; j++) Cb[sl][j][tid + 1] = cl[j];
    };
    // u + P uc on the 8 rows of plane z held in P_: row rr <-> fine row yb-2+rr (rr even: even fine row, parent rows rr/2 and rr/2+1 of the
    // tile's five; rr odd: one parent row (rr+1)/2); an even plane has the parent planes (z-1)>>1 and that + 1, an odd one (z-1)>>1 alone;
    // the even column x0 has the parent columns tid-1, tid, the odd one tid.  Terms in the order of the prolongation's row (plane, row,
    // column ascending), weights w = wk * (wi * wj).  Ghost parents are zero, so ghost rows / planes / the ghost column get +0.
    auto correct = [&](VT (&P_)[R1], int z) {
        const bool two = ((z & 1) == 0);
        const int kl = (z - 1) >> 1;
        const int sA = ((kl % 3) + 3) % 3, sB = (((kl + 1) % 3) + 3) % 3;
        const T wk = two ? 0.5 : 1.0;
#pragma unroll
        for (int rr = 0; rr < R1; rr++) {
            const bool tworows = ((rr & 1) == 0);
            const int j0 = tworows ? rr / 2 : (rr + 1) / 2;
            const T wi = tworows ? 0.5 : 1.0;
            const T wh = wk * (wi * 0.5), w1 = wk * (wi * 1.0);
            T s0 = 0.0, s1 = 0.0;
            {   // parent plane kl (always)
                const T c0 = Cb[sA][j0][tid], c1 = Cb[sA][j0][tid + 1];
                s0 += wh * c0; s0 += wh * c1; s1 += w1 * c1;
                if (tworows) { const T d0 = Cb[sA][j0 + 1][tid], d1 = Cb[sA][j0 + 1][tid + 1]; s0 += wh * d0; s0 += wh * d1; s1 += w1 * d1; }
            }
            if (two) {   // wave-uniform: an even plane has a second parent plane
                const T c0 = Cb[sB][j0][tid], c1 = Cb[sB][j0][tid + 1];
                s0 += wh * c0; s0 += wh * c1; s1 += w1 * c1;
                if (tworows) { const T d0 = Cb[sB][j0 + 1][tid], d1 = Cb[sB][j0 + 1][tid + 1]; s0 += wh * d0; s0 += wh * d1; s1 += w1 * d1; }
            }
            P_[rr].v[0] = P_[rr].v[0] + s0; P_[rr].v[1] = P_[rr].v[1] + s1;
        }
    };
    for (int i = tid; i < 2 * R2 * LW; i += 64 * WX) (&cen[0][0][0])[i] = (T)0;
    for (int i = tid; i < 2 * R2 * (WX + 1); i += 64 * WX) { (&edgeW[0][0][0])[i] = (T)0; (&edgeE[0][0][0])[i] = (T)0; }
    for (int i = tid; i < 3 * NCR * LC; i += 64 * WX) (&Cb[0][0][0])[i] = (T)0;
    __syncthreads();

    VT ua[R1], ub[R1], uc[R1], b1[R2], bn[R2], b0[TY], wm[TY], wc[TY], wp[TY];
    const int t0 = z0 - 2;
    {
        const int m = z0 >> 1;
        T c0[NCR], c1[NCR], c2[NCR];
        ldc(c0, m - 2); ldc(c1, m - 1); ldc(c2, m);
        const auto r0 = URS(t0), r1 = URS(t0 + 1), r2 = URS(t0 + 2);
        const auto rb = BRS(t0 + 1);
#pragma unroll
        for (int rr = 0; rr < R1; rr++) { ua[rr] = bufld<T>(r0, lb, urb[rr]); ub[rr] = bufld<T>(r1, lb, urb[rr]); uc[rr] = bufld<T>(r2, lb, urb[rr]); }
#pragma unroll
        for (int q = 0; q < R2; q++) { b1[q] = bufld<T>(rb, lb, brb[q]); bn[q] = v16_zero<T>(); }
        stc(c0, m - 2); stc(c1, m - 1); stc(c2, m);
    }
#pragma unroll
    for (int j = 0; j < TY; j++) { b0[j] = v16_zero<T>(); wm[j] = b0[j]; wc[j] = b0[j]; wp[j] = b0[j]; }
    __syncthreads();
    correct(ua, t0); correct(ub, t0 + 1); correct(uc, t0 + 2);
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int q = 0; q < R2; q++) {
            if (lane == 0) edgeW[(t0 + 1) & 1][q][w] = ub[q + 1].v[0];
            else edgeE[(t0 + 1) & 1][q][w + 1] = ub[q + 1].v[VX - 1];
        }
    }
    __syncthreads();

    auto step = [&](VT (&ua)[R1], VT (&ub)[R1], VT (&uc)[R1], const int t) __attribute__((always_inline)) {
        const int p = t + 1;
        T cnew[NCR];
        const bool newc = ((t & 1) != 0);                         // odd step t = 2k-3: coarse plane k = (t+3)/2 is requested, in LDS from the next step on
        if (newc) ldc(cnew, (t + 3) >> 1);
        {
            const auto rb = BRS(t + 2);
#pragma unroll
            for (int q = 0; q < R2; q++) bn[q] = (q >= 2 && q < R2 - 2) ? bufld_nt<T>(rb, lb, brb[q]) : bufld<T>(rb, lb, brb[q]);
        }
        if (t > t0) correct(uc, t + 2);                           // the plane requested in the last step (its edges below come after this)
        // ---- first sweep of plane p on rows yb-1 .. yb+TY ----
        {
            const bool pin = (p >= smin && p <= smax);
            const int eb = p & 1, cb = p & 1;
            const int qlo_t = pin ? qlo : R2;
            const unsigned span_t = (unsigned)(qhi - qlo_t);
#pragma unroll
            for (int q = 0; q < R2; q++) {
                const int rr = q + 1;
                VT o = v16_zero<T>();
                if ((unsigned)(q - qlo_t) <= span_t && qhi >= qlo_t) {
                    const T Wv = lane_up_old(ub[rr].v[VX - 1], edgeE[eb][q][w]);
                    const T Ev = lane_dn_old(ub[rr].v[0], edgeW[eb][q][w + 1]);
#pragma unroll
                    for (int e = 0; e < VX; e++) {
                        const T wv = (e == 0) ? Wv : ub[rr].v[0];
                        const T ev = (e == VX - 1) ? Ev : ub[rr].v[VX - 1];
                        T s = a.a0 * ua[rr].v[e];
                        s = s + a.a1 * ub[rr - 1].v[e];
                        s = s + a.a2 * wv;
                        s = s + a.a3 * ub[rr].v[e];
                        s = s + a.a4 * ev;
                        s = s + a.a5 * ub[rr + 1].v[e];
                        s = s + a.a6 * uc[rr].v[e];
                        const T res = b1[q].v[e] - s;
                        const T zz = res * a.dinv;
                        o.v[e] = ub[rr].v[e] + a.scale * zz;
                    }
                    if (lastlane) o.v[VX - 1] = (T)0;
                }
                *reinterpret_cast<VT *>(&cen[cb][q][xl + VX]) = o;
                if (q >= 1 && q <= TY) wp[q - 1] = o;
            }
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int q = 0; q < R2; q++) {
                    if (lane == 0) edgeW[eb ^ 1][q][w] = uc[q + 1].v[0];
                    else edgeE[eb ^ 1][q][w + 1] = uc[q + 1].v[VX - 1];
                }
            }
        }
        {
            const auto r3 = URS(t + 3);
#pragma unroll
            for (int rr = 0; rr < R1; rr++) ua[rr] = bufld<T>(r3, lb, urb[rr]);
        }
        // ---- second sweep of plane t ----
        if (t >= z0) {
            const int cb = t & 1;
            const auto ro = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out - a.rs + (long)t * a.ms), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < TY; j++) {
                const int q = j + 1;
                if (yb + j < a.ny) {
                    const VT sv = *reinterpret_cast<const VT *>(&cen[cb][q - 1][xl + VX]);
                    const VT nv = *reinterpret_cast<const VT *>(&cen[cb][q + 1][xl + VX]);
                    const T Wv = cen[cb][q][xl + VX - 1], Ev = cen[cb][q][xl + 2 * VX];
                    VT o;
#pragma unroll
                    for (int e = 0; e < VX; e++) {
                        const T wv = (e == 0) ? Wv : wc[j].v[0];
                        const T ev = (e == VX - 1) ? Ev : wc[j].v[VX - 1];
                        T s = a.a0 * wm[j].v[e];
                        s = s + a.a1 * sv.v[e];
                        s = s + a.a2 * wv;
                        s = s + a.a3 * wc[j].v[e];
                        s = s + a.a4 * ev;
                        s = s + a.a5 * nv.v[e];
                        s = s + a.a6 * wp[j].v[e];
                        const T res = b0[j].v[e] - s;
                        const T zz = res * a.dinv;
                        o.v[e] = wc[j].v[e] + a.scale * zz;
                    }
                    if (lastlane) o.v[VX - 1] = (T)0;
                    bufst_nt<T>(o, ro, lb, (unsigned)(yb + j + 1) * rowb);
                }
            }
        }
        if (newc) stc(cnew, (t + 3) >> 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TY; j++) { b0[j] = b1[j + 1]; wm[j] = wc[j]; wc[j] = wp[j]; }
#pragma unroll
        for (int q = 0; q < R2; q++) b1[q] = bn[q];
    };
    for (int t = t0; t < z1; t += 3) {
        step(ua, ub, uc, t);
        if (t + 1 < z1) step(ub, uc, ua, t + 1);
        if (t + 2 < z1) step(uc, ua, ub, t + 2);
    }
}
extern "C" int mgk_prolong_jacobi2_ok_f64(const mgk_geom *gf, const mgk_geom *gc) {
    if (!gf || !gc || gf->dim != 3 || gc->dim != 3 || gf->nx != 2 * gc->nx + 1 || gf->ny != 2 * gc->ny + 1 || gf->nz != 2 * gc->nz + 1) return 0;
    if ((gf->nx + 1) % 128 != 0 || (gf->ny + 1) % 4 != 0) return 0;
    const int w = (gf->nx + 1) / 128;
    return (w == 4 || w == 8) ? 1 : 0;                                         // rows of 512 / 1024 (n = 511, 1023)
}
// unew = J(J(u + P uc)): the prolongation, its correction and the first two post-smoothing sweeps in one pass
extern "C" int mgk_prolong_jacobi2_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                       const double *b, const double *uc, const double *u, double *unew, void *stream) {
    if (!c || !coef || !b || !uc || !u || !unew || u == unew || !mgk_prolong_jacobi2_ok_f64(gf, gc)) return fail(MGK_EINVAL, "mgk_prolong_jacobi2_f64: bad arguments / shape not built");
    PJ2Args a; memset(&a, 0, sizeof(a));
    a.u = u + gf->org; a.b = b + gf->org; a.uc = uc + gc->org; a.out = unew + gf->org;
    a.nx = gf->nx; a.ny = gf->ny; a.nz = gf->nz; a.nxc = gc->nx; a.nyc = gc->ny; a.nzc = gc->nz;
    a.rs = gf->pitch; a.ms = gf->plane; a.crs = gc->pitch; a.cms = gc->plane;
    a.a0 = coef[0]; a.a1 = coef[1]; a.a2 = coef[2]; a.a3 = coef[3]; a.a4 = coef[4]; a.a5 = coef[5]; a.a6 = coef[6];
    a.dinv = dinv; a.scale = scale;
    a.nty = (gf->ny + 3) / 4;
    const long target = ((gf->nx + 1) / 128 > 4) ? 256 : 512;    // 512-thread blocks: one per CU; 256-thread blocks: two
    long nch = (a.nty >= target) ? 1 : (target + a.nty - 1) / a.nty;
    if (g_zchunk > 0) nch = (gf->nz + g_zchunk - 1) / g_zchunk;
    int zc = (int)((gf->nz + nch - 1) / nch);
    zc = (zc + 1) & ~1;                                          // even: every chunk starts on an even plane
    if (zc < 8) zc = 8;
    a.zc = zc;
    const unsigned nblk = (unsigned)(a.nty * ((gf->nz + zc - 1) / zc));
    if ((gf->nx + 1) / 128 == 4 && g_variant != 46) hipLaunchKernelGGL((k_pj2r3<4, false>), dim3(nblk), dim3(256), 0, S(c, stream), a);   // round 3: the unrolled form on rows of
                                                                                                        // 512 too: 511^3 0.651 -> 0.601 ms (253 VGPRs, two blocks per CU)
    else if ((gf->nx + 1) / 128 == 4) hipLaunchKernelGGL((k_pj2r<4>), dim3(nblk), dim3(256), 0, S(c, stream), a);
    else if (g_variant != 46) hipLaunchKernelGGL((k_pj2r3<8, false>), dim3(nblk), dim3(512), 0, S(c, stream), a);      // marching loop unrolled by three, the plane roles
                                                                                                        // permuted instead of copied: 4.80 against 4.90 ms (46: the copying form)
    else hipLaunchKernelGGL((k_pj2r<8>), dim3(nblk), dim3(512), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}

// The same on a z-slab of a multi-GPU run, output planes [zbeg, zend) (zbeg even).  u's and b's ghost planes hold the neighbours' boundary planes
// (u BEFORE the correction), uc's ghost planes the neighbours' coarse boundary planes; `far` / `cfar`: fields of geometry gfar = (nx, ny, 2) /
// gcfar = (nxc, nyc, 2) whose ghost planes hold the neighbours' SECOND planes (lo ghost: plane nz-2 / nzc-2 of the rank below, hi ghost: plane 1
// of the rank above; of cfar only the lo ghost is read).  With them every rank corrects and sweeps the planes -2 .. nz+1 it needs itself -- same
// operands, same arithmetic as their owner: same bits.  An inner slab (has_hi) owns nz = 2 nzc planes, the last one 2 nzc + 1.
extern "C" int mgk_prolong_jacobi2_slab_ok_f64(const mgk_geom *gf, const mgk_geom *gc, int has_hi) {
    if (!gf || !gc || gf->dim != 3 || gc->dim != 3 || gf->nx != 2 * gc->nx + 1 || gf->ny != 2 * gc->ny + 1) return 0;
    if (gf->nz != (has_hi ? 2 * gc->nz : 2 * gc->nz + 1) || gf->nz < 4 || gc->nz < 2) return 0;
    if ((gf->nx + 1) % 128 != 0 || (gf->ny + 1) % 4 != 0) return 0;
    const int w = (gf->nx + 1) / 128;
    return (w == 4 || w == 8) ? 1 : 0;
}
extern "C" int mgk_prolong_jacobi2_slab_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const mgk_geom *gcfar, const double *coef,
                                            double dinv, double scale, const double *b, const double *uc, const double *u, double *unew,
                                            const double *far, const double *cfar, int has_lo, int has_hi, int zbeg, int zend, void *stream) {
    if (!c || !coef || !b || !uc || !u || !unew || u == unew || !mgk_prolong_jacobi2_slab_ok_f64(gf, gc, has_hi))
        return fail(MGK_EINVAL, "mgk_prolong_jacobi2_slab_f64: bad arguments / shape not built");
    if ((has_lo || has_hi) && (!far || !gfar || gfar->dim != 3 || gfar->nz != 2 || gfar->nx != gf->nx || gfar->ny != gf->ny || gfar->pitch != gf->pitch))
        return fail(MGK_EINVAL, "mgk_prolong_jacobi2_slab_f64: the far-plane field must have the geometry (nx, ny, 2) of the slab");
    if (has_lo && (!cfar || !gcfar || gcfar->dim != 3 || gcfar->nz != 2 || gcfar->nx != gc->nx || gcfar->ny != gc->ny || gcfar->pitch != gc->pitch))
        return fail(MGK_EINVAL, "mgk_prolong_jacobi2_slab_f64: the coarse far-plane field must have the geometry (nxc, nyc, 2) of the coarse slab");
    if (zbeg < 0 || (zbeg & 1) || zend > gf->nz || zbeg >= zend) return fail(MGK_EINVAL, "mgk_prolong_jacobi2_slab_f64: empty, odd or out-of-range plane range");
    PJ2Args a; memset(&a, 0, sizeof(a));
    a.u = u + gf->org; a.b = b + gf->org; a.uc = uc + gc->org; a.out = unew + gf->org;
    a.nx = gf->nx; a.ny = gf->ny; a.nz = gf->nz; a.nxc = gc->nx; a.nyc = gc->ny; a.nzc = gc->nz;
    a.rs = gf->pitch; a.ms = gf->plane; a.crs = gc->pitch; a.cms = gc->plane;
    a.a0 = coef[0]; a.a1 = coef[1]; a.a2 = coef[2]; a.a3 = coef[3]; a.a4 = coef[4]; a.a5 = coef[5]; a.a6 = coef[6];
    a.dinv = dinv; a.scale = scale;
    a.has_lo = has_lo ? 1 : 0; a.has_hi = has_hi ? 1 : 0;
    a.far_lo = has_lo ? far + gfar->org - gfar->plane : nullptr;
    a.far_hi = has_hi ? far + gfar->org + 2 * gfar->plane : nullptr;
    a.cfar_lo = has_lo ? cfar + gcfar->org - gcfar->plane : nullptr;
    a.zbeg = zbeg; a.zend = zend;
    const int nzr = zend - zbeg;
    a.nty = (gf->ny + 3) / 4;
    const long target = ((gf->nx + 1) / 128 > 4) ? 256 : 512;
    long nch = (a.nty >= target) ? 1 : (target + a.nty - 1) / a.nty;
    if (g_zchunk > 0) nch = (nzr + g_zchunk - 1) / g_zchunk;
    else if (c->chunk_planes > 0 && nch < (nzr + c->chunk_planes - 1) / c->chunk_planes) nch = (nzr + c->chunk_planes - 1) / c->chunk_planes;
    int zc = (int)((nzr + nch - 1) / nch);
    zc = (zc + 1) & ~1;                                          // even: every chunk starts on an even plane
    if (zc < 8) zc = 8;
    a.zc = zc;
    const unsigned nblk = (unsigned)(a.nty * ((nzr + zc - 1) / zc));
    if ((gf->nx + 1) / 128 == 4) hipLaunchKernelGGL((k_pj2r3<4, true>), dim3(nblk), dim3(256), 0, S(c, stream), a);
    else hipLaunchKernelGGL((k_pj2r3<8, true>), dim3(nblk), dim3(512), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}

template <typename T> struct j2norm { static constexpr bool built = false; };
template <> struct j2norm<double> { static constexpr bool built = true; };
template <typename T>
static int jacobi2(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                   const T *b, const T *u, T *unew, const T *far_lo, const T *far_hi, int zbeg, int zend, void *stream,
                   int *norm_parts = nullptr, int part_off = 0, bool zero_guess = false, int norm_mode = 1) {
    constexpr int VX = 16 / sizeof(T);
    if (zero_guess) u = b;                                       // (not read as u: the kernel forms the first sweep from b)
    if (!c || !g || !coef || !b || !u || !unew || (u == unew && !zero_guess) || b == unew || g->dim != 3) return fail(MGK_EINVAL, "mgk_jacobi2: bad arguments (3-D)");
    if (g->nx + 1 > 1024) return fail(MGK_EINVAL, "mgk_jacobi2: nx + 1 > 1024 is not built");
    J2Args<T> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org;
    a.nx = g->nx; a.ny = g->ny; a.nz = g->nz; a.rs = g->pitch; a.ms = g->plane;
    a.a0 = (T)coef[0]; a.a1 = (T)coef[1]; a.a2 = (T)coef[2]; a.a3 = (T)coef[3]; a.a4 = (T)coef[4]; a.a5 = (T)coef[5]; a.a6 = (T)coef[6];
    a.dinv = (T)dinv; a.scale = (T)scale;
    a.far_lo = far_lo; a.far_hi = far_hi; a.has_lo = far_lo != nullptr; a.has_hi = far_hi != nullptr;
    if (zbeg < 0 || zend > g->nz || zbeg >= zend) return fail(MGK_EINVAL, "mgk_jacobi2: empty or out-of-range plane range");
    a.zbeg = zbeg; a.zend = zend;
    const int nzr = zend - zbeg;
    a.nty = (g->ny + 3) / 4;
    const int w = (g->nx + 1 + 64 * VX - 1) / (64 * VX);        // waves per full row
    // one 512-thread block per CU (LDS) for fp64 at 1023^3: 256 tiles, one chunk; blocks of <= 256 threads (fp64 511^3, fp32):
    // two per CU, 512 blocks (measured, fp64 511^3: 128 blocks 0.83 ms, 512 blocks 0.57 ms); every chunk recomputes two
    // planes of the first sweep
    // rows of <= 2 waves (255^3, the third level of the headline): blocks of 128 threads, a chunk per SIMD is too few waves to
    // cover the latency -- 1024 blocks of 16 planes (in the 511^3 cycle: 118 -> 88 us, zero-guess form 126 -> 81; 8 planes 104 / 89)
    const long target = (w > 4) ? 256 : ((sizeof(T) == 4 || w <= 2) ? 1024 : 512);      // fp32 1023^3: 1024 blocks 2.72 ms, 512 blocks 2.84 ms
    long nch = (a.nty >= target) ? 1 : (target + a.nty - 1) / a.nty;
    if (g_zchunk > 0) nch = (nzr + g_zchunk - 1) / g_zchunk;
    else if (c->chunk_planes > 0 && nch < (nzr + c->chunk_planes - 1) / c->chunk_planes) nch = (nzr + c->chunk_planes - 1) / c->chunk_planes;
    int zc = (int)((nzr + nch - 1) / nch);
    if (zc < 8) zc = 8;
    if (zc > nzr) zc = nzr;
    a.zc = zc;
    const long ntz = (nzr + zc - 1) / zc;
    const unsigned nblk = (unsigned)(a.nty * ntz);
    hipStream_t s = S(c, stream);
    if (norm_parts) {
        // with the residual norm of the input field: the register form on full-row shapes only
        constexpr int WRn = 64 * VX;
        if (!j2norm<T>::built || (g->nx + 1) % WRn != 0 || (g->ny + 1) % 4 != 0 || part_off < 0 || (long)nblk > c->max_partials - part_off)
            return fail(MGK_EINVAL, "mgk_jacobi2_sumsq: built for fp64 full-row shapes (n = 127, 255, 511, 1023)");
        a.partials = c->partials + part_off;
        if constexpr (j2norm<T>::built) {
            if (norm_mode == 2) {
                if (w <= 1) hipLaunchKernelGGL((k_jacobi2r<T, 1, 3, 2>), dim3(nblk), dim3(64), 0, s, a);
                else if (w <= 2) hipLaunchKernelGGL((k_jacobi2r<T, 2, 3, 2>), dim3(nblk), dim3(128), 0, s, a);
                else if (w <= 4) hipLaunchKernelGGL((k_jacobi2r<T, 4, 3, 2>), dim3(nblk), dim3(256), 0, s, a);
                else hipLaunchKernelGGL((k_jacobi2r<T, 8, 3, 2>), dim3(nblk), dim3(512), 0, s, a);
            } else if (w <= 1) hipLaunchKernelGGL((k_jacobi2r<T, 1, 3, 1>), dim3(nblk), dim3(64), 0, s, a);
            else if (w <= 2) hipLaunchKernelGGL((k_jacobi2r<T, 2, 3, 1>), dim3(nblk), dim3(128), 0, s, a);
            else if (w <= 4) hipLaunchKernelGGL((k_jacobi2r<T, 4, 3, 1>), dim3(nblk), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_jacobi2r<T, 8, 3, 1>), dim3(nblk), dim3(512), 0, s, a);
        }
        HIPCHK(hipGetLastError());
        *norm_parts = (int)nblk;
        return 0;
    }
    // one 512-thread block per CU (fp64, 1023^3): the one-barrier variant (4.51 vs 5.15 ms per pass); smaller blocks run two
    // per CU and hide the second barrier, and prefer the ring variant's full-step prefetch distance (fp64 511^3: 0.66 vs
    // 0.72 ms; fp32 1023^3: 3.98 vs 7.29 ms, the register variant is at the 256-VGPR limit there)
    const bool ring = (g_variant == 1) || (g_variant == 37) || (g_variant != 2 && g_variant != 36 && !(sizeof(T) == 8 && w > 4));
    // full-row shapes: unconditional loads + DPP lane shifts (tuning variants 36 / 37 keep the predicated / ds_bpermute form)
    constexpr int WR = 64 * VX;
    const bool full = (g->nx + 1) % WR == 0 && (g->ny + 1) % 4 == 0 && g_variant != 36 && g_variant != 37;
    if (zero_guess) {
        if (!full || far_lo || far_hi || norm_parts) return fail(MGK_EINVAL, "mgk_jacobi2_zero: built for whole grids of full-row shape (the LDS-ring form of the two-sweep kernel)");
        if (g_variant == 39 || (sizeof(T) == 8 && g_variant != 45)) {   // the form before the instruction diet: 39 forces it; fp64 keeps it
                                                                        // (511^3: 0.471 against 0.488 ms; 45 forces the new form)
            if (w <= 1) hipLaunchKernelGGL((k_jacobi2<T, 1, 3, true>), dim3(nblk), dim3(64), 0, s, a);
            else if (w <= 2) hipLaunchKernelGGL((k_jacobi2<T, 2, 3, true>), dim3(nblk), dim3(128), 0, s, a);
            else if (w <= 4) hipLaunchKernelGGL((k_jacobi2<T, 4, 3, true>), dim3(nblk), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_jacobi2<T, 8, 3, true>), dim3(nblk), dim3(512), 0, s, a);
        } else if (w <= 1) hipLaunchKernelGGL((k_jacobi2b<T, 1, true>), dim3(nblk), dim3(64), 0, s, a);
        else if (w <= 2) hipLaunchKernelGGL((k_jacobi2b<T, 2, true>), dim3(nblk), dim3(128), 0, s, a);
        else if (w <= 4) hipLaunchKernelGGL((k_jacobi2b<T, 4, true>), dim3(nblk), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_jacobi2b<T, 8, true>), dim3(nblk), dim3(512), 0, s, a);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (ring) {
        if (full && g_variant != 39) {
            if (w <= 1) hipLaunchKernelGGL((k_jacobi2b<T, 1, false>), dim3(nblk), dim3(64), 0, s, a);
            else if (w <= 2) hipLaunchKernelGGL((k_jacobi2b<T, 2, false>), dim3(nblk), dim3(128), 0, s, a);
            else if (w <= 4) hipLaunchKernelGGL((k_jacobi2b<T, 4, false>), dim3(nblk), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_jacobi2b<T, 8, false>), dim3(nblk), dim3(512), 0, s, a);
        } else if (full) {
            if (w <= 1) hipLaunchKernelGGL((k_jacobi2<T, 1, 3>), dim3(nblk), dim3(64), 0, s, a);
            else if (w <= 2) hipLaunchKernelGGL((k_jacobi2<T, 2, 3>), dim3(nblk), dim3(128), 0, s, a);
            else if (w <= 4) hipLaunchKernelGGL((k_jacobi2<T, 4, 3>), dim3(nblk), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_jacobi2<T, 8, 3>), dim3(nblk), dim3(512), 0, s, a);
        } else {
            if (w <= 1) hipLaunchKernelGGL((k_jacobi2<T, 1, 0>), dim3(nblk), dim3(64), 0, s, a);
            else if (w <= 2) hipLaunchKernelGGL((k_jacobi2<T, 2, 0>), dim3(nblk), dim3(128), 0, s, a);
            else if (w <= 4) hipLaunchKernelGGL((k_jacobi2<T, 4, 0>), dim3(nblk), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_jacobi2<T, 8, 0>), dim3(nblk), dim3(512), 0, s, a);
        }
    } else {
        if (full) {
            if (w <= 1) hipLaunchKernelGGL((k_jacobi2r<T, 1, 3>), dim3(nblk), dim3(64), 0, s, a);
            else if (w <= 2) hipLaunchKernelGGL((k_jacobi2r<T, 2, 3>), dim3(nblk), dim3(128), 0, s, a);
            else if (w <= 4) hipLaunchKernelGGL((k_jacobi2r<T, 4, 3>), dim3(nblk), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_jacobi2r<T, 8, 3>), dim3(nblk), dim3(512), 0, s, a);
        } else {
            if (w <= 1) hipLaunchKernelGGL((k_jacobi2r<T, 1, 0>), dim3(nblk), dim3(64), 0, s, a);
            else if (w <= 2) hipLaunchKernelGGL((k_jacobi2r<T, 2, 0>), dim3(nblk), dim3(128), 0, s, a);
            else if (w <= 4) hipLaunchKernelGGL((k_jacobi2r<T, 4, 0>), dim3(nblk), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_jacobi2r<T, 8, 0>), dim3(nblk), dim3(512), 0, s, a);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_jacobi2_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                               const double *b, const double *u, double *unew, void *stream) {
    if (!g) return fail(MGK_EINVAL, "mgk_jacobi2_f64: bad arguments");
    return jacobi2<double>(c, g, coef, dinv, scale, b, u, unew, nullptr, nullptr, 0, g->nz, stream);
}
extern "C" int mgk_jacobi2_f32(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                               const float *b, const float *u, float *unew, void *stream) {
    if (!g) return fail(MGK_EINVAL, "mgk_jacobi2_f32: bad arguments");
    return jacobi2<float>(c, g, coef, dinv, scale, b, u, unew, nullptr, nullptr, 0, g->nz, stream);
}
// THREE sweeps from a zero initial guess in one pass that reads b alone: unew = J(J(J0(b))), J0(b) = scale * (b * dinv) (k_jacobi_zero).
// Whole grids of full-row shape on the LDS-ring form (fp32: n = 255 .. 1023; fp64: n = 127 .. 511); mgk_jacobi2_zero_ok_* tells.
template <typename T>
static bool j2zero_ok(const mgk_geom *g) {
    constexpr int VX = 16 / sizeof(T);
    if (!g || g->dim != 3 || (g->nx + 1) % (64 * VX) != 0 || (g->ny + 1) % 4 != 0 || g->nx + 1 > 1024) return false;
    const int w = (g->nx + 1) / (64 * VX);
    if (g_variant == 2 || g_variant == 36 || g_variant == 37) return false;
    return w == 1 || w == 2 || w == 4 || w == 8;                 // (fp64 rows of 1024 too: the ring form, 148 KB of LDS, only from the zero guess)
}
extern "C" int mgk_jacobi2_zero_ok_f64(const mgk_geom *g) { return j2zero_ok<double>(g) ? 1 : 0; }
extern "C" int mgk_jacobi2_zero_ok_f32(const mgk_geom *g) { return j2zero_ok<float>(g) ? 1 : 0; }
extern "C" int mgk_jacobi2_zero_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                    const double *b, double *unew, void *stream) {
    if (!g) return fail(MGK_EINVAL, "mgk_jacobi2_zero_f64: bad arguments");
    return jacobi2<double>(c, g, coef, dinv, scale, b, nullptr, unew, nullptr, nullptr, 0, g->nz, stream, nullptr, 0, true);
}
extern "C" int mgk_jacobi2_zero_f32(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                    const float *b, float *unew, void *stream) {
    if (!g) return fail(MGK_EINVAL, "mgk_jacobi2_zero_f32: bad arguments");
    return jacobi2<float>(c, g, coef, dinv, scale, b, nullptr, unew, nullptr, nullptr, 0, g->nz, stream, nullptr, 0, true);
}
// Two sweeps AND || b - A u ||^2 of the input field (formed by the first sweep anyway): closes cycle k (src/solver.c:1545-1546)
// and makes the first two pre-smoothing sweeps of cycle k+1 (:1531) in one pass.  unew is only adopted if another cycle runs.
extern "C" int mgk_jacobi2_sumsq_ok_f64(const mgk_geom *g) {
    return (g && g->dim == 3 && (g->nx + 1) % 128 == 0 && (g->ny + 1) % 4 == 0 && g->nx + 1 <= 1024) ? 1 : 0;
}
extern "C" int mgk_jacobi2_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                     const double *b, const double *u, double *unew, double *sumsq_host, void *stream) {
    if (!g || !sumsq_host) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_f64: bad arguments");
    int nparts = 0;
    int rc = jacobi2<double>(c, g, coef, dinv, scale, b, u, unew, nullptr, nullptr, 0, g->nz, stream, &nparts);
    if (rc) return rc;
    return finish_to_host(c, nparts, 1, S(c, stream), sumsq_host);
}
// Two sweeps and sumsq = || b - A J(u) ||^2, the residual of the FIRST sweep's output (which is not stored): unew = J(J(u)).
extern "C" int mgk_jacobi2_sumsq_mid_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                         const double *b, const double *u, double *unew, double *sumsq_host, void *stream) {
    if (!g || !sumsq_host) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_mid_f64: bad arguments");
    int nparts = 0;
    int rc = jacobi2<double>(c, g, coef, dinv, scale, b, u, unew, nullptr, nullptr, 0, g->nz, stream, &nparts, 0, false, 2);
    if (rc) return rc;
    return finish_to_host(c, nparts, 1, S(c, stream), sumsq_host);
}
// ... on the planes [zbeg, zend) of a z-slab (far planes as for mgk_jacobi2_slab_f64), block partials deposited from slot part_off on and
// NOT reduced: interior planes while the ghosts travel, then the boundary planes, then ONE mgk_partials_finish (fixed order)
extern "C" int mgk_jacobi2_sumsq_slab_f64(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                                          const double *b, const double *u, double *unew, const double *far, int has_lo, int has_hi,
                                          int zbeg, int zend, int part_off, int *nparts, void *stream) {
    if (!g || !nparts) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_slab_f64: bad arguments");
    if ((has_lo || has_hi) && (!gfar || !far || gfar->dim != 3 || gfar->nz != 2 || gfar->nx != g->nx || gfar->ny != g->ny || gfar->pitch != g->pitch || g->nz < 2))
        return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_slab_f64: the far-plane field must have the geometry (nx, ny, 2) of the slab");
    const double *lo = has_lo ? far + gfar->org - gfar->plane : nullptr;
    const double *hi = has_hi ? far + gfar->org + 2 * gfar->plane : nullptr;
    return jacobi2<double>(c, g, coef, dinv, scale, b, u, unew, lo, hi, zbeg, zend, stream, nparts, part_off);
}
// ... the mid-iterate form (mgk_jacobi2_sumsq_mid_f64) on the planes [zbeg, zend) of a z-slab: unew = J(J(u)), partials of || b - A J(u) ||^2
extern "C" int mgk_jacobi2_sumsq_mid_slab_f64(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                                              const double *b, const double *u, double *unew, const double *far, int has_lo, int has_hi,
                                              int zbeg, int zend, int part_off, int *nparts, void *stream) {
    if (!g || !nparts) return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_mid_slab_f64: bad arguments");
    if ((has_lo || has_hi) && (!gfar || !far || gfar->dim != 3 || gfar->nz != 2 || gfar->nx != g->nx || gfar->ny != g->ny || gfar->pitch != g->pitch || g->nz < 2))
        return fail(MGK_EINVAL, "mgk_jacobi2_sumsq_mid_slab_f64: the far-plane field must have the geometry (nx, ny, 2) of the slab");
    const double *lo = has_lo ? far + gfar->org - gfar->plane : nullptr;
    const double *hi = has_hi ? far + gfar->org + 2 * gfar->plane : nullptr;
    return jacobi2<double>(c, g, coef, dinv, scale, b, u, unew, lo, hi, zbeg, zend, stream, nparts, part_off, false, 2);
}
// The same on a z-slab.  `far` is a field of geometry (nx, ny, nz = 2) whose ghost planes hold the neighbours' second plane
// (lo ghost: plane nz-2 of the rank below, hi ghost: plane 1 of the rank above; mgk_geom of it in gfar); u's own ghost
// planes hold their last / first plane and b's ghost planes their b.  has_lo / has_hi: a neighbour exists on that side.
template <typename T>
static int jacobi2_slab(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                        const T *b, const T *u, T *unew, const T *far, int has_lo, int has_hi, int zbeg, int zend, void *stream) {
    if (!g || !gfar || !far || gfar->dim != 3 || gfar->nz != 2 || gfar->nx != g->nx || gfar->ny != g->ny || gfar->pitch != g->pitch ||
        g->nz < 2)
        return fail(MGK_EINVAL, "mgk_jacobi2_slab: the far-plane field must have the geometry (nx, ny, 2) of the slab");
    const T *lo = has_lo ? far + gfar->org - gfar->plane : nullptr;
    const T *hi = has_hi ? far + gfar->org + 2 * gfar->plane : nullptr;
    return jacobi2<T>(c, g, coef, dinv, scale, b, u, unew, lo, hi, zbeg, zend, stream);
}
extern "C" int mgk_jacobi2_slab_f64(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                                    const double *b, const double *u, double *unew, const double *far, int has_lo, int has_hi,
                                    int zbeg, int zend, void *stream) {
    return jacobi2_slab<double>(c, g, gfar, coef, dinv, scale, b, u, unew, far, has_lo, has_hi, zbeg, zend, stream);
}
extern "C" int mgk_jacobi2_slab_f32(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gfar, const double *coef, double dinv, double scale,
                                    const float *b, const float *u, float *unew, const float *far, int has_lo, int has_hi,
                                    int zbeg, int zend, void *stream) {
    return jacobi2_slab<float>(c, g, gfar, coef, dinv, scale, b, u, unew, far, has_lo, has_hi, zbeg, zend, stream);
}

// ------------------------------------------------------------------------------------------
// The LAST pre-smoothing sweep fused with the residual and its full weighting (src/solver.c:1531 last iteration + :1534-1535):
//     out = J(u);   b_c = R (b - A out)   [; uc0 = scale_c * (b_c * dinv_c)]
// in one pass -- 8 B (u) + 8 B (b) read, 8 B (out) + 1 B (b_c) [+ 1 B] written per fine unknown instead of 24 + 18 B for the
// sweep and the fused residual+restriction as two passes.  Structure of k_jacobi2r (full-row tile marching along z, the lane
// owns a column pair of all rows, the swept centre plane in LDS, one barrier per plane) with the second stage replaced by
// k_rrrow's residual + running full-weighting sums.  A tile of TY fine rows owns TY/2 coarse rows, which read the residual on
// TY+1 rows, hence the sweep on TY+3 rows and u on TY+5 rows (the tile's neighbours recompute the shared rows; they come from L2).
// Same per-point expressions and summation order as the kernels it replaces: bit-identical.
// Shapes: full rows (nx + 1 == 64 * VX * WX), ny + 1 a multiple of 4, whole grid (no z-slab).
// ------------------------------------------------------------------------------------------
template <typename T>
struct SRRArgs {
    const T *u, *b;
    T *out, *bc, *uc0;
    int nx, ny, nz, nxc, nyc, nzc;
    long rs, ms, crs, cms;
    int nty, kcc;            // tiles in y, coarse planes per chunk
    T a0, a1, a2, a3, a4, a5, a6, dinv, scale, dinv_c, scale_c;
    // z-slab of a multi-GPU run (k_srr4b): planes -1 / nz of u and b are the fields' ghost planes; plane -2 of u (far_lo), planes
    // nz+1 and nz+2 of u (far_hi, far2_hi) and plane nz+1 of b (bfar_hi) come in separate plane buffers (pointers at their interior
    // origin).  An inner slab (has_hi) owns nz = 2 nzc fine planes and completes its last coarse plane with the residual of plane nz.
    const T *far_lo, *far_hi, *far2_hi, *bfar_hi;
    int has_lo, has_hi;
    int kcbeg, kcend;        // coarse planes [kcbeg, kcend) produced by this launch (whole grid / slab: 0, nzc)
};
template <typename T, int WX, int TY>
__global__ void __launch_bounds__(64 * WX) k_srr(const SRRArgs<T> a) {
    constexpr int VX = 16 / sizeof(T), NCJ = VX / 2, NCR = TY / 2, RS = TY + 1, R2 = TY + 3, R1 = TY + 5;
    constexpr int TX = 64 * VX * WX, LW = TX + 2 * VX;
    __shared__ __attribute__((aligned(16))) T cen[2][R2][LW];
    __shared__ T edgeW[2][R2][WX], edgeE[2][R2][WX], eR[2][RS][WX];
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int yb = TY * ty;
    const int kc0 = a.kcbeg + tz * a.kcc, kc1 = min(kc0 + a.kcc, a.kcend);      // (whole grids only: no far planes in this form)
    if (kc0 >= kc1) return;
    const int z0 = 2 * kc0, z1 = min(2 * kc1 + 1, a.nz);          // planes whose residual this chunk forms: [z0, z1)
    const int zs1 = (kc1 == a.nzc) ? a.nz : 2 * kc1;              // planes of the swept field this chunk stores: [z0, zs1)
    const int xl = VX * tid, x0 = xl;
    const bool lastlane = (tid == 64 * WX - 1);                   // its last element is the ghost column x = nx: stays 0
    const unsigned lb = (unsigned)(x0 * (int)sizeof(T));
    bool s1ok[R2], rok[RS];
#pragma unroll
    for (int q = 0; q < R2; q++) { const int y = yb - 1 + q; s1ok[q] = y >= 0 && y < a.ny; }
#pragma unroll
    for (int j = 0; j < RS; j++) rok[j] = (yb + j < a.ny);
    bool crow[NCR];
#pragma unroll
    for (int cl = 0; cl < NCR; cl++) crow[cl] = (NCR * ty + cl < a.nyc);
    long uro[R1], bro[R2];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) uro[rr] = (long)max(-1, min(yb - 2 + rr, a.ny)) * a.rs;
#pragma unroll
    for (int q = 0; q < R2; q++) bro[q] = (long)max(0, min(yb - 1 + q, a.ny)) * a.rs;
    // unconditional loads, rows / planes clamped on the scalar unit: whatever a clamped load brings in only reaches first-sweep
    // values that are forced to 0
    auto LDU = [&](int p, int rr) -> VT { return ldrow(a.u + (long)max(-1, min(p, a.nz)) * a.ms + uro[rr], lb); };
    auto LDB = [&](int p, int q, bool stream) -> VT {
        const T *pl = a.b + (long)max(0, min(p, a.nz - 1)) * a.ms + bro[q];
        return stream ? ldrow_stream(pl, lb) : ldrow(pl, lb);
    };
    for (int i = tid; i < 2 * R2 * LW; i += 64 * WX) (&cen[0][0][0])[i] = (T)0;
    const int jc0 = NCJ * tid;
    const T w2[3][3] = {{(T)0.0625, (T)0.125, (T)0.0625}, {(T)0.125, (T)0.25, (T)0.125}, {(T)0.0625, (T)0.125, (T)0.0625}};

    VT ua[R1], ub[R1], uc[R1], b1[R2], bn[R2], b0[RS], wm[RS], wc[RS], wp[RS];
    const int t0 = z0 - 2;                                        // first step: the sweep of plane z0 - 1
#pragma unroll
    for (int rr = 0; rr < R1; rr++) { ua[rr] = LDU(t0, rr); ub[rr] = LDU(t0 + 1, rr); uc[rr] = LDU(t0 + 2, rr); }
#pragma unroll
    for (int q = 0; q < R2; q++) { b1[q] = LDB(t0 + 1, q, false); bn[q] = v16_zero<T>(); }
#pragma unroll
    for (int j = 0; j < RS; j++) { b0[j] = v16_zero<T>(); wm[j] = b0[j]; wc[j] = b0[j]; wp[j] = b0[j]; }
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int q = 0; q < R2; q++) {
            if (lane == 0) edgeW[(t0 + 1) & 1][q][w] = ub[q + 1].v[0];
            else edgeE[(t0 + 1) & 1][q][w] = ub[q + 1].v[VX - 1];
        }
    }
    T acc[NCR][NCJ], accn[NCR][NCJ];
#pragma unroll
    for (int cl = 0; cl < NCR; cl++)
#pragma unroll
        for (int q = 0; q < NCJ; q++) { acc[cl][q] = (T)0; accn[cl][q] = (T)0; }
    __syncthreads();

    for (int t = t0; t < z1; t++) {
        const int p = t + 1;                                      // plane the sweep produces in this step
#pragma unroll
        for (int q = 0; q < R2; q++) bn[q] = LDB(t + 2, q, q >= 3 && q <= R2 - 4);     // b of the next step; rows a neighbouring tile reads too (three on either side): cached
        // ---- the sweep of plane p on rows yb-1 .. yb+TY+1 ----
        {
            const bool pin = (p >= 0 && p < a.nz);
            const bool pst = (p >= z0 && p < zs1);
            const int eb = p & 1, cb = p & 1;
#pragma unroll
            for (int q = 0; q < R2; q++) {
                const int rr = q + 1;
                T Wv = lane_up<true>(ub[rr].v[VX - 1]), Ev = lane_dn<true>(ub[rr].v[0]);
                if (lane == 0) Wv = (w > 0) ? edgeE[eb][q][w - 1] : (T)0;
                if (lane == 63) Ev = (w < WX - 1) ? edgeW[eb][q][w + 1] : (T)0;
                VT o;
#pragma unroll
                for (int e = 0; e < VX; e++) {
                    const T wv = (e == 0) ? Wv : ub[rr].v[e - 1 < 0 ? 0 : e - 1];
                    const T ev = (e == VX - 1) ? Ev : ub[rr].v[e + 1 > VX - 1 ? VX - 1 : e + 1];
                    T s = a.a0 * ua[rr].v[e];
                    s = s + a.a1 * ub[rr - 1].v[e];
                    s = s + a.a2 * wv;
                    s = s + a.a3 * ub[rr].v[e];
                    s = s + a.a4 * ev;
                    s = s + a.a5 * ub[rr + 1].v[e];
                    s = s + a.a6 * uc[rr].v[e];
                    const T res = b1[q].v[e] - s;
                    const T zz = res * a.dinv;
                    o.v[e] = ub[rr].v[e] + a.scale * zz;
                    if (!pin || !s1ok[q] || (lastlane && e == VX - 1)) o.v[e] = (T)0;
                }
                *reinterpret_cast<VT *>(&cen[cb][q][xl + VX]) = o;
                if (q >= 1 && q <= RS) wp[q - 1] = o;
                if (q >= 1 && q <= TY) { if (pst && s1ok[q]) stv_stream(a.out + (long)p * a.ms + (long)(yb + q - 1) * a.rs + x0, o); }
            }
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int q = 0; q < R2; q++) {
                    if (lane == 0) edgeW[eb ^ 1][q][w] = uc[q + 1].v[0];
                    else edgeE[eb ^ 1][q][w] = uc[q + 1].v[VX - 1];
                }
            }
        }
        // plane t+3 of u into the registers of plane t (the sweep is done with them)
#pragma unroll
        for (int rr = 0; rr < R1; rr++) ua[rr] = LDU(t + 3, rr);
        // ---- residual of the swept plane t on rows yb .. yb+TY: x / y neighbours from cen[t & 1] (written in step t-1) ----
        VT res[RS];
        if (t >= z0) {
            const int cb = t & 1;
#pragma unroll
            for (int j = 0; j < RS; j++) {
                const int q = j + 1;
                const VT sv = *reinterpret_cast<const VT *>(&cen[cb][q - 1][xl + VX]);
                const VT nv = *reinterpret_cast<const VT *>(&cen[cb][q + 1][xl + VX]);
                const T Wv = cen[cb][q][xl + VX - 1], Ev = cen[cb][q][xl + 2 * VX];
#pragma unroll
                for (int e = 0; e < VX; e++) {
                    const T wv = (e == 0) ? Wv : wc[j].v[e - 1 < 0 ? 0 : e - 1];
                    const T ev = (e == VX - 1) ? Ev : wc[j].v[e + 1 > VX - 1 ? VX - 1 : e + 1];
                    T s = a.a0 * wm[j].v[e];
                    s = s + a.a1 * sv.v[e];
                    s = s + a.a2 * wv;
                    s = s + a.a3 * wc[j].v[e];
                    s = s + a.a4 * ev;
                    s = s + a.a5 * nv.v[e];
                    s = s + a.a6 * wp[j].v[e];
                    res[j].v[e] = rok[j] ? b0[j].v[e] - s : (T)0;
                }
                if (lastlane) res[j].v[VX - 1] = (T)0;
            }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < RS; j++) eR[cb][j][w] = res[j].v[0];
            }
        }
        __syncthreads();            // cen[(t+1)&1] = the swept plane t+1 complete; wave-edge values of this residual plane visible
        // ---- full weighting: running sums of the coarse planes t/2-1 (dk = 2) and t/2 (dk = 0) / (t-1)/2 (dk = 1) ----
        if (t >= z0) {
            const bool even = ((t & 1) == 0);
            const T wk = even ? (T)0.25 : (T)0.5;
            T nx_[RS];
#pragma unroll
            for (int j = 0; j < RS; j++) {
                nx_[j] = lane_dn<true>(res[j].v[0]);
                if (lane == 63) nx_[j] = (w < WX - 1) ? eR[t & 1][j][w + 1] : (T)0;
            }
#pragma unroll
            for (int cl = 0; cl < NCR; cl++) {
                if (!crow[cl]) continue;
#pragma unroll
                for (int q = 0; q < NCJ; q++) {
#pragma unroll
                    for (int di = 0; di < 3; di++) {
#pragma unroll
                        for (int dj = 0; dj < 3; dj++) {
                            const int e = 2 * q + dj;
                            const T val = (e < VX) ? res[2 * cl + di].v[e < VX ? e : 0] : nx_[2 * cl + di];
                            const T pr = (wk * w2[di][dj]) * val;
                            acc[cl][q] += pr;
                            if (even) accn[cl][q] += pr;
                        }
                    }
                }
            }
            if (even) {
                const int kc = t / 2 - 1;     // completed coarse plane
#pragma unroll
                for (int cl = 0; cl < NCR; cl++)
#pragma unroll
                    for (int q = 0; q < NCJ; q++) {
                        if (kc >= kc0 && crow[cl] && jc0 + q < a.nxc) {
                            const long oc = (long)kc * a.cms + (long)(NCR * ty + cl) * a.crs + jc0 + q;
                            a.bc[oc] = acc[cl][q];
                            if (a.uc0) { const T zq = acc[cl][q] * a.dinv_c; a.uc0[oc] = a.scale_c * zq; }
                        }
                        acc[cl][q] = accn[cl][q]; accn[cl][q] = (T)0;
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < RS; j++) { b0[j] = b1[j + 1]; wm[j] = wc[j]; wc[j] = wp[j]; }
#pragma unroll
        for (int q = 0; q < R2; q++) b1[q] = bn[q];
#pragma unroll
        for (int rr = 0; rr < R1; rr++) { VT tmpv = ua[rr]; ua[rr] = ub[rr]; ub[rr] = uc[rr]; uc[rr] = tmpv; }
    }
}

// Tiles of 4 rows (half the redundant row loads and first-sweep work of the 2-row tiles) do not fit in 256 registers with the swept
// planes t-1, t, t+1 of the lane's own columns in registers: k_srr4b keeps them in LDS -- a ring of three planes of the 5 residual
// rows plus the two outer rows of the centre plane, double buffered: 19 rows of 1028 doubles, 156 KB of the CU's 160 KB.
// Its instruction overhead is taken out (a first version spent a third of its vector instructions on addressing, masking and SGPR
// reloads; 888 -> 699 vector instructions per step, 5.45 -> 5.31 ms at 1023^3 -- the pass is bound by its HBM traffic, 28.2 B per
// unknown at ~5.7 TB/s, the re-reads of rows shared between tiles that miss the 4 MB L2 included):
//  * loads / stores through buffer descriptors (one per plane, built on the scalar unit): the row offset is a 32-bit SGPR operand, the
//    lane offset one constant VGPR -- no 64-bit vector address arithmetic, no spilled 64-bit row offsets (was 3 vector instructions
//    and 2 VGPRs per load);
//  * wave-edge neighbours: the DPP shift keeps the `old` operand in the lane that has no source, so the value from the neighbouring
//    wave (or the zero pad at the ends of the row) is passed as `old`: no lane-0 / lane-63 selects;
//  * rows / planes outside the grid are skipped by wave-uniform branches instead of per-element selects.
template <int WX>
__global__ void __launch_bounds__(64 * WX) k_srr4b(const SRRArgs<double> a) {
    typedef double T;
    constexpr int TY = 4;
    constexpr int VX = 2, NCR = TY / 2, RS = TY + 1, R2 = TY + 3, R1 = TY + 5;
    constexpr int TX = 64 * VX * WX, LW = TX + 2 * VX;
    __shared__ __attribute__((aligned(16))) T ring[3][RS][LW];       // swept plane p, rows yb .. yb+TY, in slot p % 3
    __shared__ __attribute__((aligned(16))) T halo[2][2][LW];        // its rows yb-1 and yb+TY+1 (read while it is the centre plane), slot p & 1
    // wave-edge values: [.][.][0] of edgeE and [.][.][WX] of edgeW / eR are zero pads (the ends of the row)
    __shared__ T edgeW[2][R2][WX + 1], edgeE[2][R2][WX + 1], eR[2][RS][WX + 1];
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int yb = TY * ty;
    const int kc0 = a.kcbeg + tz * a.kcc, kc1 = min(kc0 + a.kcc, a.kcend);
    if (kc0 >= kc1) return;
    const int pmin = a.has_lo ? -2 : -1, pmax = a.has_hi ? a.nz + 2 : a.nz;      // u planes that exist
    const int smin = a.has_lo ? -1 : 0, smax = a.has_hi ? a.nz + 1 : a.nz - 1;   // planes on which the sweep is real
    const int z0 = 2 * kc0, z1 = min(2 * kc1 + 1, a.has_hi ? a.nz + 1 : a.nz);   // planes whose residual this chunk forms: [z0, z1)
    const int zs1 = (kc1 == a.nzc) ? a.nz : 2 * kc1;              // planes of the swept field this chunk stores: [z0, zs1)
    const int xl = VX * tid;
    const bool lastlane = (tid == 64 * WX - 1);                   // its last element is the ghost column x = nx: stays 0
    const unsigned lb = (unsigned)(xl * (int)sizeof(T));
    // rows q of the sweep (y = yb - 1 + q) inside the grid: qlo <= q <= qhi.  Rows outside are never written: their LDS rows keep the
    // zeros of the initial fill.  (Two integers instead of a predicate per row: the kernel is short of scalar registers.)
    const int qlo = max(0, 1 - yb), qhi = min(R2 - 1, a.ny - yb);
    const bool crow0 = (NCR * ty < a.nyc), crow1 = (NCR * ty + 1 < a.nyc);
    // byte offsets of the rows from row -1 of a plane (rows beyond the ghost rows alias them: whatever such a load brings in only
    // reaches sweep values that are skipped)
    const unsigned rowb = (unsigned)(a.rs * (long)sizeof(T));
    const unsigned plane_bytes = (unsigned)(a.ny + 2) * rowb;
    unsigned urb[R1], brb[R2];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) urb[rr] = (unsigned)(max(-1, min(yb - 2 + rr, a.ny)) + 1) * rowb;
#pragma unroll
    for (int q = 0; q < R2; q++) brb[q] = (unsigned)(max(0, min(yb - 1 + q, a.ny)) + 1) * rowb;
    auto URS = [&](int p) {
        const int pp = max(pmin, min(p, pmax));
        const T *pl = (pp == -2) ? a.far_lo : (pp == a.nz + 1) ? a.far_hi : (pp == a.nz + 2) ? a.far2_hi : a.u + (long)pp * a.ms;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(pl - a.rs), 0, (int)plane_bytes, 0x00020000);
    };
    auto BRS = [&](int p) {
        const int pp = max(smin, min(p, smax));
        const T *pl = (pp == a.nz + 1) ? a.bfar_hi : a.b + (long)pp * a.ms;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(pl - a.rs), 0, (int)plane_bytes, 0x00020000);
    };
    for (int i = tid; i < 3 * RS * LW; i += 64 * WX) (&ring[0][0][0])[i] = (T)0;
    for (int i = tid; i < 2 * 2 * LW; i += 64 * WX) (&halo[0][0][0])[i] = (T)0;
    for (int i = tid; i < 2 * R2 * (WX + 1); i += 64 * WX) { (&edgeW[0][0][0])[i] = (T)0; (&edgeE[0][0][0])[i] = (T)0; }
    for (int i = tid; i < 2 * RS * (WX + 1); i += 64 * WX) (&eR[0][0][0])[i] = (T)0;
    const int jc0 = tid;
    const T w2[3][3] = {{(T)0.0625, (T)0.125, (T)0.0625}, {(T)0.125, (T)0.25, (T)0.125}, {(T)0.0625, (T)0.125, (T)0.0625}};

    // u planes: rows yb-1 .. yb+TY+1 (the rows of the sweep, index q); the two outer rows yb-2 / yb+TY+2 are read of the CENTRE plane
    // only (its y neighbours of the first / last sweep row): they are loaded one step later into h0 / h8
    VT ua[R2], ub[R2], uc[R2], h0, h8, b1[R2], bn[R2], b0[RS];
    const int t0 = z0 - 2;                                        // first step: the sweep of plane z0 - 1
    {
        const auto r0 = URS(t0), r1 = URS(t0 + 1), r2 = URS(t0 + 2);
        const auto rb = BRS(t0 + 1);
#pragma unroll
        for (int q = 0; q < R2; q++) { ua[q] = bufld<T>(r0, lb, urb[q + 1]); ub[q] = bufld<T>(r1, lb, urb[q + 1]); uc[q] = bufld<T>(r2, lb, urb[q + 1]); }
        h0 = bufld<T>(r1, lb, urb[0]); h8 = bufld<T>(r1, lb, urb[R1 - 1]);
#pragma unroll
        for (int q = 0; q < R2; q++) { b1[q] = bufld<T>(rb, lb, brb[q]); bn[q] = v16_zero<T>(); }
    }
#pragma unroll
    for (int j = 0; j < RS; j++) b0[j] = v16_zero<T>();
    __syncthreads();                                              // the zero fill is complete before the first edge values land
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int q = 0; q < R2; q++) {
            if (lane == 0) edgeW[(t0 + 1) & 1][q][w] = ub[q].v[0];
            else edgeE[(t0 + 1) & 1][q][w + 1] = ub[q].v[VX - 1];
        }
    }
    T acc0 = (T)0, acc1 = (T)0, accn0 = (T)0, accn1 = (T)0;
    __syncthreads();

    // one marching step; A: plane t, B: t+1, C: t+2 of u; A is reloaded with plane t+3.  (Unrolling the loop by three with the
    // roles permuted instead of copying the planes spills at 8 waves and is no faster: the pass is not issue bound.)
    auto step = [&](VT (&A)[R2], VT (&B)[R2], VT (&C)[R2], const int t) __attribute__((always_inline)) {
        const int p = t + 1;                                      // plane the sweep produces in this step
        {
            const auto rb = BRS(t + 2);
#pragma unroll
            for (int q = 0; q < R2; q++) bn[q] = (q == 3) ? bufld_nt<T>(rb, lb, brb[q]) : bufld<T>(rb, lb, brb[q]);     // rows a neighbouring tile reads too: cached
        }
        // ---- the sweep of plane p on rows yb-1 .. yb+TY+1 ----
        {
            const bool pin = (p >= smin && p <= smax);
            const bool pst = (p >= z0 && p < zs1);
            const int eb = p & 1, hb = p & 1, sl = (p + 3) % 3;
            // (loop-variant bounds: planes outside the grid have no valid row)
            const int qlo_t = pin ? qlo : R2;
            const unsigned span_t = (unsigned)(qhi - qlo_t);
            const auto ro = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out - a.rs + (long)max(0, min(p, a.nz - 1)) * a.ms), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
            for (int q = 0; q < R2; q++) {
                if ((unsigned)(q - qlo_t) <= span_t && qhi >= qlo_t) {      // wave-uniform: the row is inside the grid
                    const T Wv = lane_up_old(B[q].v[VX - 1], edgeE[eb][q][w]);
                    const T Ev = lane_dn_old(B[q].v[0], edgeW[eb][q][w + 1]);
                    const VT &Sr = (q == 0) ? h0 : B[q > 0 ? q - 1 : 0];
                    const VT &Nr = (q == R2 - 1) ? h8 : B[q < R2 - 1 ? q + 1 : q];
                    VT o;
#pragma unroll
                    for (int e = 0; e < VX; e++) {
                        const T wv = (e == 0) ? Wv : B[q].v[e - 1 < 0 ? 0 : e - 1];
                        const T ev = (e == VX - 1) ? Ev : B[q].v[e + 1 > VX - 1 ? VX - 1 : e + 1];
                        T s = a.a0 * A[q].v[e];
                        s = s + a.a1 * Sr.v[e];
                        s = s + a.a2 * wv;
                        s = s + a.a3 * B[q].v[e];
                        s = s + a.a4 * ev;
                        s = s + a.a5 * Nr.v[e];
                        s = s + a.a6 * C[q].v[e];
                        const T res = b1[q].v[e] - s;
                        const T zz = res * a.dinv;
                        o.v[e] = B[q].v[e] + a.scale * zz;
                    }
                    if (lastlane) o.v[VX - 1] = (T)0;
                    if (q == 0) *reinterpret_cast<VT *>(&halo[hb][0][xl + VX]) = o;
                    else if (q == R2 - 1) *reinterpret_cast<VT *>(&halo[hb][1][xl + VX]) = o;
                    else *reinterpret_cast<VT *>(&ring[sl][q - 1][xl + VX]) = o;
                    if (q >= 1 && q <= TY) { if (pst) bufst_nt<T>(o, ro, lb, (unsigned)(yb + q) * rowb); }
                } else if (!pin && q >= 1 && q <= RS) {
                    // plane -1 / nz: zeros where the residual of the plane next to it reads the lane's own columns
                    *reinterpret_cast<VT *>(&ring[sl][q - 1][xl + VX]) = v16_zero<T>();
                }
            }
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int q = 0; q < R2; q++) {
                    if (lane == 0) edgeW[eb ^ 1][q][w] = C[q].v[0];
                    else edgeE[eb ^ 1][q][w + 1] = C[q].v[VX - 1];
                }
            }
        }
        // plane t+3 of u into the registers of plane t (the sweep is done with them)
        {
            const auto r3 = URS(t + 3), r2 = URS(t + 2);
#pragma unroll
            for (int q = 0; q < R2; q++) A[q] = bufld<T>(r3, lb, urb[q + 1]);
            h0 = bufld<T>(r2, lb, urb[0]); h8 = bufld<T>(r2, lb, urb[R1 - 1]);      // outer rows of the next centre plane
        }
        // ---- residual of the swept plane t on rows yb .. yb+TY: everything from LDS.  (Rows beyond the grid give values that no
        //      valid coarse row reads, and the ghost column of the last lane is not read either: no masking here.) ----
        VT res[RS];
        if (t >= z0) {
            const int cb = t & 1, sm = (t + 2) % 3, sc = t % 3, sp = (t + 1) % 3;
#pragma unroll
            for (int j = 0; j < RS; j++) {
                const VT cv = *reinterpret_cast<const VT *>(&ring[sc][j][xl + VX]);
                const VT dn = *reinterpret_cast<const VT *>(&ring[sm][j][xl + VX]);
                const VT upv = *reinterpret_cast<const VT *>(&ring[sp][j][xl + VX]);
                const VT sv = (j == 0) ? *reinterpret_cast<const VT *>(&halo[cb][0][xl + VX]) : *reinterpret_cast<const VT *>(&ring[sc][j > 0 ? j - 1 : 0][xl + VX]);
                const VT nv = (j == RS - 1) ? *reinterpret_cast<const VT *>(&halo[cb][1][xl + VX]) : *reinterpret_cast<const VT *>(&ring[sc][j < RS - 1 ? j + 1 : j][xl + VX]);
                const T Wv = ring[sc][j][xl + VX - 1], Ev = ring[sc][j][xl + 2 * VX];
#pragma unroll
                for (int e = 0; e < VX; e++) {
                    const T wv = (e == 0) ? Wv : cv.v[e - 1 < 0 ? 0 : e - 1];
                    const T ev = (e == VX - 1) ? Ev : cv.v[e + 1 > VX - 1 ? VX - 1 : e + 1];
                    T s = a.a0 * dn.v[e];
                    s = s + a.a1 * sv.v[e];
                    s = s + a.a2 * wv;
                    s = s + a.a3 * cv.v[e];
                    s = s + a.a4 * ev;
                    s = s + a.a5 * nv.v[e];
                    s = s + a.a6 * upv.v[e];
                    res[j].v[e] = b0[j].v[e] - s;
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < RS; j++) eR[cb][j][w] = res[j].v[0];
            }
        }
        __syncthreads();            // the swept plane t+1 complete in LDS; wave-edge values of this residual plane visible
        // ---- full weighting: running sums of the coarse planes t/2-1 (dk = 2) and t/2 (dk = 0) / (t-1)/2 (dk = 1) ----
        if (t >= z0) {
            const bool even = ((t & 1) == 0);
            const T wk = even ? (T)0.25 : (T)0.5;
            T nx_[RS];
#pragma unroll
            for (int j = 0; j < RS; j++) nx_[j] = lane_dn_old(res[j].v[0], eR[t & 1][j][w + 1]);
#pragma unroll
            for (int cl = 0; cl < NCR; cl++) {
                if (!(cl == 0 ? crow0 : crow1)) continue;
                T sa = (cl == 0) ? acc0 : acc1, sn = (cl == 0) ? accn0 : accn1;
#pragma unroll
                for (int di = 0; di < 3; di++) {
#pragma unroll
                    for (int dj = 0; dj < 3; dj++) {
                        const T val = (dj < VX) ? res[2 * cl + di].v[dj < VX ? dj : 0] : nx_[2 * cl + di];
                        const T pr = (wk * w2[di][dj]) * val;
                        sa += pr;
                        if (even) sn += pr;
                    }
                }
                if (cl == 0) { acc0 = sa; accn0 = sn; } else { acc1 = sa; accn1 = sn; }
            }
            if (even) {
                const int kc = t / 2 - 1;     // completed coarse plane
                if (kc >= kc0 && jc0 < a.nxc) {
                    if (crow0) {
                        const long oc = (long)kc * a.cms + (long)(NCR * ty) * a.crs + jc0;
                        a.bc[oc] = acc0;
                        if (a.uc0) { const T zq = acc0 * a.dinv_c; a.uc0[oc] = a.scale_c * zq; }
                    }
                    if (crow1) {
                        const long oc = (long)kc * a.cms + (long)(NCR * ty + 1) * a.crs + jc0;
                        a.bc[oc] = acc1;
                        if (a.uc0) { const T zq = acc1 * a.dinv_c; a.uc0[oc] = a.scale_c * zq; }
                    }
                }
                acc0 = accn0; accn0 = (T)0; acc1 = accn1; accn1 = (T)0;
            }
        }
#pragma unroll
        for (int j = 0; j < RS; j++) b0[j] = b1[j + 1];
#pragma unroll
        for (int q = 0; q < R2; q++) b1[q] = bn[q];
    };
    for (int t = t0; t < z1; t++) {
        step(ua, ub, uc, t);
#pragma unroll
        for (int q = 0; q < R2; q++) { VT tmpv = ua[q]; ua[q] = ub[q]; ub[q] = uc[q]; uc[q] = tmpv; }
    }
}

// shapes the fused sweep + residual + restriction is built for: full rows of 1 / 2 / 4 / 8 waves, whole 3-D grid
template <typename T>
static bool srr_shape_ok(const mgk_geom *gf, const mgk_geom *gc) {
    constexpr int VX = 16 / sizeof(T);
    if (!gf || !gc || gf->dim != 3 || gc->dim != 3) return false;
    if (gf->nx != 2 * gc->nx + 1 || gf->ny != 2 * gc->ny + 1 || gf->nz != 2 * gc->nz + 1) return false;
    const int wr = 64 * VX;
    if ((gf->nx + 1) % wr != 0 || (gf->ny + 1) % 4 != 0) return false;
    const int w = (gf->nx + 1) / wr;
    return w == 1 || w == 2 || w == 4 || w == 8;
}
extern "C" int mgk_sweep_residual_restrict_ok_f64(const mgk_geom *gf, const mgk_geom *gc) { return srr_shape_ok<double>(gf, gc) ? 1 : 0; }

template <typename T, int TY>
static void launch_srr(int w, unsigned nblk, hipStream_t s, const SRRArgs<T> &a) {
    if (w <= 1) hipLaunchKernelGGL((k_srr<T, 1, TY>), dim3(nblk), dim3(64), 0, s, a);
    else if (w <= 2) hipLaunchKernelGGL((k_srr<T, 2, TY>), dim3(nblk), dim3(128), 0, s, a);
    else if (w <= 4) hipLaunchKernelGGL((k_srr<T, 4, TY>), dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_srr<T, 8, TY>), dim3(nblk), dim3(512), 0, s, a);
}
// out = J(u); bc = R (b - A out); uc0 (optional) = scale_c * (bc * dinv_c)
static int sweep_residual_restrict(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                   const double *b, const double *u, double *unew, double *bc, double *uc0, double dinv_c, double scale_c,
                                   const double *far_lo, const double *far_hi, const double *far2_hi, const double *bfar_hi,
                                   int kcbeg, int kcend, void *stream) {
    typedef double T;
    const bool slab = far_lo || far_hi;
    if (!c || !gf || !gc || !coef || !b || !u || !unew || u == unew || !bc) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_f64: bad arguments");
    if (kcbeg < 0 || kcend > gc->nz || kcbeg >= kcend) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict: empty or out-of-range coarse plane range");
    SRRArgs<T> a; memset(&a, 0, sizeof(a));
    a.u = u + gf->org; a.b = b + gf->org; a.out = unew + gf->org; a.bc = bc + gc->org; a.uc0 = uc0 ? uc0 + gc->org : nullptr;
    a.nx = gf->nx; a.ny = gf->ny; a.nz = gf->nz; a.nxc = gc->nx; a.nyc = gc->ny; a.nzc = gc->nz;
    a.rs = gf->pitch; a.ms = gf->plane; a.crs = gc->pitch; a.cms = gc->plane;
    a.a0 = coef[0]; a.a1 = coef[1]; a.a2 = coef[2]; a.a3 = coef[3]; a.a4 = coef[4]; a.a5 = coef[5]; a.a6 = coef[6];
    a.dinv = dinv; a.scale = scale; a.dinv_c = dinv_c; a.scale_c = scale_c;
    a.far_lo = far_lo; a.far_hi = far_hi; a.far2_hi = far2_hi; a.bfar_hi = bfar_hi;
    a.has_lo = far_lo != nullptr; a.has_hi = far_hi != nullptr;
    a.kcbeg = kcbeg; a.kcend = kcend;
    constexpr int VX = 16 / sizeof(T);
    const int w = (gf->nx + 1) / (64 * VX);
    const int nkc = kcend - kcbeg;
    // tiles of 4 rows with the swept planes in LDS (1023^3: 5.3 ms against 6.2 ms for tiles of 2 rows with them in registers;
    // 511^3: 0.72 against 0.82; tiles of 4 rows all in registers need 316 VGPRs at 8 waves).  Rows of <= 2 waves (n <= 255): tiles of
    // 2 rows, two blocks per CU -- 255^3 0.119 against 0.136 ms.  Tuning variants: 40 forces tiles of 2 rows, 41 tiles of 4
    const int TYsel = slab ? 4 : (g_variant == 40) ? 2 : (g_variant == 41) ? 4 : (w >= 4 ? 4 : 2);
    a.nty = (gf->ny + TYsel - 1) / TYsel;
    // blocks: a multiple of what the chip holds at once (512-thread blocks: one per CU); every chunk recomputes three planes
    // (rows of one wave, 127^3: 2048 one-wave blocks of 2 coarse planes instead of 1024 of 4 -- 34.2 -> 28.5 us inside the 257^3 cycle)
    const long target = (w > 4) ? (TYsel == 4 ? 256 : 512) : (w <= 1 ? 2048 : 1024);
    long nch = (a.nty >= target) ? 1 : (target + a.nty - 1) / a.nty;
    if (g_zchunk > 0) nch = (nkc + g_zchunk - 1) / g_zchunk;
    else if (slab && c->chunk_planes > 0 && nch < (2 * nkc + c->chunk_planes - 1) / c->chunk_planes) nch = (2 * nkc + c->chunk_planes - 1) / c->chunk_planes;
    int kcc = (int)((nkc + nch - 1) / nch);
    const int kmin = (w <= 1) ? 2 : 4;
    if (kcc < kmin && g_zchunk <= 0) kcc = kmin;
    if (kcc > nkc) kcc = nkc;
    a.kcc = kcc;
    const unsigned nblk = (unsigned)(a.nty * ((nkc + kcc - 1) / kcc));
    if (TYsel == 4) {
        hipStream_t st = S(c, stream);
        if (w <= 1) hipLaunchKernelGGL((k_srr4b<1>), dim3(nblk), dim3(64), 0, st, a);
        else if (w <= 2) hipLaunchKernelGGL((k_srr4b<2>), dim3(nblk), dim3(128), 0, st, a);
        else if (w <= 4) hipLaunchKernelGGL((k_srr4b<4>), dim3(nblk), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_srr4b<8>), dim3(nblk), dim3(512), 0, st, a);
    } else launch_srr<T, 2>(w, nblk, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_sweep_residual_restrict_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                               const double *b, const double *u, double *unew, double *bc, double *uc0,
                                               double dinv_c, double scale_c, void *stream) {
    if (!srr_shape_ok<double>(gf, gc)) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_f64: built for full-row 3-D shapes (n = 127, 255, 511, 1023), whole grid");
    return sweep_residual_restrict(c, gf, gc, coef, dinv, scale, b, u, unew, bc, uc0, dinv_c, scale_c, nullptr, nullptr, nullptr, nullptr, 0, gc->nz, stream);
}
// The same on a z-slab, coarse planes [kcbeg, kcend) of the slab's share.  `far` (geometry gfar = (nx, ny, 2), see mgk_jacobi2_slab_*):
// lo ghost = plane nz-2 of the rank below, hi ghost = plane 1 of the rank above; `far2`: hi ghost = plane 2 of the rank above; `bfar`:
// hi ghost = plane 1 of b of the rank above; u's and b's own ghost planes hold the neighbours' last / first plane.  A slab with a rank
// above has nzf = 2 nzc and evaluates the sweep on the planes nz, nz+1 and the residual of plane nz itself (same operands, same
// arithmetic as the owner: same bits), so its last coarse plane is complete; the last slab has nzf = 2 nzc + 1.
extern "C" int mgk_sweep_residual_restrict_slab_ok_f64(const mgk_geom *gf, const mgk_geom *gc) {
    if (!gf || !gc || gf->dim != 3 || gc->dim != 3 || gf->nx != 2 * gc->nx + 1 || gf->ny != 2 * gc->ny + 1) return 0;
    if (gf->nz != 2 * gc->nz && gf->nz != 2 * gc->nz + 1) return 0;
    if ((gf->nx + 1) % 128 != 0 || (gf->ny + 1) % 4 != 0) return 0;
    const int w = (gf->nx + 1) / 128;
    return (w == 1 || w == 2 || w == 4 || w == 8) && gc->nz >= 1 && gf->nz >= 4;
}
extern "C" int mgk_sweep_residual_restrict_slab_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *coef,
                                                    double dinv, double scale, const double *b, const double *u, double *unew,
                                                    const double *far, const double *far2, const double *bfar, int has_lo, int has_hi,
                                                    double *bc, int kcbeg, int kcend, void *stream) {
    if (!mgk_sweep_residual_restrict_slab_ok_f64(gf, gc)) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64: shape not built");
    if ((has_lo || has_hi) && (!gfar || !far || gfar->dim != 3 || gfar->nz != 2 || gfar->nx != gf->nx || gfar->ny != gf->ny || gfar->pitch != gf->pitch))
        return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64: the far-plane fields must have the geometry (nx, ny, 2) of the slab");
    if (has_hi && (!far2 || !bfar || gf->nz != 2 * gc->nz)) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64: a slab with a rank above has nzf = 2 nzc and needs far2 / bfar");
    if (!has_hi && gf->nz != 2 * gc->nz + 1) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_slab_f64: the last slab has nzf = 2 nzc + 1");
    const double *lo = has_lo ? far + gfar->org - gfar->plane : nullptr;
    const double *hi = has_hi ? far + gfar->org + 2 * gfar->plane : nullptr;
    const double *hi2 = has_hi ? far2 + gfar->org + 2 * gfar->plane : nullptr;
    const double *bhi = has_hi ? bfar + gfar->org + 2 * gfar->plane : nullptr;
    return sweep_residual_restrict(c, gf, gc, coef, dinv, scale, b, u, unew, bc, nullptr, 0.0, 0.0, lo, hi, hi2, bhi, kcbeg, kcend, stream);
}

// 2-D version of the two-sweep pass: blocks of 64*WX lanes march along y over an x tile of 128*WX columns; a lane owns one
// column pair, rows y-1 .. y+2 of u and the three live rows of the first sweep stay in its registers, x neighbours come by
// wave shuffle (wave edges through LDS, written one step ahead: one barrier per row).  Tiles overlap by one lane on either
// side: every lane makes the first sweep, the inner lanes the second one.
struct J2dArgs {
    const double *u, *b;
    double *out;
    int nx, ny;
    long rs;
    int ntx, yc;
    double a0, a2, a3, a4, a6, dinv, scale;
    const double *ctab, *dtab;      // optional (stretched meshes): 5 coefficients {(i-1), W, C, E, (i+1)} and 1/diag per grid row
    double *partials;               // NORM: per-block partial of || b - A u ||^2 of the INPUT field
};
template <int WX, bool NORM = false>
__global__ void __launch_bounds__(64 * WX) k_jacobi2_2d(const J2dArgs a) {
    constexpr int VX = 2, NT = 64 * WX, TXE = VX * (NT - 2);          // columns a block produces
    __shared__ double eUW[2][WX], eUE[2][WX], ePW[2][WX], ePE[2][WX];   // wave-edge values of u(row) / u'(row), double buffered
    using VT = V16<double>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tx = blockIdx.x % a.ntx, tyc = blockIdx.x / a.ntx;
    const int x0 = tx * TXE - VX + VX * tid;                          // may be -2 (left of the grid) on the first lane
    const int y0 = tyc * a.yc, y1 = min(y0 + a.yc, a.ny);
    if (y0 >= y1) { if (NORM && tid == 0) a.partials[blockIdx.x] = 0.0; return; }
    double nacc = 0.0;
    const bool xin = x0 >= 0 && x0 < a.nx;                            // the pair holds at least one grid column
    const bool lastvec = (x0 + VX > a.nx);
    const bool outl = xin && tid >= 1 && tid <= NT - 2;               // this lane stores second-sweep values
    const double *up_ = a.u + x0, *bp_ = a.b + x0;
    auto ldu = [&](int r) -> VT { return ldv(up_ + (long)r * a.rs, xin && r >= -1 && r <= a.ny); };
    auto ldb = [&](int r) -> VT { return ldv(bp_ + (long)r * a.rs, xin && r >= 0 && r < a.ny); };

    const int t0 = y0 - 2;
    VT ua = ldu(t0), ub = ldu(t0 + 1), uc = ldu(t0 + 2), ud = v16_zero<double>();
    VT b1 = ldb(t0 + 1), bn = v16_zero<double>(), b0 = bn;
    VT wm = bn, wc = bn, wp = bn;                                     // first-sweep rows t-1, t, t+1
    if (lane == 0) { eUW[(t0 + 1) & 1][w] = ub.v[0]; ePW[t0 & 1][w] = 0.0; }
    if (lane == 63) { eUE[(t0 + 1) & 1][w] = ub.v[VX - 1]; ePE[t0 & 1][w] = 0.0; }
    __syncthreads();
    for (int t = t0; t < y1; t++) {
        const int p = t + 1;
        ud = ldu(t + 3);
        bn = (t + 1 < y1) ? ldb(t + 2) : v16_zero<double>();
        // coefficients of the two rows worked on in this step: launch constants, or the table rows p (first sweep) and t (second)
        double p0 = a.a0, p2 = a.a2, p3 = a.a3, p4 = a.a4, p6 = a.a6, pd = a.dinv, q0 = a.a0, q2 = a.a2, q3 = a.a3, q4 = a.a4, q6 = a.a6, qd = a.dinv;
        if (a.ctab) {
            const double *cp = a.ctab + 5 * (long)min(max(p, 0), a.ny - 1), *cq = a.ctab + 5 * (long)min(max(t, 0), a.ny - 1);
            p0 = cp[0]; p2 = cp[1]; p3 = cp[2]; p4 = cp[3]; p6 = cp[4]; pd = a.dtab[min(max(p, 0), a.ny - 1)];
            q0 = cq[0]; q2 = cq[1]; q3 = cq[2]; q4 = cq[3]; q6 = cq[4]; qd = a.dtab[min(max(t, 0), a.ny - 1)];
        }
        // ---- first sweep of row p ----
        {
            double Wv = __shfl_up(ub.v[VX - 1], 1, 64), Ev = __shfl_down(ub.v[0], 1, 64);
            if (lane == 0) Wv = (w > 0) ? eUE[p & 1][w - 1] : 0.0;
            if (lane == 63) Ev = (w < WX - 1) ? eUW[p & 1][w + 1] : 0.0;
            const bool pin = (p >= 0 && p < a.ny);
#pragma unroll
            for (int e = 0; e < VX; e++) {
                const double wv = (e == 0) ? Wv : ub.v[0];
                const double ev = (e == VX - 1) ? Ev : ub.v[VX - 1];
                double s = p0 * ua.v[e];
                s = s + p2 * wv;
                s = s + p3 * ub.v[e];
                s = s + p4 * ev;
                s = s + p6 * uc.v[e];
                const double res = b1.v[e] - s;
                const double zz = res * pd;
                wp.v[e] = ub.v[e] + a.scale * zz;
                if (!pin || !xin || (lastvec && x0 + e >= a.nx)) wp.v[e] = 0.0;
                if (NORM) nacc += (outl && p >= y0 && p < y1 && !(lastvec && x0 + e >= a.nx)) ? res * res : 0.0;     // points this block owns
            }
        }
        // ---- second sweep of row t (x neighbours of u'(t): shuffles of wc, wave edges written in the previous step) ----
        if (t >= y0) {
            double Wv = __shfl_up(wc.v[VX - 1], 1, 64), Ev = __shfl_down(wc.v[0], 1, 64);
            if (lane == 0) Wv = (w > 0) ? ePE[t & 1][w - 1] : 0.0;
            if (lane == 63) Ev = (w < WX - 1) ? ePW[t & 1][w + 1] : 0.0;
            VT o;
#pragma unroll
            for (int e = 0; e < VX; e++) {
                const double wv = (e == 0) ? Wv : wc.v[0];
                const double ev = (e == VX - 1) ? Ev : wc.v[VX - 1];
                double s = q0 * wm.v[e];
                s = s + q2 * wv;
                s = s + q3 * wc.v[e];
                s = s + q4 * ev;
                s = s + q6 * wp.v[e];
                const double res = b0.v[e] - s;
                const double zz = res * qd;
                o.v[e] = wc.v[e] + a.scale * zz;
                if (lastvec && x0 + e >= a.nx) o.v[e] = 0.0;
            }
            if (outl) stv_policy(a.out + (long)t * a.rs + x0, o, mgk_store_nt_2d(a.ny, a.rs));
        }
        // wave edges for the next step: u(row p+1) = uc, u'(row p) = wp
        if (lane == 0) { eUW[(p + 1) & 1][w] = uc.v[0]; ePW[p & 1][w] = wp.v[0]; }
        if (lane == 63) { eUE[(p + 1) & 1][w] = uc.v[VX - 1]; ePE[p & 1][w] = wp.v[VX - 1]; }
        __syncthreads();
        b0 = b1; b1 = bn; wm = wc; wc = wp; ua = ub; ub = uc; uc = ud;
    }
    if (NORM) {
        __shared__ double red[16];
        const double sblk = block_sum(nacc, red);
        if (tid == 0) a.partials[blockIdx.x] = sblk;
    }
}
static int jacobi2_2d(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                      const double *b, const double *u, double *unew, void *stream, int *norm_parts = nullptr) {
    if (!c || !g || (!coef && !ctab) || (ctab && !dtab) || !b || !u || !unew || u == unew || g->dim != 2) return fail(MGK_EINVAL, "mgk_jacobi2_2d: bad arguments (2-D)");
    J2dArgs a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org;
    a.nx = g->nx; a.ny = g->ny; a.rs = g->pitch;
    if (coef) { a.a0 = coef[0]; a.a2 = coef[1]; a.a3 = coef[2]; a.a4 = coef[3]; a.a6 = coef[4]; }
    a.dinv = dinv; a.scale = scale; a.ctab = ctab; a.dtab = dtab;
    constexpr int WX = 4, TXE = 2 * (64 * WX - 2);
    a.ntx = (g->nx + TXE - 1) / TXE;
    long nch = (2048 + a.ntx - 1) / a.ntx;
    if (g_zchunk > 0) nch = (g->ny + g_zchunk - 1) / g_zchunk;
    int yc = (int)((g->ny + nch - 1) / nch);
    if (yc < 16) yc = 16;
    if (yc > g->ny) yc = g->ny;
    a.yc = yc;
    const long nty = (g->ny + yc - 1) / yc;
    if (norm_parts) {
        if (a.ntx * nty > c->max_partials) return fail(MGK_EINVAL, "mgk_jacobi2_2d_sumsq_f64: more blocks than partial slots");
        a.partials = c->partials;
        hipLaunchKernelGGL((k_jacobi2_2d<WX, true>), dim3((unsigned)(a.ntx * nty)), dim3(64 * WX), 0, S(c, stream), a);
        *norm_parts = (int)(a.ntx * nty);
    } else hipLaunchKernelGGL((k_jacobi2_2d<WX>), dim3((unsigned)(a.ntx * nty)), dim3(64 * WX), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_jacobi2_2d_sumsq_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *dtab, double scale,
                                                const double *b, const double *u, double *unew, double *sumsq_host, void *stream) {
    if (!ctab || !sumsq_host) return fail(MGK_EINVAL, "mgk_jacobi2_2d_sumsq_rowcoef_f64: bad arguments");
    int nparts = 0;
    int rc = jacobi2_2d(c, g, nullptr, 1.0, scale, ctab, dtab, b, u, unew, stream, &nparts);
    if (rc) return rc;
    return finish_to_host(c, nparts, 1, S(c, stream), sumsq_host);
}
// two sweeps and || b - A u ||^2 of the input field (2-D form of mgk_jacobi2_sumsq_f64)
extern "C" int mgk_jacobi2_2d_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                        const double *b, const double *u, double *unew, double *sumsq_host, void *stream) {
    if (!coef || !sumsq_host) return fail(MGK_EINVAL, "mgk_jacobi2_2d_sumsq_f64: bad arguments");
    int nparts = 0;
    int rc = jacobi2_2d(c, g, coef, dinv, scale, nullptr, nullptr, b, u, unew, stream, &nparts);
    if (rc) return rc;
    return finish_to_host(c, nparts, 1, S(c, stream), sumsq_host);
}
extern "C" int mgk_jacobi2_2d_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                  const double *b, const double *u, double *unew, void *stream) {
    return jacobi2_2d(c, g, coef, dinv, scale, nullptr, nullptr, b, u, unew, stream);
}
extern "C" int mgk_jacobi2_2d_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *dtab, double scale,
                                          const double *b, const double *u, double *unew, void *stream) {
    if (!ctab) return fail(MGK_EINVAL, "mgk_jacobi2_2d_rowcoef_f64: null table");
    return jacobi2_2d(c, g, nullptr, 1.0, scale, ctab, dtab, b, u, unew, stream);
}

// ------------------------------------------------------------------------------------------
// row-wise elementwise / reduction kernels: one lane per aligned x pair, blocks stride over rows
// ------------------------------------------------------------------------------------------
// every block of a row kernel owns a CONTIGUOUS range of rows (long sequential streams per block keep
// HBM pages open; measured +10..15 % over a row-strided assignment at 1023^3)
#define ROW_RANGE(NROWS)                                                                   \
    const long rpb_ = ((NROWS) + gridDim.y - 1) / gridDim.y;                               \
    const long row0_ = (long)blockIdx.y * rpb_;                                            \
    const long row1_ = (row0_ + rpb_ < (NROWS)) ? row0_ + rpb_ : (NROWS);

struct RowArgs {
    int nx, ny, nz, npairs;
    long pitch, plane;
    long nrows;           // ny*nz
};
__device__ __forceinline__ long row_offset(const RowArgs &a, long row) {
    const long k = row / a.ny, i = row - k * a.ny;
    return k * a.plane + i * a.pitch;
}

template <typename T>
__global__ void __launch_bounds__(256) k_jacobi_zero(RowArgs a, T dinv, T scale, const T *b, T *out, const T *dtab) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.npairs) return;
    const int x0 = 2 * p;
    ROW_RANGE(a.nrows) for (long row = row0_; row < row1_; row++) {
        const long o = row_offset(a, row) + x0;
        P2<T> bv = ldp_stream(b + o), r;
        if (dtab) dinv = dtab[row % a.ny];          // 2-D stretched meshes: 1/diag of grid row i
        T zx = bv.x * dinv, zy = bv.y * dinv;
        r.x = scale * zx; r.y = scale * zy;
        if (x0 + 1 == a.nx) r.y = (T)0;
        stp_stream(out + o, r);
    }
}

__global__ void __launch_bounds__(256) k_sumsq(RowArgs a, const double *x, double *partials) {
    __shared__ double red[16];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int x0 = 2 * p;
    double acc = 0.0;
    if (p < a.npairs) {
        ROW_RANGE(a.nrows) for (long row = row0_; row < row1_; row++) {
            double2 v = ld2_stream(x + row_offset(a, row) + x0, true);
            if (x0 + 1 == a.nx) v.y = 0.0;
            acc += v.x * v.x + v.y * v.y;
        }
    }
    double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fill_separable(RowArgs a, const double *cx, const double *sy, const double *sz, double *out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.npairs) return;
    const int x0 = 2 * p;
    const double c0 = cx[x0], c1 = (x0 + 1 < a.nx) ? cx[x0 + 1] : 0.0;
    ROW_RANGE(a.nrows) for (long row = row0_; row < row1_; row++) {
        const long k = row / a.ny, i = row - k * a.ny;
        double2 r;
        r.x = c0 * sy[i]; r.y = c1 * sy[i];
        if (sz) { r.x = r.x * sz[k]; r.y = r.y * sz[k]; }
        if (x0 + 1 == a.nx) r.y = 0.0;
        *reinterpret_cast<double2 *>(out + k * a.plane + i * a.pitch + x0) = r;
    }
}

__global__ void __launch_bounds__(256) k_error_sums(RowArgs a, const double *u, const double *sx, const double *sy,
                                                    const double *sz, double *partials, int maxp) {
    __shared__ double red[16];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int x0 = 2 * p;
    double emax = 0.0, e1 = 0.0, e2 = 0.0;
    if (p < a.npairs) {
        const double s0 = sx[x0], s1 = (x0 + 1 < a.nx) ? sx[x0 + 1] : 0.0;
        ROW_RANGE(a.nrows) for (long row = row0_; row < row1_; row++) {
            const long k = row / a.ny, i = row - k * a.ny;
            double2 v = *reinterpret_cast<const double2 *>(u + k * a.plane + i * a.pitch + x0);
            double sol0 = s0 * sy[i], sol1 = s1 * sy[i];
            if (sz) { sol0 = sol0 * sz[k]; sol1 = sol1 * sz[k]; }
            double d0 = fabs(v.x - sol0), d1 = (x0 + 1 < a.nx) ? fabs(v.y - sol1) : 0.0;
            emax = fmax(emax, fmax(d0, d1));
            e1 += d0 + d1;
            e2 += d0 * d0 + d1 * d1;
        }
    }
    const int slot = blockIdx.y * gridDim.x + blockIdx.x;
    double m = block_max(emax, red);
    if (threadIdx.x == 0) partials[slot] = m;
    double s = block_sum(e1, red);
    if (threadIdx.x == 0) partials[maxp + slot] = s;
    s = block_sum(e2, red);
    if (threadIdx.x == 0) partials[2 * maxp + slot] = s;
}

__global__ void __launch_bounds__(256) k_pack(RowArgs a, const double *compact, double *padded, int to_padded) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.nx) return;
    ROW_RANGE(a.nrows) for (long row = row0_; row < row1_; row++) {
        const long o = row_offset(a, row) + j, cidx = row * a.nx + j;
        if (to_padded) padded[o] = compact[cidx];
        else ((double *)compact)[cidx] = padded[o];
    }
}

static RowArgs row_args(const mgk_geom *g) {
    RowArgs a;
    a.nx = g->nx; a.ny = g->ny; a.nz = g->nz; a.npairs = (g->nx + 1) / 2;
    a.pitch = g->pitch; a.plane = g->plane; a.nrows = (long)g->ny * g->nz;
    return a;
}
static void row_grid(const RowArgs &a, int cols, dim3 &grid, dim3 &block, int max_blocks) {
    block = dim3(256);
    unsigned gx = (cols + 255) / 256;
    long gy = a.nrows;
    long cap = max_blocks / (long)gx;
    if (cap < 1) cap = 1;
    if (gy > cap) gy = cap;
    grid = dim3(gx, (unsigned)gy);
}

extern "C" int mgk_jacobi_zero_f64(mgk_ctx *c, const mgk_geom *g, double dinv, double scale,
                                   const double *b, double *unew, void *stream) {
    if (!c || !g || !b || !unew) return fail(MGK_EINVAL, "mgk_jacobi_zero_f64: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.npairs, grid, block, 1024);
    hipLaunchKernelGGL(k_jacobi_zero<double>, grid, block, 0, S(c, stream), a, dinv, scale, b + g->org, unew + g->org, (const double *)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mgk_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *x, double *sumsq_host, void *stream) {
    if (!c || !g || !x || !sumsq_host) return fail(MGK_EINVAL, "mgk_sumsq_f64: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.npairs, grid, block, 1024);
    hipLaunchKernelGGL(k_sumsq, grid, block, 0, S(c, stream), a, x + g->org, c->partials);
    HIPCHK(hipGetLastError());
    return finish_to_host(c, (int)(grid.x * grid.y), 1, S(c, stream), sumsq_host);
}

extern "C" int mgk_fill_separable_f64(mgk_ctx *c, const mgk_geom *g, const double *cx, const double *sy,
                                      const double *sz, double *out, void *stream) {
    if (!c || !g || !cx || !sy || !out || (g->dim == 3 && !sz)) return fail(MGK_EINVAL, "mgk_fill_separable_f64: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.npairs, grid, block, 1024);
    hipLaunchKernelGGL(k_fill_separable, grid, block, 0, S(c, stream), a, cx, sy, g->dim == 3 ? sz : nullptr, out + g->org);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mgk_error_sums_f64(mgk_ctx *c, const mgk_geom *g, const double *u, const double *sx,
                                  const double *sy, const double *sz, double *err3_host, void *stream) {
    if (!c || !g || !u || !sx || !sy || !err3_host || (g->dim == 3 && !sz)) return fail(MGK_EINVAL, "mgk_error_sums_f64: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.npairs, grid, block, 1024);
    hipStream_t s = S(c, stream);
    hipLaunchKernelGGL(k_error_sums, grid, block, 0, s, a, u + g->org, sx, sy, g->dim == 3 ? sz : nullptr, c->partials, c->max_partials);
    HIPCHK(hipGetLastError());
    const int np = (int)(grid.x * grid.y);
    hipLaunchKernelGGL(k_finish_max, dim3(1), dim3(256), 0, s, c->partials, np, c->result_dev, 0);
    hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(256), 0, s, c->partials + c->max_partials, np, c->result_dev, 1);
    hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(256), 0, s, c->partials + 2L * c->max_partials, np, c->result_dev, 2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->result_host, c->result_dev, sizeof(double) * 3, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int q = 0; q < 3; q++) err3_host[q] = c->result_host[q];
    return 0;
}

extern "C" int mgk_pack_f64(mgk_ctx *c, const mgk_geom *g, const double *compact_dev, double *padded_dev, void *stream) {
    if (!c || !g || !compact_dev || !padded_dev) return fail(MGK_EINVAL, "mgk_pack_f64: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.nx, grid, block, 8192);
    hipLaunchKernelGGL(k_pack, grid, block, 0, S(c, stream), a, compact_dev, padded_dev + g->org, 1);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_unpack_f64(mgk_ctx *c, const mgk_geom *g, const double *padded_dev, double *compact_dev, void *stream) {
    if (!c || !g || !compact_dev || !padded_dev) return fail(MGK_EINVAL, "mgk_unpack_f64: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.nx, grid, block, 8192);
    hipLaunchKernelGGL(k_pack, grid, block, 0, S(c, stream), a, compact_dev, (double *)padded_dev + g->org, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// grid transfer
// ------------------------------------------------------------------------------------------
struct XferArgs {
    int nxf, nyf, nzf, nxc, nyc, nzc;
    long pf, plf, pc, plc;       // pitches / planes of fine and coarse
};

// full weighting: one lane per coarse point; fine values summed in ascending fine index
// (dk, di, dj) exactly like the 9/27-entry row of res (src/solver.c:1081-1090).
template <typename T, int DIM>
__global__ void __launch_bounds__(256) k_restrict(XferArgs a, const T *rf, T *bc) {
    const int jc = blockIdx.x * blockDim.x + threadIdx.x;
    if (jc > a.nxc) return;             // jc == nxc: the right ghost, written as 0
    const long nrows = (long)a.nyc * a.nzc;
    // weights: src/matbuild.c:422-431 (0.125-0.0625|1-i| ...); 3-D extension multiplies by {1/4,1/2,1/4}
    const T w2[3][3] = {{(T)0.0625, (T)0.125, (T)0.0625}, {(T)0.125, (T)0.25, (T)0.125}, {(T)0.0625, (T)0.125, (T)0.0625}};
    const T w1[3] = {(T)0.25, (T)0.5, (T)0.25};
    ROW_RANGE(nrows)
    int kc = (int)(row0_ / a.nyc), ic = (int)(row0_ - (long)kc * a.nyc);
    for (long row = row0_; row < row1_; row++) {
        T sum = (T)0;
        if (jc < a.nxc) {
            const T *base = rf + (DIM == 3 ? (long)(2 * kc) * a.plf : 0) + (long)(2 * ic) * a.pf + 2 * jc;
            // per fine row: the aligned pair (2jc, 2jc+1) as one vector load, then 2jc+2 (same lines as the next lane's pair)
            if (DIM == 3) {
#pragma unroll
                for (int dk = 0; dk < 3; dk++)
#pragma unroll
                    for (int di = 0; di < 3; di++) {
                        const T *rowp = base + dk * a.plf + di * a.pf;
                        const P2<T> pr = *reinterpret_cast<const P2<T> *>(rowp);
                        const T third = rowp[2];
                        sum += (w1[dk] * w2[di][0]) * pr.x;
                        sum += (w1[dk] * w2[di][1]) * pr.y;
                        sum += (w1[dk] * w2[di][2]) * third;
                    }
            } else {
#pragma unroll
                for (int di = 0; di < 3; di++) {
                    const T *rowp = base + di * a.pf;
                    const P2<T> pr = *reinterpret_cast<const P2<T> *>(rowp);
                    const T third = rowp[2];
                    sum += w2[di][0] * pr.x;
                    sum += w2[di][1] * pr.y;
                    sum += w2[di][2] * third;
                }
            }
        }
        bc[(DIM == 3 ? (long)kc * a.plc : 0) + (long)ic * a.pc + jc] = sum;
        if (++ic == a.nyc) { ic = 0; kc++; }
    }
}

// bilinear / trilinear interpolation + correction: one lane per aligned fine pair.
// Row of pro summed in ascending coarse index (kc, ic, jc) (src/solver.c:1140-1148), then u + rv.
// Out-of-grid parents are read from the coarse ghosts (0 at a global boundary, halo data at a slab
// boundary), which adds +0 where the assembled row has no entry.
template <typename T, int DIM>
__global__ void __launch_bounds__(256) k_prolong_add(XferArgs a, const T *uc, T *uf) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int x0 = 2 * p;                 // even fine x; x0+1 odd
    if (x0 >= a.nxf) return;
    const long nrows = (long)a.nyf * a.nzf;
    const int jc1 = x0 / 2, jc0 = jc1 - 1;      // parents of x0 (weights 1/2, 1/2); parent of x0+1 is jc1 (weight 1)
    const bool last = (x0 + 1 >= a.nxf);
    ROW_RANGE(nrows)
    int k = (int)(row0_ / a.nyf), i = (int)(row0_ - (long)k * a.nyf);
    for (long row = row0_; row < row1_; row++) {
        const int iodd = i & 1, kodd = k & 1;
        const int ic0 = iodd ? (i - 1) / 2 : i / 2 - 1, nic = iodd ? 1 : 2;
        const int kc0 = (DIM == 3) ? (kodd ? (k - 1) / 2 : k / 2 - 1) : 0, nkc = (DIM == 3) ? (kodd ? 1 : 2) : 1;
        const T wi = iodd ? (T)1 : (T)0.5, wk = (DIM == 3) ? (kodd ? (T)1 : (T)0.5) : (T)1;
        const T wh = (DIM == 3) ? wk * (wi * (T)0.5) : wi * (T)0.5;
        const T w1_ = (DIM == 3) ? wk * (wi * (T)1) : wi * (T)1;
        T *fp = uf + (DIM == 3 ? (long)k * a.plf : 0) + (long)i * a.pf + x0;
        P2<T> v = ldp_stream(fp);
        const T *cr = uc + (DIM == 3 ? (long)kc0 * a.plc : 0) + (long)ic0 * a.pc;
        T s0 = (T)0, s1 = (T)0;
        for (int qk = 0; qk < nkc; qk++)
            for (int qi = 0; qi < nic; qi++) {
                const T *c = cr + (long)qk * a.plc + (long)qi * a.pc;
                const T c0 = c[jc0], c1 = c[jc1];
                s0 += wh * c0;
                s0 += wh * c1;
                s1 += w1_ * c1;
            }
        v.x = v.x + s0;
        v.y = last ? (T)0 : v.y + s1;
        stp_stream(fp, v);
        if (++i == a.nyf) { i = 0; k++; }
    }
}

static int xfer_args(const mgk_geom *gf, const mgk_geom *gc, XferArgs &a) {
    if (gf->dim != gc->dim) return fail(MGK_EINVAL, "grid transfer: dimension mismatch");
    if (gf->nx != 2 * gc->nx + 1 || gf->ny != 2 * gc->ny + 1) return fail(MGK_EINVAL, "grid transfer: need nf = 2*nc+1 in x and y");
    if (gf->dim == 3 && !(gf->nz == 2 * gc->nz + 1 || gf->nz == 2 * gc->nz))
        return fail(MGK_EINVAL, "grid transfer: need nzf = 2*nzc (+1 on the last slab)");
    a.nxf = gf->nx; a.nyf = gf->ny; a.nzf = gf->nz; a.nxc = gc->nx; a.nyc = gc->ny; a.nzc = gc->nz;
    a.pf = gf->pitch; a.plf = gf->plane; a.pc = gc->pitch; a.plc = gc->plane;
    return 0;
}

extern "C" int mgk_restrict_fw_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc,
                                   const double *rf, double *bc, void *stream) {
    if (!c || !gf || !gc || !rf || !bc) return fail(MGK_EINVAL, "mgk_restrict_fw_f64: bad arguments");
    XferArgs a;
    int rc = xfer_args(gf, gc, a);
    if (rc) return rc;
    dim3 block(256), grid((gc->nx + 1 + 255) / 256, 1);
    long rows = (long)gc->ny * gc->nz, cap = 1024 / grid.x; if (cap < 1) cap = 1;
    grid.y = (unsigned)(rows < cap ? rows : cap);
    if (gf->dim == 3) hipLaunchKernelGGL((k_restrict<double, 3>), grid, block, 0, S(c, stream), a, rf + gf->org, bc + gc->org);
    else hipLaunchKernelGGL((k_restrict<double, 2>), grid, block, 0, S(c, stream), a, rf + gf->org, bc + gc->org);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mgk_prolong_add_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc,
                                   const double *uc, double *uf, void *stream) {
    if (!c || !gf || !gc || !uc || !uf) return fail(MGK_EINVAL, "mgk_prolong_add_f64: bad arguments");
    XferArgs a;
    int rc = xfer_args(gf, gc, a);
    if (rc) return rc;
    const int npairs = (gf->nx + 1) / 2;
    dim3 block(256), grid((npairs + 255) / 256, 1);
    long rows = (long)gf->ny * gf->nz, cap = 1024 / grid.x; if (cap < 1) cap = 1;
    grid.y = (unsigned)(rows < cap ? rows : cap);
    if (gf->dim == 3) hipLaunchKernelGGL((k_prolong_add<double, 3>), grid, block, 0, S(c, stream), a, uc + gc->org, uf + gf->org);
    else hipLaunchKernelGGL((k_prolong_add<double, 2>), grid, block, 0, S(c, stream), a, uc + gc->org, uf + gf->org);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// The I-cycle's coupled level operator for G grids in one level (src/solver.c:489-510 levelMatrixA):
//     M(g,g) = A_g,   M(g0,g1) = R^(g0-g1) A_g1  (g1 < g0, fillRestrictionPortion :255-345),
//     M(g1,g0) = W    (g1 < g0, fillProlongationPortion :347-470: A_g1 P^(g0-g1) cut to P's window of 2S-1 points, S = 2^(g0-g1)).
// With s_0 = A_0 x_0, s_g = R s_(g-1) + A_g x_g (mgk_apply_f64, mgk_restrict_fw_f64, k_apply_add) the lower triangle and the
// diagonal are one cascade; k_window_add then adds the upper blocks: yf(i,j) += sum over the <= 4 coarse points (ic,jc) whose
// window holds (i,j) of w[i - S ic][j - S jc] * xc(ic,jc), ascending coarse index (the column order of the assembled row).
// One lane per point; xc's ghost ring is zero, so a parent outside the grid adds an exact zero.
// ------------------------------------------------------------------------------------------
struct ApplyAddArgs { int n, pitch; double c[5]; };
__global__ void __launch_bounds__(256) k_apply_add_2d(ApplyAddArgs a, const double *x, double *y) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n) return;
    for (int i = blockIdx.y; i < a.n; i += gridDim.y) {
        const long o = (long)i * a.pitch + j;
        double z = y[o];
        z = z + a.c[0] * x[o - a.pitch];
        z = z + a.c[1] * x[o - 1];
        z = z + a.c[2] * x[o];
        z = z + a.c[3] * x[o + 1];
        z = z + a.c[4] * x[o + a.pitch];
        y[o] = z;
    }
}
extern "C" int mgk_apply_add_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, const double *x, double *y, void *stream) {
    if (!c || !g || !coef || !x || !y || x == y || g->dim != 2 || g->nx != g->ny) return fail(MGK_EINVAL, "mgk_apply_add_f64: bad arguments (square 2-D grids)");
    ApplyAddArgs a;
    a.n = g->nx; a.pitch = g->pitch;
    for (int k = 0; k < 5; k++) a.c[k] = coef[k];
    dim3 block(256), grid((a.n + 255) / 256, (unsigned)(a.n < 2048 ? a.n : 2048));
    hipLaunchKernelGGL(k_apply_add_2d, grid, block, 0, S(c, stream), a, x + g->org, y + g->org);
    HIPCHK(hipGetLastError());
    return 0;
}
struct WindowArgs { int nf, nc, pf, pc, S, W; };
__global__ void __launch_bounds__(256) k_window_add(WindowArgs a, const double *wtab, const double *xc, double *yf) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.nf) return;
    // parents along an axis: (p+1) % S == 0 -> the one coarse point p / S; otherwise p / S - 1 and p / S
    const int jm = j / a.S, nj = ((j + 1) % a.S == 0) ? 1 : 2, jc0 = jm - (nj - 1);
    for (int i = blockIdx.y; i < a.nf; i += gridDim.y) {
        const int im = i / a.S, ni = ((i + 1) % a.S == 0) ? 1 : 2, ic0 = im - (ni - 1);
        double y = yf[(long)i * a.pf + j];
        for (int p = 0; p < ni; p++) {
            const int ic = ic0 + p, di = i - a.S * ic;
            for (int q = 0; q < nj; q++) {
                const int jc = jc0 + q, dj = j - a.S * jc;
                y = y + wtab[di * a.W + dj] * xc[(long)ic * a.pc + jc];
            }
        }
        yf[(long)i * a.pf + j] = y;
    }
}
extern "C" int mgk_window_add_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, int stride, const double *wtab_dev,
                                  const double *xc, double *yf, void *stream) {
    if (!c || !gf || !gc || !wtab_dev || !xc || !yf) return fail(MGK_EINVAL, "mgk_window_add_f64: bad arguments");
    if (gf->dim != 2 || gc->dim != 2 || gf->nx != gf->ny || gc->nx != gc->ny || stride < 2 || (stride & (stride - 1)) ||
        (long)gf->nx + 1 != (long)stride * (gc->nx + 1))
        return fail(MGK_EINVAL, "mgk_window_add_f64: needs square 2-D grids with nf + 1 = stride (nc + 1), stride a power of two");
    WindowArgs a;
    a.nf = gf->nx; a.nc = gc->nx; a.pf = gf->pitch; a.pc = gc->pitch; a.S = stride; a.W = 2 * stride - 1;
    dim3 block(256), grid((a.nf + 255) / 256, (unsigned)(a.nf < 2048 ? a.nf : 2048));
    hipLaunchKernelGGL(k_window_add, grid, block, 0, S(c, stream), a, wtab_dev, xc + gc->org, yf + gf->org);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// y = B x for a dense row-major m x n matrix: the exact coarse-grid solve of PCMG (PETSc's default coarse solver is PCLU,
// src/solver.c:1931-1932 takes it as it comes): B = A^-1 of the coarsest grid, inverted once on the host.  One wavefront per row,
// lanes stride over the columns (coalesced), 64-lane shuffle tree.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dense_mult(int m, int n, const double *B, const double *x, double *y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= m) return;
    const double *br = B + (long)row * n;
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) acc += br[j] * x[j];
    acc = wave_sum(acc);
    if (lane == 0) y[row] = acc;
}
extern "C" int mgk_dense_mult_f64(mgk_ctx *c, int m, int n, const double *B_dev, const double *x, double *y, void *stream) {
    if (!c || m < 1 || n < 1 || !B_dev || !x || !y || x == y) return fail(MGK_EINVAL, "mgk_dense_mult_f64: bad arguments");
    hipLaunchKernelGGL(k_dense_mult, dim3((m + 3) / 4), dim3(256), 0, S(c, stream), m, n, B_dev, x, y);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// flat vector kernels (PETSc Vec BLAS-1 surface of the shim) and a generic CSR SpMV.
// They run over a whole allocation (padded fields included: ghosts are zero and stay zero under
// every linear combination), one element per lane-iteration, grid-stride.
// ------------------------------------------------------------------------------------------
enum { F_AXPY = 0, F_AYPX = 1, F_AXPBYPCZ = 2, F_FILL = 3, F_SCALE = 4, F_PWMULT = 5, F_COPY = 6 };
template <int OP>
__global__ void __launch_bounds__(256) k_flat(long n, double a, double b, double c, const double *x, const double *y, double *z) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) {
        if (OP == F_AXPY) z[q] = z[q] + a * x[q];                     // VecAXPY(z, a, x)
        else if (OP == F_AYPX) z[q] = x[q] + a * z[q];                // VecAYPX(z, a, x)
        else if (OP == F_AXPBYPCZ) z[q] = (a * x[q] + b * y[q]) + c * z[q];   // VecAXPBYPCZ(z,a,b,c,x,y)
        else if (OP == F_FILL) z[q] = a;                              // VecSet
        else if (OP == F_SCALE) z[q] = a * z[q];                      // VecScale
        else if (OP == F_PWMULT) z[q] = x[q] * y[q];                  // VecPointwiseMult
        else z[q] = x[q];                                             // VecCopy
    }
}
__global__ void __launch_bounds__(256) k_flat_dot(long n, const double *x, const double *y, double *partials) {
    __shared__ double red[16];
    const long stride = (long)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) acc += x[q] * y[q];
    double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// MatMult on assembled AIJ, ascending columns, separate multiply and add, one running sum per row (the order of the oracle's CSR
// leg: results are bit-identical to it).  A wavefront owns 64 consecutive rows: it stages the contiguous (col, val) range of its
// rows in LDS with coalesced loads -- 64 entries per step instead of 64 lanes walking 64 different rows -- and each lane then
// sums its own row out of LDS in order.  Row groups with more than CSR_CAP entries (long rows: not the reference's 5/9/14-entry
// operators) read global memory directly.
// Column indices are element offsets into x (already translated when x is a padded grid field);
// rows map to y / addto through (row_n, row_pitch, row_org) when y is a padded 2-D grid field (row_n > 0).
#define CSR_CAP 1024
__global__ void __launch_bounds__(256) k_csr_mult(long nrows, const long *rowptr, const int *col, const double *val,
                                                  const double *x, double *y, double alpha, const double *addto,
                                                  int row_n, long row_pitch, long row_org) {
    __shared__ double sval[4][CSR_CAP];
    __shared__ int scol[4][CSR_CAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long nwaves = (long)gridDim.x * 4;
    for (long r0 = ((long)blockIdx.x * 4 + w) * 64; r0 < nrows; r0 += nwaves * 64) {
        const long r = r0 + lane;
        const long rlast = (r0 + 64 < nrows) ? r0 + 64 : nrows;
        const long q0 = rowptr[r0], q1 = rowptr[rlast];          // wave-uniform entry range of the 64 rows
        const bool staged = (q1 - q0 <= CSR_CAP);
        if (staged) {
            for (long q = q0 + lane; q < q1; q += 64) { sval[w][q - q0] = val[q]; scol[w][q - q0] = col[q]; }
        }
        // (the four waves of a block work on different rows and never exchange data: no workgroup barrier; a wave sees its own LDS writes)
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the staging stores have landed
        if (r < nrows) {
            double sum = 0.0;
            const long a0 = rowptr[r], a1 = rowptr[r + 1];
            if (staged) for (long q = a0; q < a1; q++) sum += sval[w][q - q0] * x[scol[w][q - q0]];
            else for (long q = a0; q < a1; q++) sum += val[q] * x[col[q]];
            const long o = row_n > 0 ? row_org + (r / row_n) * row_pitch + (r % row_n) : r;
            y[o] = addto ? addto[o] + alpha * sum : sum;
        }
    }
}
// Measured-ceiling probe: STREAM triad a = b + s*c with the access mix and hints of a Jacobi sweep
// (two streamed reads, one streamed write, 24 B per element; 16-B lane vectors; contiguous chunk per block).
template <int NT>
__global__ void __launch_bounds__(1024) k_triad(long nvec, double *a, const double *b, const double *c, double s) {
    constexpr int U = 4;                          // independent 16-B loads in flight per lane and stream
    // grid-stride: at any moment the resident blocks cover one contiguous window that moves through the arrays
    // (like the sweep's plane march), so the accesses spread over every HBM channel
    const long q1 = nvec;
    const d2v *bv = reinterpret_cast<const d2v *>(b), *cv = reinterpret_cast<const d2v *>(c);
    d2v *av = reinterpret_cast<d2v *>(a);
    for (long base = (long)blockIdx.x * U * blockDim.x + threadIdx.x; base < q1; base += (long)gridDim.x * U * blockDim.x) {
        d2v x[U], y[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const long q = base + (long)k * blockDim.x;
            if (q < q1) {
                if (NT) { x[k] = __builtin_nontemporal_load(bv + q); y[k] = __builtin_nontemporal_load(cv + q); }
                else { x[k] = bv[q]; y[k] = cv[q]; }
            }
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            const long q = base + (long)k * blockDim.x;
            if (q < q1) {
                d2v r; r.x = x[k].x + s * y[k].x; r.y = x[k].y + s * y[k].y;
                if (NT) __builtin_nontemporal_store(r, av + q); else av[q] = r;
            }
        }
    }
}
extern "C" int mgk_stream_triad_f64(mgk_ctx *c, long n, double *a, const double *b, const double *cc, double s,
                                    int blocks, int nontemporal, void *stream) {
    if (!c || !a || !b || !cc || n < 2 || (n & 1) || blocks < 1) return fail(MGK_EINVAL, "mgk_stream_triad_f64: bad arguments (n even)");
    if (nontemporal) hipLaunchKernelGGL(k_triad<1>, dim3((unsigned)blocks), dim3(1024), 0, S(c, stream), n / 2, a, b, cc, s);
    else hipLaunchKernelGGL(k_triad<0>, dim3((unsigned)blocks), dim3(1024), 0, S(c, stream), n / 2, a, b, cc, s);
    HIPCHK(hipGetLastError());
    return 0;
}
static unsigned flat_grid(long n) {
    long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}
#define FLAT_LAUNCH(OP, n, a, b, c_, x, y, z)                                                   \
    do { hipLaunchKernelGGL(k_flat<OP>, dim3(flat_grid(n)), dim3(256), 0, S(c, stream), n, a, b, c_, x, y, z); \
         HIPCHK(hipGetLastError()); return 0; } while (0)
extern "C" int mgk_flat_axpy(mgk_ctx *c, long n, double a, const double *x, double *y, void *stream) { FLAT_LAUNCH(F_AXPY, n, a, 0.0, 0.0, x, nullptr, y); }
extern "C" int mgk_flat_aypx(mgk_ctx *c, long n, double a, const double *x, double *y, void *stream) { FLAT_LAUNCH(F_AYPX, n, a, 0.0, 0.0, x, nullptr, y); }
extern "C" int mgk_flat_axpbypcz(mgk_ctx *c, long n, double a, double b, double g, const double *x, const double *y, double *z, void *stream) { FLAT_LAUNCH(F_AXPBYPCZ, n, a, b, g, x, y, z); }
extern "C" int mgk_flat_fill(mgk_ctx *c, long n, double a, double *z, void *stream) { FLAT_LAUNCH(F_FILL, n, a, 0.0, 0.0, nullptr, nullptr, z); }
extern "C" int mgk_flat_scale(mgk_ctx *c, long n, double a, double *z, void *stream) { FLAT_LAUNCH(F_SCALE, n, a, 0.0, 0.0, nullptr, nullptr, z); }
extern "C" int mgk_flat_pointwise_mult(mgk_ctx *c, long n, const double *x, const double *y, double *z, void *stream) { FLAT_LAUNCH(F_PWMULT, n, 0.0, 0.0, 0.0, x, y, z); }
extern "C" int mgk_flat_dot(mgk_ctx *c, long n, const double *x, const double *y, double *dot_host, void *stream) {
    if (!c || !x || !y || !dot_host) return fail(MGK_EINVAL, "mgk_flat_dot: bad arguments");
    unsigned g = flat_grid(n);
    if ((int)g > c->max_partials) g = c->max_partials;
    hipLaunchKernelGGL(k_flat_dot, dim3(g), dim3(256), 0, S(c, stream), n, x, y, c->partials);
    HIPCHK(hipGetLastError());
    return finish_to_host(c, (int)g, 1, S(c, stream), dot_host);
}
extern "C" int mgk_csr_mult_f64(mgk_ctx *c, long nrows, const long *rowptr, const int *col, const double *val,
                                const double *x, double *y, double alpha, const double *addto,
                                int row_n, long row_pitch, long row_org, void *stream) {
    if (!c || !rowptr || !col || !val || !x || !y) return fail(MGK_EINVAL, "mgk_csr_mult_f64: bad arguments");
    hipLaunchKernelGGL(k_csr_mult, dim3(flat_grid(nrows)), dim3(256), 0, S(c, stream), nrows, rowptr, col, val, x, y, alpha, addto,
                       row_n, row_pitch, row_org);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// fp32 fields and the fp64<->fp32 bridges of the mixed-precision cycle (BASELINE config 5:
// fp32 smoother sweeps, fp64 residual and correction).  Same kernels, T = float: one lane owns
// four unknowns (16 bytes), a wave row is 256 unknowns; same canonical arithmetic, in fp32.
// ------------------------------------------------------------------------------------------
extern "C" int mgk_geom_init_f32(mgk_geom *g, int dim, int nx, int ny, int nz) {
    if (!g || dim != 3 || nx < 1 || ny < 1 || nz < 1 || (nx & 1) == 0)
        return fail(MGK_EINVAL, "mgk_geom_init_f32: need dim 3 and odd nx");
    g->dim = dim; g->nx = nx; g->ny = ny; g->nz = nz;
    g->pitch = ((32 + nx + 1 + 31) / 32) * 32;            // rows start on 128-byte lines, interior x=0 at column 32
    g->plane = (long)g->pitch * (ny + 2);
    g->org = g->plane + g->pitch + 32;
    g->total = g->plane * (nz + 2) + g->pitch;
    return 0;
}

extern "C" int mgk_jacobi_range_f32(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                    const float *b, const float *u, float *unew, int zbeg, int zend, void *stream) {
    if (!c || !g || !coef || !b || !u || !unew || u == unew) return fail(MGK_EINVAL, "mgk_jacobi_f32: bad arguments");
    if (zbeg < 0 || zend > g->nz || zbeg >= zend) return fail(MGK_EINVAL, "mgk_jacobi_range_f32: empty or out-of-range plane range");
    StArgs<float> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org;
    set_coef(a, g, coef); a.dinv = (float)dinv; a.scale = (float)scale;
    a.zbeg = zbeg; a.zend = zend;
    if (jrow_ok<float>(g)) return launch_jrow<float, false>(c, g, a, zbeg, zend, S(c, stream), nullptr, 0, nullptr);
    return dispatch_st<MODE_JACOBI>(c, g, a, S(c, stream), nullptr);
}
extern "C" int mgk_jacobi_f32(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                              const float *b, const float *u, float *unew, void *stream) {
    if (!g) return fail(MGK_EINVAL, "mgk_jacobi_f32: bad arguments");
    return mgk_jacobi_range_f32(c, g, coef, dinv, scale, b, u, unew, 0, g->nz, stream);
}
extern "C" int mgk_residual_range_f32(mgk_ctx *c, const mgk_geom *g, const double *coef,
                                      const float *b, const float *u, float *r, int zbeg, int zend, void *stream) {
    if (!c || !g || !coef || !b || !u || !r || u == r) return fail(MGK_EINVAL, "mgk_residual_f32: bad arguments");
    if (zbeg < 0 || zend > g->nz || zbeg >= zend) return fail(MGK_EINVAL, "mgk_residual_range_f32: empty or out-of-range plane range");
    StArgs<float> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = r + g->org;
    set_coef(a, g, coef);
    a.zbeg = zbeg; a.zend = zend;
    return dispatch_st<MODE_RESIDUAL>(c, g, a, S(c, stream), nullptr);
}
extern "C" int mgk_residual_f32(mgk_ctx *c, const mgk_geom *g, const double *coef,
                                const float *b, const float *u, float *r, void *stream) {
    if (!g) return fail(MGK_EINVAL, "mgk_residual_f32: bad arguments");
    return mgk_residual_range_f32(c, g, coef, b, u, r, 0, g->nz, stream);
}
extern "C" int mgk_jacobi_zero_f32(mgk_ctx *c, const mgk_geom *g, double dinv, double scale,
                                   const float *b, float *unew, void *stream) {
    if (!c || !g || !b || !unew) return fail(MGK_EINVAL, "mgk_jacobi_zero_f32: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.npairs, grid, block, 1024);
    hipLaunchKernelGGL(k_jacobi_zero<float>, grid, block, 0, S(c, stream), a, (float)dinv, (float)scale, b + g->org, unew + g->org, (const float *)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_restrict_fw_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc,
                                   const float *rf, float *bc, void *stream) {
    if (!c || !gf || !gc || !rf || !bc || gf->dim != 3) return fail(MGK_EINVAL, "mgk_restrict_fw_f32: bad arguments");
    XferArgs a;
    int rc = xfer_args(gf, gc, a);
    if (rc) return rc;
    dim3 block(256), grid((gc->nx + 1 + 255) / 256, 1);
    long rows = (long)gc->ny * gc->nz, cap = 1024 / grid.x; if (cap < 1) cap = 1;
    grid.y = (unsigned)(rows < cap ? rows : cap);
    hipLaunchKernelGGL((k_restrict<float, 3>), grid, block, 0, S(c, stream), a, rf + gf->org, bc + gc->org);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_prolong_add_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc,
                                   const float *uc, float *uf, void *stream) {
    if (!c || !gf || !gc || !uc || !uf || gf->dim != 3) return fail(MGK_EINVAL, "mgk_prolong_add_f32: bad arguments");
    XferArgs a;
    int rc = xfer_args(gf, gc, a);
    if (rc) return rc;
    const int npairs = (gf->nx + 1) / 2;
    dim3 block(256), grid((npairs + 255) / 256, 1);
    long rows = (long)gf->ny * gf->nz, cap = 1024 / grid.x; if (cap < 1) cap = 1;
    grid.y = (unsigned)(rows < cap ? rows : cap);
    hipLaunchKernelGGL((k_prolong_add<float, 3>), grid, block, 0, S(c, stream), a, uc + gc->org, uf + gf->org);
    HIPCHK(hipGetLastError());
    return 0;
}

// fp64 residual r = b - A u, written as fp32 (the right-hand side of the fp32 correction cycle) and reduced
// to sum r^2 in fp64 in the same pass: reads 16 B, writes 4 B per unknown
extern "C" int mgk_residual_f64_to_f32(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const double *coef,
                                       const double *b, const double *u, float *r32, double *sumsq_host, void *stream) {
    if (!c || !g || !g32 || !coef || !b || !u || !r32 || !sumsq_host || g->dim != 3 ||
        g->nx != g32->nx || g->ny != g32->ny || g->nz != g32->nz)
        return fail(MGK_EINVAL, "mgk_residual_f64_to_f32: bad arguments");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out32 = r32 + g32->org; a.ors = g32->pitch; a.oms = g32->plane;
    a.partials = c->partials;
    set_coef(a, g, coef);
    int nblk = 0;
    int rc = dispatch_st<MODE_RES32>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}

// the same, also writing e0 = scale32 * (r32 * dinv32): the first sweep of the fp32 correction cycle from its zero guess (what
// mgk_jacobi_zero_f32 would compute from r32: one launch and one read of r32 less)
extern "C" int mgk_residual_f64_to_f32_jz(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const double *coef,
                                          const double *b, const double *u, float *r32, float *e0, double dinv, double scale,
                                          double *sumsq_host, void *stream) {
    if (!c || !g || !g32 || !coef || !b || !u || !r32 || !e0 || e0 == r32 || !sumsq_host || g->dim != 3 ||
        g->nx != g32->nx || g->ny != g32->ny || g->nz != g32->nz)
        return fail(MGK_EINVAL, "mgk_residual_f64_to_f32_jz: bad arguments");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out32 = r32 + g32->org; a.ors = g32->pitch; a.oms = g32->plane;
    a.jz32 = e0 + g32->org; a.dinv32 = (float)dinv; a.scale32 = (float)scale;
    a.partials = c->partials;
    set_coef(a, g, coef);
    int nblk = 0;
    int rc = dispatch_st<MODE_RES32>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}

// The whole outer step of the mixed-precision iteration in one pass: unew = u + (double) e32 (written), r32 = (float)(b - A unew),
// sum r^2.  Every u value the block touches is corrected on the fly, like the fused prolongation sweep does with the
// coarse interpolant: 8 (u) + 4 (e32) + 8 (b) read, 8 (unew) + 4 (r32) written = 32 B instead of 20 + 20.
extern "C" int mgk_correct_residual_f64_f32(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const double *coef,
                                            const double *b, const double *u, const float *e32, double *unew, float *r32,
                                            double *sumsq_host, void *stream) {
    if (!c || !g || !g32 || !coef || !b || !u || !e32 || !unew || !r32 || !sumsq_host || u == unew || (const float *)r32 == e32 ||
        g->dim != 3 || g->nx != g32->nx || g->ny != g32->ny || g->nz != g32->nz)
        return fail(MGK_EINVAL, "mgk_correct_residual_f64_f32: bad arguments");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org;
    a.out32 = r32 + g32->org; a.e32 = e32 + g32->org; a.ors = g32->pitch; a.oms = g32->plane;
    a.partials = c->partials;
    set_coef(a, g, coef);
    int nblk = 0;
    int rc = dispatch_st<MODE_CRES32>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}

extern "C" int mgk_correct_residual_f64_f32_jz(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const double *coef,
                                               const double *b, const double *u, const float *e32, double *unew, float *r32,
                                               float *e0, double dinv, double scale, double *sumsq_host, void *stream) {
    if (!c || !g || !g32 || !coef || !b || !u || !e32 || !unew || !r32 || !e0 || e0 == r32 || (const float *)e0 == e32 || !sumsq_host ||
        u == unew || (const float *)r32 == e32 || g->dim != 3 || g->nx != g32->nx || g->ny != g32->ny || g->nz != g32->nz)
        return fail(MGK_EINVAL, "mgk_correct_residual_f64_f32_jz: bad arguments");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org;
    a.out32 = r32 + g32->org; a.e32 = e32 + g32->org; a.ors = g32->pitch; a.oms = g32->plane;
    a.jz32 = e0 + g32->org; a.dinv32 = (float)dinv; a.scale32 = (float)scale;
    a.partials = c->partials;
    set_coef(a, g, coef);
    int nblk = 0;
    int rc = dispatch_st<MODE_CRES32>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}

// u64 += (double) e32  (the fp64 correction step)
__global__ void __launch_bounds__(256) k_correct(RowArgs a, long pitch32, long plane32, const float *e, double *u) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.npairs) return;
    const int x0 = 2 * p;
    ROW_RANGE(a.nrows)
    int k = (int)(row0_ / a.ny), i = (int)(row0_ - (long)k * a.ny);
    for (long row = row0_; row < row1_; row++) {
        double *up = u + (long)k * a.plane + (long)i * a.pitch + x0;
        const P2<float> ev = *reinterpret_cast<const P2<float> *>(e + (long)k * plane32 + (long)i * pitch32 + x0);
        P2<double> v = ldp_stream(up);
        v.x = v.x + (double)ev.x;
        v.y = (x0 + 1 == a.nx) ? 0.0 : v.y + (double)ev.y;
        stp_stream(up, v);
        if (++i == a.ny) { i = 0; k++; }
    }
}
extern "C" int mgk_correct_f64_from_f32(mgk_ctx *c, const mgk_geom *g, const mgk_geom *g32, const float *e32, double *u, void *stream) {
    if (!c || !g || !g32 || !e32 || !u || g->nx != g32->nx || g->ny != g32->ny || g->nz != g32->nz)
        return fail(MGK_EINVAL, "mgk_correct_f64_from_f32: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.npairs, grid, block, 1024);
    hipLaunchKernelGGL(k_correct, grid, block, 0, S(c, stream), a, (long)g32->pitch, g32->plane, e32 + g32->org, u + g->org);
    HIPCHK(hipGetLastError());
    return 0;
}

// compact fp64 host/device array <-> padded fp32 field (tests)
__global__ void __launch_bounds__(256) k_pack32(RowArgs a, const double *compact, float *padded, int to_padded) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.nx) return;
    ROW_RANGE(a.nrows) for (long row = row0_; row < row1_; row++) {
        const long o = row_offset(a, row) + j, cidx = row * a.nx + j;
        if (to_padded) padded[o] = (float)compact[cidx];
        else ((double *)compact)[cidx] = (double)padded[o];
    }
}
extern "C" int mgk_pack_f32(mgk_ctx *c, const mgk_geom *g32, const double *compact_dev, float *padded_dev, void *stream) {
    if (!c || !g32 || !compact_dev || !padded_dev) return fail(MGK_EINVAL, "mgk_pack_f32: bad arguments");
    RowArgs a = row_args(g32);
    dim3 grid, block;
    row_grid(a, a.nx, grid, block, 1024);
    hipLaunchKernelGGL(k_pack32, grid, block, 0, S(c, stream), a, compact_dev, padded_dev + g32->org, 1);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_unpack_f32(mgk_ctx *c, const mgk_geom *g32, const float *padded_dev, double *compact_dev, void *stream) {
    if (!c || !g32 || !compact_dev || !padded_dev) return fail(MGK_EINVAL, "mgk_unpack_f32: bad arguments");
    RowArgs a = row_args(g32);
    dim3 grid, block;
    row_grid(a, a.nx, grid, block, 1024);
    hipLaunchKernelGGL(k_pack32, grid, block, 0, S(c, stream), a, compact_dev, (float *)padded_dev + g32->org, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// K2+K3 fused: r = b - A u restricted on the fly, r is never written (src/solver.c:1534-1535).
// One block owns a full-row tile of RR = 5 fine rows (tile stride 4: the 5th row is recomputed by the next
// tile, L2 hits) = 2 coarse rows, marches z; every lane owns one pair column for all 5 rows, so y neighbours
// are registers, x neighbours come from the LDS tile of the current plane.  The residual plane goes to a
// second LDS tile from which lane t accumulates coarse column t of both coarse rows, fine plane by fine plane,
// in exactly the order of the assembled row of res: (dk, di, dj) ascending.  An even fine plane is dk = 2 of
// coarse plane z/2-1 and dk = 0 of z/2 (equal weights: the products are computed once).
// Traffic: 16 B per fine unknown read (+ the recomputed row from L2) and 1 B written.  3-D, fp64.
// ------------------------------------------------------------------------------------------
template <typename T>
struct RRArgs {
    const T *u, *b;
    T *bc;
    int nx, ny, nz;          // fine
    int nxc, nyc, nzc;       // coarse
    long rs, ms, crs, cms;   // fine / coarse row and plane strides
    int kcc, nty;            // coarse planes per chunk, tiles in y
    int kcbeg, kcend;        // coarse planes [kcbeg, kcend) produced by this launch (whole grid / slab: 0, nzc)
    T a0, a1, a2, a3, a4, a5, a6;
    T *uc0;                  // optional: the coarse level's first sweep from a zero guess, uc0 = scale_c * (bc * dinv_c)
    T dinv_c, scale_c;
    const T *far_hi;         // inner z-slab, optional: plane nz + 1 of u (the rank above's plane 1; interior origin of that plane).
                             // With it -- and u's and b's hi ghost planes valid -- the residual of plane nz (the neighbour's first
                             // plane) is evaluated here too, so the last coarse plane is complete: no partial plane, no finish
};

template <typename T, int WX>
__global__ void __launch_bounds__(64 * WX) k_resrestrict(const RRArgs<T> a) {
    constexpr int VX = 16 / sizeof(T);           // fine columns per lane
    constexpr int NCJ = VX / 2;                  // coarse columns per lane
    constexpr int RR = 5, TX = 64 * VX * WX, LW = TX + 2 * VX;
    __shared__ __attribute__((aligned(16))) T lds[2][RR][LW];
    __shared__ __attribute__((aligned(16))) T rt[RR][LW];
    using VT = V16<T>;
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int xl = VX * tid, x0 = xl;            // full-row tile: tx = 0
    const int yb = 4 * ty;
    const int kc0 = a.kcbeg + tz * a.kcc, kc1 = min(kc0 + a.kcc, a.kcend);
    if (kc0 >= kc1) return;
    const int z0 = 2 * kc0, z1u = 2 * kc1 + 1;   // fine planes z0 .. z1u-1 (= 2*kc1, shared with the next chunk)
    // z-slab of a rank that is not the last one: nz = 2 nzc, the closing plane 2 nzc belongs to the next rank.  The last
    // coarse plane is then left PARTIAL (its dk = 0, 1 terms); k_restrict_finish adds the dk = 2 terms from the
    // neighbour's residual plane in the same order, so the sum is the one of the whole grid.
    const int z1 = a.far_hi ? z1u : min(z1u, a.nz);
    const bool xok = x0 < a.nx;
    const bool lastvec = (x0 + VX > a.nx);
    bool rok[RR];
#pragma unroll
    for (int r = 0; r < RR; r++) rok[r] = xok && (yb + r < a.ny);
    const long rowoff = (long)yb * a.rs;
    const T *up_ = a.u + rowoff + x0, *bp_ = a.b + rowoff + x0;
    const T *fh_ = a.far_hi ? a.far_hi + rowoff + x0 : nullptr;
    auto uplane = [&](int p) -> const T * { return (fh_ && p == a.nz + 1) ? fh_ : up_ + (long)p * a.ms; };
    const bool okS = xok, okN = xok && (yb + RR <= a.ny);
    // coarse ownership: lane `tid` -> coarse columns NCJ*tid .. NCJ*tid+NCJ-1, coarse rows 2*ty + {0,1}
    const int jc0 = NCJ * tid;
    const bool crow0 = (2 * ty < a.nyc), crow1 = (2 * ty + 1 < a.nyc);
    const T w2[3][3] = {{(T)0.0625, (T)0.125, (T)0.0625}, {(T)0.125, (T)0.25, (T)0.125}, {(T)0.0625, (T)0.125, (T)0.0625}};

    VT um[RR], uc[RR], up[RR], uq[RR], bcur[RR], bn[RR];
    VT hS, hN, hSn = v16_zero<T>(), hNn = hSn;
#pragma unroll
    for (int r = 0; r < RR; r++) {
        const long ro = (long)r * a.rs;
        um[r] = ldv(up_ + (long)(z0 - 1) * a.ms + ro, rok[r]);
        uc[r] = ldv(up_ + (long)z0 * a.ms + ro, rok[r]);
        up[r] = ldv(uplane(z0 + 1) + ro, rok[r]);
        // rows 0 and RR-1 are shared with the neighbouring tiles: cached loads, so the second reader hits L2
        bcur[r] = (r == 0 || r == RR - 1) ? ldv(bp_ + (long)z0 * a.ms + ro, rok[r]) : ldv_stream(bp_ + (long)z0 * a.ms + ro, rok[r]);
        *reinterpret_cast<VT *>(&lds[0][r][xl + VX]) = uc[r];
    }
    if (tid == 0) {
#pragma unroll
        for (int r = 0; r < RR; r++) { lds[0][r][VX - 1] = (T)0; lds[1][r][VX - 1] = (T)0; lds[0][r][TX + VX] = (T)0; lds[1][r][TX + VX] = (T)0; }
    }
    hS = ldv(up_ + (long)z0 * a.ms - a.rs, okS);
    hN = ldv(up_ + (long)z0 * a.ms + (long)RR * a.rs, okN);

    T acc[2][NCJ], accn[2][NCJ];
#pragma unroll
    for (int cl = 0; cl < 2; cl++)
#pragma unroll
        for (int q = 0; q < NCJ; q++) { acc[cl][q] = (T)0; accn[cl][q] = (T)0; }

    for (int z = z0; z < z1; z++) {
        const int buf = (z - z0) & 1;
        const bool more = (z + 1 < z1);
        if (more) {
#pragma unroll
            for (int r = 0; r < RR; r++) {
                const long ro = (long)r * a.rs;
                uq[r] = ldv(uplane(z + 2) + ro, rok[r]);
                bn[r] = (r == 0 || r == RR - 1) ? ldv(bp_ + (long)(z + 1) * a.ms + ro, rok[r]) : ldv_stream(bp_ + (long)(z + 1) * a.ms + ro, rok[r]);
            }
            hSn = ldv(up_ + (long)(z + 1) * a.ms - a.rs, okS);
            hNn = ldv(up_ + (long)(z + 1) * a.ms + (long)RR * a.rs, okN);
        }
        __syncthreads();                          // u tile of plane z complete; rt of plane z-1 consumed
#pragma unroll
        for (int r = 0; r < RR; r++) {
            const T Wv = lds[buf][r][xl + VX - 1], Ev = lds[buf][r][xl + 2 * VX];
            const VT s2 = (r == 0) ? hS : uc[r > 0 ? r - 1 : 0];
            const VT n2 = (r == RR - 1) ? hN : uc[r < RR - 1 ? r + 1 : r];
            VT res;
#pragma unroll
            for (int e = 0; e < VX; e++) {
                const T wv = (e == 0) ? Wv : uc[r].v[e > 0 ? e - 1 : 0];
                const T ev = (e == VX - 1) ? Ev : uc[r].v[e < VX - 1 ? e + 1 : e];
                T t = a.a0 * um[r].v[e];
                t = t + a.a1 * s2.v[e];
                t = t + a.a2 * wv;
                t = t + a.a3 * uc[r].v[e];
                t = t + a.a4 * ev;
                t = t + a.a5 * n2.v[e];
                t = t + a.a6 * up[r].v[e];
                res.v[e] = bcur[r].v[e] - t;
                if ((lastvec && x0 + e >= a.nx) || !rok[r]) res.v[e] = (T)0;
            }
            *reinterpret_cast<VT *>(&rt[r][xl + VX]) = res;
        }
        __syncthreads();                          // residual plane z visible
        {
            const bool even = ((z & 1) == 0);
            const T wk = even ? (T)0.25 : (T)0.5;
#pragma unroll
            for (int cl = 0; cl < 2; cl++) {
                if (!(cl == 0 ? crow0 : crow1)) continue;
#pragma unroll
                for (int q = 0; q < NCJ; q++) {
                    if (jc0 + q >= a.nxc) continue;
#pragma unroll
                    for (int di = 0; di < 3; di++) {
                        const T *row = &rt[2 * cl + di][2 * (jc0 + q) + VX];
#pragma unroll
                        for (int dj = 0; dj < 3; dj++) {
                            const T p = (wk * w2[di][dj]) * row[dj];
                            acc[cl][q] += p;
                            if (even) accn[cl][q] += p;
                        }
                    }
                }
            }
            if (even) {
                const int kc = z / 2 - 1;         // completed coarse plane
#pragma unroll
                for (int cl = 0; cl < 2; cl++)
#pragma unroll
                    for (int q = 0; q < NCJ; q++) {
                        if (kc >= kc0 && (cl == 0 ? crow0 : crow1) && jc0 + q < a.nxc) {
                            const long oc = (long)kc * a.cms + (long)(2 * ty + cl) * a.crs + jc0 + q;
                            a.bc[oc] = acc[cl][q];
                            if (a.uc0) { const T z = acc[cl][q] * a.dinv_c; a.uc0[oc] = a.scale_c * z; }   // k_jacobi_zero's arithmetic
                        }
                        acc[cl][q] = accn[cl][q]; accn[cl][q] = (T)0;
                    }
            }
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < RR; r++) {
                *reinterpret_cast<VT *>(&lds[buf ^ 1][r][xl + VX]) = up[r];
                um[r] = uc[r]; uc[r] = up[r]; up[r] = uq[r]; bcur[r] = bn[r];
            }
            hS = hSn; hN = hNn;
        }
    }
    if (z1u > z1) {                               // slab: partial last coarse plane (planes 2 kc, 2 kc + 1 of kc = kc1 - 1)
#pragma unroll
        for (int cl = 0; cl < 2; cl++)
#pragma unroll
            for (int q = 0; q < NCJ; q++)
                if ((cl == 0 ? crow0 : crow1) && jc0 + q < a.nxc)
                    a.bc[(long)(kc1 - 1) * a.cms + (long)(2 * ty + cl) * a.crs + jc0 + q] = acc[cl][q];
    }
}


// ------------------------------------------------------------------------------------------
// Register / shuffle forms of the two fused fine-level kernels (round 2).
//
// Same tiles, same arithmetic and summation order as k_resrestrict and k_stencil<MODE_PJACOBI> -- results are bit-identical --
// but shaped like k_jacobi2r: a block is a full-row tile marching along z; lane t owns the column pair (fp32: quad) x = VX t ..
// of ALL rows of the tile including the two halo rows, so y neighbours are its own registers; x neighbours come by wave
// shuffle, the two values that cross a wave boundary through a few bytes of LDS; ONE barrier per plane.  No plane tiles in
// LDS (the old forms held 116-124 KB and were latency bound at one block per CU with one plane of prefetch): the planes live in a
// ring of NP = PD + 3 register sets indexed statically (the marching loop is unrolled by the ring period: no register
// rotation), loaded PD planes ahead of their first use.  Row bases are wave-uniform, the lane offset is a 32-bit constant.
// ------------------------------------------------------------------------------------------
// Loads of the row kernels are UNCONDITIONAL: the row / plane index is clamped on the scalar unit to something that exists
// (rows beyond the ghost row alias the ghost row, planes beyond the chunk alias its last plane), and the shapes are restricted to
// full rows (nx + 1 a multiple of the wave row) so that no lane lies outside the grid.  No exec masking, no branch per load.

template <typename T, int WX, int PD, int FORM>
__global__ void __launch_bounds__(64 * WX) k_rrrow(const RRArgs<T> a) {
    constexpr bool DPP = (FORM & 2) != 0;
    constexpr int VX = 16 / sizeof(T), NCJ = VX / 2;
    constexpr int RR = 5, R1 = RR + 2, NP = PD + 3;
    __shared__ T eW[2][RR][WX], eE[2][RR][WX], eR[2][RR][WX];
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int x0 = VX * tid, yb = 4 * ty;
    const int kc0 = a.kcbeg + tz * a.kcc, kc1 = min(kc0 + a.kcc, a.kcend);
    if (kc0 >= kc1) return;
    const int z0 = 2 * kc0, z1u = 2 * kc1 + 1;
    const int z1 = a.far_hi ? z1u : min(z1u, a.nz);          // planes z0 .. z1-1 are processed; plane z1 is the last one read
    const bool lastlane = (tid == 64 * WX - 1);               // its last element is the right ghost column x = nx: stays 0
    const unsigned lb = (unsigned)(x0 * (int)sizeof(T));
    bool rok[RR];                                             // wave-uniform: rows of the tile inside the grid
#pragma unroll
    for (int r = 0; r < RR; r++) rok[r] = (yb + r < a.ny);
    // wave-uniform row bases; rows beyond the ghost row y = ny alias it (zeros: a ghost ROW is always a global boundary)
    long uro[R1], bro[RR];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) uro[rr] = (long)min(yb - 1 + rr, a.ny) * a.rs;
#pragma unroll
    for (int r = 0; r < RR; r++) bro[r] = (long)min(yb + r, a.ny) * a.rs;
    auto uplane = [&](int p) -> const T * {                   // planes beyond z1 alias plane z1 (loaded, never used)
        const int pp = min(p, z1);
        return (a.far_hi && pp == a.nz + 1) ? a.far_hi : a.u + (long)pp * a.ms;
    };
    const int jc0 = NCJ * tid;
    const bool crow0 = (2 * ty < a.nyc), crow1 = (2 * ty + 1 < a.nyc);
    const T w2[3][3] = {{(T)0.0625, (T)0.125, (T)0.0625}, {(T)0.125, (T)0.25, (T)0.125}, {(T)0.0625, (T)0.125, (T)0.0625}};

    VT U[NP][R1], B[NP][RR];
    // ---- prologue: u planes z0-1 .. z0+PD into slots NP-1, 0 .. PD; b planes z0 .. z0+PD-1 into slots 0 .. PD-1 ----
#pragma unroll
    for (int q = -1; q <= PD; q++) {
        const T *pl = uplane(z0 + q);
#pragma unroll
        for (int rr = 0; rr < R1; rr++) U[(q + NP) % NP][rr] = ldrow(pl + uro[rr], lb);
    }
#pragma unroll
    for (int q = 0; q < PD; q++) {
        const T *pl = a.b + (long)min(z0 + q, z1 - 1) * a.ms;
#pragma unroll
        for (int r = 0; r < RR; r++) B[q][r] = (r == 0 || r == RR - 1) ? ldrow(pl + bro[r], lb) : ldrow_stream(pl + bro[r], lb);
    }
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int r = 0; r < RR; r++) {
            if (lane == 0) eW[z0 & 1][r][w] = U[0][r + 1].v[0];
            else eE[z0 & 1][r][w] = U[0][r + 1].v[VX - 1];
        }
    }
    T acc[2][NCJ], accn[2][NCJ];
#pragma unroll
    for (int cl = 0; cl < 2; cl++)
#pragma unroll
        for (int q = 0; q < NCJ; q++) { acc[cl][q] = (T)0; accn[cl][q] = (T)0; }
    __syncthreads();

    for (int zb = z0; zb < z1; zb += NP) {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int z = zb + k;
            if (z < z1) {
                const int cm = (k + NP - 1) % NP, cc = k, cp = (k + 1) % NP, cl_ = (k + PD + 1) % NP, cb = (k + PD) % NP;
                // ---- loads first used PD steps from now ----
                {
                    const T *pl = uplane(z + PD + 1);
#pragma unroll
                    for (int rr = 0; rr < R1; rr++) U[cl_][rr] = ldrow(pl + uro[rr], lb);
                    const T *pb = a.b + (long)min(z + PD, z1 - 1) * a.ms;
#pragma unroll
                    for (int r = 0; r < RR; r++) B[cb][r] = (r == 0 || r == RR - 1) ? ldrow(pb + bro[r], lb) : ldrow_stream(pb + bro[r], lb);
                }
                // ---- residual of plane z on the 5 rows of the tile.  The plane z+1 term and b enter the canonical sum LAST:
                //      everything that needs only planes z-1 and z comes first, so the loads issued one step ago have almost
                //      two steps to arrive ----
                VT res[RR];
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    const VT &c = U[cc][r + 1];
                    T Wv = lane_up<DPP>(c.v[VX - 1]), Ev = lane_dn<DPP>(c.v[0]);
                    if (lane == 0) Wv = (w > 0) ? eE[z & 1][r][w - 1] : (T)0;
                    if (lane == 63) Ev = (w < WX - 1) ? eW[z & 1][r][w + 1] : (T)0;
                    if constexpr (sizeof(T) == 4) {
                        // fp32: the same sums on the 4-wide vector type, so that the compiler issues packed v_pk_mul_f32 / v_pk_add_f32
                        auto ld4 = [](const VT &x) { f4v q; q.x = x.v[0]; q.y = x.v[1]; q.z = x.v[2]; q.w = x.v[3]; return q; };
                        const f4v C4 = ld4(c);
                        f4v W4; W4.x = Wv; W4.y = c.v[0]; W4.z = c.v[1]; W4.w = c.v[2];
                        f4v E4; E4.x = c.v[1]; E4.y = c.v[2]; E4.z = c.v[3]; E4.w = Ev;
                        f4v t4 = a.a0 * ld4(U[cm][r + 1]);
                        t4 = t4 + a.a1 * ld4(U[cc][r]);
                        t4 = t4 + a.a2 * W4;
                        t4 = t4 + a.a3 * C4;
                        t4 = t4 + a.a4 * E4;
                        t4 = t4 + a.a5 * ld4(U[cc][r + 2]);
                        res[r].v[0] = t4.x; res[r].v[1] = t4.y; res[r].v[2] = t4.z; res[r].v[VX - 1] = t4.w;
                    } else {
#pragma unroll
                    for (int e = 0; e < VX; e++) {
                        const T wv = (e == 0) ? Wv : c.v[e > 0 ? e - 1 : 0];
                        const T ev = (e == VX - 1) ? Ev : c.v[e < VX - 1 ? e + 1 : e];
                        T t = a.a0 * U[cm][r + 1].v[e];
                        t = t + a.a1 * U[cc][r].v[e];
                        t = t + a.a2 * wv;
                        t = t + a.a3 * c.v[e];
                        t = t + a.a4 * ev;
                        t = t + a.a5 * U[cc][r + 2].v[e];
                        res[r].v[e] = t;
                    }
                    }
                }
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    if constexpr (sizeof(T) == 4) {
                        auto ld4 = [](const VT &x) { f4v q; q.x = x.v[0]; q.y = x.v[1]; q.z = x.v[2]; q.w = x.v[3]; return q; };
                        const f4v t4 = ld4(res[r]) + a.a6 * ld4(U[cp][r + 1]);
                        const f4v r4 = ld4(B[cc][r]) - t4;
                        res[r].v[0] = rok[r] ? r4.x : (T)0; res[r].v[1] = rok[r] ? r4.y : (T)0; res[r].v[2] = rok[r] ? r4.z : (T)0; res[r].v[VX - 1] = rok[r] ? r4.w : (T)0;
                    } else {
#pragma unroll
                    for (int e = 0; e < VX; e++) {
                        const T t = res[r].v[e] + a.a6 * U[cp][r + 1].v[e];
                        res[r].v[e] = rok[r] ? B[cc][r].v[e] - t : (T)0;
                    }
                    }
                    if (lastlane) res[r].v[VX - 1] = (T)0;
                }
                // ---- wave-edge values of plane z+1 for the next step (other buffer) and of this residual plane ----
                if (lane == 0 || lane == 63) {
#pragma unroll
                    for (int r = 0; r < RR; r++) {
                        if (lane == 0) { eW[(z + 1) & 1][r][w] = U[cp][r + 1].v[0]; eR[z & 1][r][w] = res[r].v[0]; }
                        else eE[(z + 1) & 1][r][w] = U[cp][r + 1].v[VX - 1];
                    }
                }
                __syncthreads();      // edges of u(z+1) and of the residual of plane z are visible
                // ---- full weighting: the running sums of coarse planes z/2-1 (dk = 2) and z/2 (dk = 0) / (z-1)/2 (dk = 1) ----
                {
                    const bool even = ((z & 1) == 0);
                    const T wk = even ? (T)0.25 : (T)0.5;
                    T nx_[RR];
#pragma unroll
                    for (int r = 0; r < RR; r++) {
                        nx_[r] = lane_dn<DPP>(res[r].v[0]);
                        if (lane == 63) nx_[r] = (w < WX - 1) ? eR[z & 1][r][w + 1] : (T)0;
                    }
#pragma unroll
                    for (int cl = 0; cl < 2; cl++) {
                        if (!(cl == 0 ? crow0 : crow1)) continue;
#pragma unroll
                        for (int q = 0; q < NCJ; q++) {
#pragma unroll
                            for (int di = 0; di < 3; di++) {
#pragma unroll
                                for (int dj = 0; dj < 3; dj++) {
                                    const int e = 2 * q + dj;
                                    const T val = (e < VX) ? res[2 * cl + di].v[e < VX ? e : 0] : nx_[2 * cl + di];
                                    const T p = (wk * w2[di][dj]) * val;
                                    acc[cl][q] += p;
                                    if (even) accn[cl][q] += p;
                                }
                            }
                        }
                    }
                    if (even) {
                        const int kc = z / 2 - 1;     // completed coarse plane
#pragma unroll
                        for (int cl = 0; cl < 2; cl++)
#pragma unroll
                            for (int q = 0; q < NCJ; q++) {
                                if (kc >= kc0 && (cl == 0 ? crow0 : crow1) && jc0 + q < a.nxc) {
                                    const long oc = (long)kc * a.cms + (long)(2 * ty + cl) * a.crs + jc0 + q;
                                    a.bc[oc] = acc[cl][q];
                                    if (a.uc0) { const T zq = acc[cl][q] * a.dinv_c; a.uc0[oc] = a.scale_c * zq; }
                                }
                                acc[cl][q] = accn[cl][q]; accn[cl][q] = (T)0;
                            }
                    }
                }
            }
        }
    }
    if (z1u > z1) {                               // slab without the far plane: partial last coarse plane
#pragma unroll
        for (int cl = 0; cl < 2; cl++)
#pragma unroll
            for (int q = 0; q < NCJ; q++)
                if ((cl == 0 ? crow0 : crow1) && jc0 + q < a.nxc)
                    a.bc[(long)(kc1 - 1) * a.cms + (long)(2 * ty + cl) * a.crs + jc0 + q] = acc[cl][q];
    }
}

template <typename T>
__device__ __forceinline__ V16<T> ldrowp(const T *row_uniform, unsigned lane_bytes, bool ok) {
    return ok ? *reinterpret_cast<const V16<T> *>(reinterpret_cast<const char *>(row_uniform) + lane_bytes) : v16_zero<T>();
}
template <typename T>
__device__ __forceinline__ V16<T> ldrowp_stream(const T *row_uniform, unsigned lane_bytes, bool ok) {
    return ldv_stream(reinterpret_cast<const T *>(reinterpret_cast<const char *>(row_uniform) + lane_bytes), ok);
}
// fused prolongation + first post-smoothing sweep, register / shuffle form: unew = J(u + P uc)
template <typename T>
struct PJArgs {
    const T *u, *b, *uc;
    T *out;
    int nx, ny, nz, nxc, nyc, nzc;
    long rs, ms, crs, cms;
    int nty, zc, zbeg, zend;
    T a0, a1, a2, a3, a4, a5, a6, dinv, scale;
};

// s += the interpolant of one fine row from the coarse registers: C[j][0 .. NCL-1] = coarse row j at the columns x0/2-1 ..;
// `two` parent planes (even fine plane: A then Bc, weight 1/2 each) or one (odd: A alone, weight 1); the fine row has one parent
// row j0 (odd row, weight 1) or two, j0 and j0+1 (even row).  Terms in the order of prolong_vec: ascending (kc, ic, jc).
template <typename T, int TYC, int NCL>
__device__ __forceinline__ V16<T> corr_row(const T (&A)[TYC][NCL], const T (&Bc)[TYC][NCL], bool two, int j0, bool tworows) {
    constexpr int VX = 16 / sizeof(T);
    const T wk = two ? (T)0.5 : (T)1, wi = tworows ? (T)0.5 : (T)1;
    const T wh = wk * (wi * (T)0.5), w1 = wk * (wi * (T)1);
    V16<T> s = v16_zero<T>();
#pragma unroll
    for (int qk = 0; qk < 2; qk++) {
        if (qk == 1 && !two) break;
#pragma unroll
        for (int qi = 0; qi < 2; qi++) {
            if (qi == 1 && !tworows) break;
            const T c0 = qk == 0 ? A[j0 + qi][0] : Bc[j0 + qi][0];
            const T c1 = qk == 0 ? A[j0 + qi][1] : Bc[j0 + qi][1];
            s.v[0] += wh * c0;
            s.v[0] += wh * c1;
            s.v[1] += w1 * c1;
            if (VX == 4) {
                const T c2 = qk == 0 ? A[j0 + qi][NCL - 1] : Bc[j0 + qi][NCL - 1];
                s.v[VX - 2] += wh * c1;
                s.v[VX - 2] += wh * c2;
                s.v[VX - 1] += w1 * c2;
            }
        }
    }
    return s;
}

template <typename T, int WX, int PD, int FORM>
__global__ void __launch_bounds__(64 * WX) k_pjrow(const PJArgs<T> a) {
    constexpr bool UNC = (FORM & 1) != 0, DPP = (FORM & 2) != 0;
    constexpr int VX = 16 / sizeof(T), NCL = VX / 2 + 1;
    constexpr int TY = 4, R1 = TY + 2, TYC = 4, NP = PD + 3;
    __shared__ T eW[2][TY][WX], eE[2][TY][WX];
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int x0 = VX * tid, yb = TY * ty;
    const int z0 = a.zbeg + tz * a.zc, z1 = min(z0 + a.zc, a.zend);
    if (z0 >= z1) return;
    const bool xok = x0 < a.nx, lastvec = (x0 + VX > a.nx);
    const unsigned lb = (unsigned)(x0 * (int)sizeof(T));
    bool ok[R1];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) ok[rr] = xok && (yb - 1 + rr <= a.ny);
    const T *ub = a.u + (long)(yb - 1) * a.rs, *bb = a.b + (long)yb * a.rs;
    T *ob = a.out + (long)yb * a.rs;
    // FORM bit 0: unconditional loads, rows / planes clamped on the scalar unit (full-row shapes: no lane outside the grid)
    long uro[R1], bro[TY];
#pragma unroll
    for (int rr = 0; rr < R1; rr++) uro[rr] = (long)min(yb - 1 + rr, a.ny) * a.rs;
#pragma unroll
    for (int j = 0; j < TY; j++) bro[j] = (long)min(yb + j, a.ny) * a.rs;
    auto LDU = [&](int p, int rr) -> VT {
        if (UNC) return ldrow(a.u + (long)min(p, z1) * a.ms + uro[rr], lb);
        return ldrowp(ub + (long)p * a.ms + (long)rr * a.rs, lb, ok[rr] && p <= z1);
    };
    auto LDB = [&](int p, int j) -> VT {
        if (UNC) return ldrow_stream(a.b + (long)min(p, z1 - 1) * a.ms + bro[j], lb);
        return ldrowp_stream(bb + (long)p * a.ms + (long)j * a.rs, lb, ok[j + 1] && (yb + j < a.ny) && (p < z1));
    };
    // coarse rows icb .. icb+3 at the coarse columns x0/2 - 1 + q
    const int icb = yb / 2 - 1, jcb = x0 / 2 - 1;
    const T *cb_ = a.uc + (long)icb * a.crs + jcb;
    long cro[TYC];
#pragma unroll
    for (int j = 0; j < TYC; j++) cro[j] = (long)min(icb + j, a.nyc) * a.crs;
    auto ldc = [&](T (&C_)[TYC][NCL], int kc) {
        const bool kv = (kc >= -1 && kc <= a.nzc);
        const T *plc = a.uc + (long)max(-1, min(kc, a.nzc)) * a.cms + jcb;
#pragma unroll
        for (int j = 0; j < TYC; j++)
#pragma unroll
            for (int q = 0; q < NCL; q++) {
                if (UNC) C_[j][q] = plc[cro[j] + q];
                else C_[j][q] = (kv && xok && icb + j <= a.nyc && jcb + q <= a.nxc) ? cb_[(long)kc * a.cms + (long)j * a.crs + q] : (T)0;
            }
    };
    T cLo[TYC][NCL], cHi[TYC][NCL], cSt[TYC][NCL];
    VT U[NP][R1], B[NP][TY];
    // u + P uc on the six rows of plane p held in U[slot]: row rr <-> fine row yb-1+rr (yb a multiple of 4: rr even = odd fine row,
    // one parent row rr/2; rr odd = even fine row, parent rows (rr-1)/2 and (rr+1)/2).  An odd plane p has ONE parent plane,
    // p >> 1 (passed as A); an even one two, (p >> 1) - 1 and p >> 1 (A, Bc).
#define PJ_CORRECT(slot, A_, B_, two_)                                                                   \
    do {                                                                                                 \
        _Pragma("unroll") for (int rr = 0; rr < R1; rr++)                                                \
            if (ok[rr]) U[slot][rr] = vadd(U[slot][rr], corr_row<T, TYC, NCL>(A_, B_, two_, rr / 2, (rr & 1) != 0)); \
    } while (0)

    // ---- prologue ----
    {
        const int m = z0 >> 1;                    // floor also for z0 = -... (z0 >= 0 here)
        ldc(cLo, m - 1); ldc(cHi, m); ldc(cSt, m + 1);
    }
#pragma unroll
    for (int q = -1; q <= PD; q++) {
        const int p = z0 + q, slot = (q + NP) % NP;
#pragma unroll
        for (int rr = 0; rr < R1; rr++) U[slot][rr] = LDU(p, rr);
    }
#pragma unroll
    for (int q = 0; q < PD; q++) {
#pragma unroll
        for (int j = 0; j < TY; j++)
            B[q][j] = LDB(z0 + q, j);
    }
    if (z0 & 1) {           // z0 odd: plane z0-1 is even (parents m-1, m), plane z0 odd (parent m)
        PJ_CORRECT(NP - 1, cLo, cHi, true);
        PJ_CORRECT(0, cHi, cHi, false);
    } else {                // z0 even: plane z0-1 is odd (parent m-1), plane z0 even (parents m-1, m)
        PJ_CORRECT(NP - 1, cLo, cLo, false);
        PJ_CORRECT(0, cLo, cHi, true);
    }
    if (lane == 0 || lane == 63) {
#pragma unroll
        for (int j = 0; j < TY; j++) {
            if (lane == 0) eW[z0 & 1][j][w] = U[0][j + 1].v[0];
            else eE[z0 & 1][j][w] = U[0][j + 1].v[VX - 1];
        }
    }
    __syncthreads();

    for (int zb = z0; zb < z1; zb += NP) {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int z = zb + k;
            if (z < z1) {
                const int cm = (k + NP - 1) % NP, cc = k, cp = (k + 1) % NP, cl_ = (k + PD + 1) % NP, cbs = (k + PD) % NP;
                // ---- loads first used PD steps from now ----
                {
#pragma unroll
                    for (int rr = 0; rr < R1; rr++) U[cl_][rr] = LDU(z + PD + 1, rr);
#pragma unroll
                    for (int j = 0; j < TY; j++) B[cbs][j] = LDB(z + PD, j);
                }
                // ---- correct plane p = z+1 (first use: the z neighbour of this step's sweep), publish its wave edges ----
                {
                    const int p = z + 1;
                    if ((p & 1) == 0) {           // a new pair of parent planes: (p/2 - 1, p/2); fetch p/2 + 1 for two steps from now
#pragma unroll
                        for (int j = 0; j < TYC; j++)
#pragma unroll
                            for (int q = 0; q < NCL; q++) { cLo[j][q] = cHi[j][q]; cHi[j][q] = cSt[j][q]; }
                        ldc(cSt, (p >> 1) + 1);
                        PJ_CORRECT(cp, cLo, cHi, true);
                    } else {
                        PJ_CORRECT(cp, cHi, cHi, false);
                    }
                    if (lane == 0 || lane == 63) {
#pragma unroll
                        for (int j = 0; j < TY; j++) {
                            if (lane == 0) eW[p & 1][j][w] = U[cp][j + 1].v[0];
                            else eE[p & 1][j][w] = U[cp][j + 1].v[VX - 1];
                        }
                    }
                }
                // ---- sweep of plane z ----
#pragma unroll
                for (int j = 0; j < TY; j++) {
                    const VT &c = U[cc][j + 1];
                    T Wv = lane_up<DPP>(c.v[VX - 1]), Ev = lane_dn<DPP>(c.v[0]);
                    if (lane == 0) Wv = (w > 0) ? eE[z & 1][j][w - 1] : (T)0;
                    if (lane == 63) Ev = (w < WX - 1) ? eW[z & 1][j][w + 1] : (T)0;
                    VT o;
#pragma unroll
                    for (int e = 0; e < VX; e++) {
                        const T wv = (e == 0) ? Wv : c.v[e > 0 ? e - 1 : 0];
                        const T ev = (e == VX - 1) ? Ev : c.v[e < VX - 1 ? e + 1 : e];
                        T t = a.a0 * U[cm][j + 1].v[e];
                        t = t + a.a1 * U[cc][j].v[e];
                        t = t + a.a2 * wv;
                        t = t + a.a3 * c.v[e];
                        t = t + a.a4 * ev;
                        t = t + a.a5 * U[cc][j + 2].v[e];
                        t = t + a.a6 * U[cp][j + 1].v[e];
                        const T res = B[cc][j].v[e] - t;
                        const T zz = res * a.dinv;
                        o.v[e] = c.v[e] + a.scale * zz;
                        if (lastvec && x0 + e >= a.nx) o.v[e] = (T)0;
                    }
                    if (xok && yb + j < a.ny)
                        stv_stream(reinterpret_cast<T *>(reinterpret_cast<char *>(ob + (long)z * a.ms + (long)j * a.rs) + lb), o);
                }
                __syncthreads();      // edges of plane z+1 visible; the buffer of plane z is free for plane z+2
            }
        }
    }
#undef PJ_CORRECT
}


// ------------------------------------------------------------------------------------------
// One Richardson + Jacobi sweep in the register / shuffle shape of k_pjrow (no correction): full-row tile of 4 rows, lane = column
// pair of all 6 rows, DPP lane shifts, unconditional clamped loads, static ring of PD + 3 planes (without the coarse parents
// there is room for two planes of prefetch), one barrier per plane.  NORM: also the sum of squares of the residual of the
// INPUT field (the norm that closes a cycle fused with the first sweep of the next one, mgk_jacobi_sumsq_*).
// Same expressions, same order as k_stencil<MODE_JACOBI / MODE_JNORM>: bit-identical output.
// ------------------------------------------------------------------------------------------
template <typename T, int WX, int PD, bool NORM>
__global__ void __launch_bounds__(64 * WX) k_jrow(const PJArgs<T> a, double *partials) {
    constexpr int VX = 16 / sizeof(T);
    constexpr int TY = 4, R1 = TY + 2, NP = PD + 3;
    __shared__ T eW[2][TY][WX], eE[2][TY][WX];
    using VT = V16<T>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int ty = bid % a.nty, tz = bid / a.nty;
    const int x0 = VX * tid, yb = TY * ty;
    const int z0 = a.zbeg + tz * a.zc, z1 = min(z0 + a.zc, a.zend);
    double acc = 0.0;
    if (z0 < z1) {
        const bool lastlane = (tid == 64 * WX - 1);
        const unsigned lb = (unsigned)(x0 * (int)sizeof(T));
        long uro[R1], bro[TY];
#pragma unroll
        for (int rr = 0; rr < R1; rr++) uro[rr] = (long)min(yb - 1 + rr, a.ny) * a.rs;
#pragma unroll
        for (int j = 0; j < TY; j++) bro[j] = (long)min(yb + j, a.ny) * a.rs;
        VT U[NP][R1], B[NP][TY];
#pragma unroll
        for (int q = -1; q <= PD; q++) {
            const T *pl = a.u + (long)min(z0 + q, z1) * a.ms;
#pragma unroll
            for (int rr = 0; rr < R1; rr++) U[(q + NP) % NP][rr] = ldrow(pl + uro[rr], lb);
        }
#pragma unroll
        for (int q = 0; q < PD; q++) {
            const T *pl = a.b + (long)min(z0 + q, z1 - 1) * a.ms;
#pragma unroll
            for (int j = 0; j < TY; j++) B[q][j] = ldrow_stream(pl + bro[j], lb);
        }
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int j = 0; j < TY; j++) {
                if (lane == 0) eW[z0 & 1][j][w] = U[0][j + 1].v[0];
                else eE[z0 & 1][j][w] = U[0][j + 1].v[VX - 1];
            }
        }
        __syncthreads();
        for (int zb = z0; zb < z1; zb += NP) {
#pragma unroll
            for (int k = 0; k < NP; k++) {
                const int z = zb + k;
                if (z < z1) {
                    const int cm = (k + NP - 1) % NP, cc = k, cp = (k + 1) % NP;
                    {
                        const T *pl = a.u + (long)min(z + PD + 1, z1) * a.ms;
#pragma unroll
                        for (int rr = 0; rr < R1; rr++) U[(k + PD + 1) % NP][rr] = ldrow(pl + uro[rr], lb);
                        const T *pb = a.b + (long)min(z + PD, z1 - 1) * a.ms;
#pragma unroll
                        for (int j = 0; j < TY; j++) B[(k + PD) % NP][j] = ldrow_stream(pb + bro[j], lb);
                    }
                    VT part[TY];
#pragma unroll
                    for (int j = 0; j < TY; j++) {
                        const VT &c = U[cc][j + 1];
                        T Wv = lane_up<true>(c.v[VX - 1]), Ev = lane_dn<true>(c.v[0]);
                        if (lane == 0) Wv = (w > 0) ? eE[z & 1][j][w - 1] : (T)0;
                        if (lane == 63) Ev = (w < WX - 1) ? eW[z & 1][j][w + 1] : (T)0;
#pragma unroll
                        for (int e = 0; e < VX; e++) {
                            const T wv = (e == 0) ? Wv : c.v[e > 0 ? e - 1 : 0];
                            const T ev = (e == VX - 1) ? Ev : c.v[e < VX - 1 ? e + 1 : e];
                            T t = a.a0 * U[cm][j + 1].v[e];
                            t = t + a.a1 * U[cc][j].v[e];
                            t = t + a.a2 * wv;
                            t = t + a.a3 * c.v[e];
                            t = t + a.a4 * ev;
                            t = t + a.a5 * U[cc][j + 2].v[e];
                            part[j].v[e] = t;
                        }
                    }
                    if (lane == 0 || lane == 63) {            // wave edges of plane z+1 for the next step (first touch of that plane)
#pragma unroll
                        for (int j = 0; j < TY; j++) {
                            if (lane == 0) eW[(z + 1) & 1][j][w] = U[cp][j + 1].v[0];
                            else eE[(z + 1) & 1][j][w] = U[cp][j + 1].v[VX - 1];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < TY; j++) {
                        VT o;
                        const bool rowin = (yb + j < a.ny);
#pragma unroll
                        for (int e = 0; e < VX; e++) {
                            const T t = part[j].v[e] + a.a6 * U[cp][j + 1].v[e];
                            const T res = B[cc][j].v[e] - t;
                            const T zz = res * a.dinv;
                            o.v[e] = U[cc][j + 1].v[e] + a.scale * zz;
                            if (NORM && rowin && !(lastlane && e == VX - 1)) acc += (double)res * (double)res;
                        }
                        if (lastlane) o.v[VX - 1] = (T)0;
                        if (rowin)
                            stv_stream(reinterpret_cast<T *>(reinterpret_cast<char *>(a.out + (long)z * a.ms + (long)(yb + j) * a.rs) + lb), o);
                    }
                    __syncthreads();
                }
            }
        }
    }
    if (NORM) {
        __shared__ double red[16];
        const double s = block_sum(acc, red);
        if (threadIdx.x == 0) partials[blockIdx.x] = s;
    }
}
template <typename T, bool NORM>
static int launch_jrow(mgk_ctx *c, const mgk_geom *g, const StArgs<T> &a, int zbeg, int zend, hipStream_t s, double *partials, int max_partials, int *nparts) {
    constexpr int VX = 16 / sizeof(T);
    const int w = (g->nx + 1) / (64 * VX);
    PJArgs<T> q; memset(&q, 0, sizeof(q));
    q.u = a.u; q.b = a.b; q.out = a.out;
    q.nx = g->nx; q.ny = g->ny; q.nz = g->nz; q.rs = g->pitch; q.ms = g->plane;
    q.a0 = a.a0; q.a1 = a.a1; q.a2 = a.a2; q.a3 = a.a3; q.a4 = a.a4; q.a5 = a.a5; q.a6 = a.a6; q.dinv = a.dinv; q.scale = a.scale;
    q.zbeg = zbeg; q.zend = zend;
    q.nty = (g->ny + 3) / 4;
    const int nzr = zend - zbeg;
    // 512-thread blocks: one per CU (256 tiles at 1023^3, one chunk); smaller blocks several per CU.  Small levels: short chunks
    const long target = (w > 4) ? 256 : (w > 2 ? 512 : 1024);
    long nch = (q.nty >= target) ? 1 : (target + q.nty - 1) / q.nty;
    if (g_zchunk > 0) nch = (nzr + g_zchunk - 1) / g_zchunk;
    else if (c->chunk_planes > 0 && nch < (nzr + c->chunk_planes - 1) / c->chunk_planes) nch = (nzr + c->chunk_planes - 1) / c->chunk_planes;
    int zc = (int)((nzr + nch - 1) / nch);
    if (zc < 4) zc = 4;
    if (zc > nzr) zc = nzr;
    long nblk = (long)q.nty * ((nzr + zc - 1) / zc);
    if (NORM && nblk > max_partials) {
        const long ntz = max_partials / q.nty;
        if (ntz < 1) return fail(MGK_EINVAL, "row sweep: partial buffer too small");
        zc = (int)((nzr + ntz - 1) / ntz);
        nblk = (long)q.nty * ((nzr + zc - 1) / zc);
    }
    q.zc = zc;
    if (nparts) *nparts = (int)nblk;
#define JROW_LAUNCH(WXV) hipLaunchKernelGGL((k_jrow<T, WXV, (sizeof(T) == 8 ? 2 : 1), NORM>), dim3((unsigned)nblk), dim3(64 * WXV), 0, s, q, partials)
    if (w <= 1) JROW_LAUNCH(1);
    else if (w <= 2) JROW_LAUNCH(2);
    else if (w <= 4) JROW_LAUNCH(4);
    else JROW_LAUNCH(8);
#undef JROW_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}

// closes the partial last coarse plane of a slab: bc(last) += sum_{di,dj} (1/4 w2[di][dj]) r(ghost plane), ascending (di, dj):
// the dk = 2 terms of the row of res, appended to the running sum exactly as the whole-grid kernel would.
template <typename T>
__global__ void __launch_bounds__(256) k_restrict_finish(XferArgs a, const T *rghost, T *bclast) {
    const int jc = blockIdx.x * blockDim.x + threadIdx.x;
    const int ic = blockIdx.y;
    if (jc >= a.nxc || ic >= a.nyc) return;
    const T w2[3][3] = {{(T)0.0625, (T)0.125, (T)0.0625}, {(T)0.125, (T)0.25, (T)0.125}, {(T)0.0625, (T)0.125, (T)0.0625}};
    const T wk = (T)0.25;
    T acc = bclast[(long)ic * a.pc + jc];
#pragma unroll
    for (int di = 0; di < 3; di++) {
        const T *row = rghost + (long)(2 * ic + di) * a.pf + 2 * jc;
#pragma unroll
        for (int dj = 0; dj < 3; dj++) acc += (wk * w2[di][dj]) * row[dj];
    }
    bclast[(long)ic * a.pc + jc] = acc;
}

// form of the fused kernels when the tuning knob leaves the choice open: the register / shuffle kernels with unconditional loads
// and DPP lane shifts wherever the shape allows them (measured on MI355X, LDS-tile form -> row form: fp64 fused prolongation sweep
// 1023^3 5.36 -> 4.45 ms, 511^3 0.77 -> 0.56, 255^3 0.117 -> 0.070; fused residual + restriction 1023^3 3.74-4.04 -> 3.35 ms,
// 255^3 0.086 -> 0.065; fp32 1023^3 2.88 -> 2.40 / 1.81 -> 1.80 ms, 511^3 0.44 -> 0.30 / 0.36 -> 0.25)
static int row_form_default(int w, size_t esz) { (void)w; (void)esz; return 34; }
// the row kernels load unconditionally: full rows (nx + 1 a whole number of wave rows, at most 8) and ny + 1 a multiple of 4
template <typename T>
static bool row_shape_ok(const mgk_geom *g) {
    constexpr int WR = 64 * (16 / (int)sizeof(T));
    const int w = (g->nx + 1) / WR;
    return g->dim == 3 && (g->nx + 1) % WR == 0 && (w == 1 || w == 2 || w == 4 || w == 8) && (g->ny + 1) % 4 == 0 && g->ny >= 3;
}
// tuning variants: 30 LDS-tile kernels; 31 row kernels; 33 / 34 / 35 experimental forms of the row kernels (see the launchers)
static int row_form(int w, size_t esz) {
    return (g_variant >= 31 && g_variant <= 35) ? g_variant : (g_variant == 30) ? 0 : row_form_default(w, esz);
}
// (measured on one box, LDS-tile kernel -> row form: fp64 1023^3 sweep 4.48 -> 4.30 ms, sweep+norm 4.48 -> 4.36; 511^3 0.534 -> 0.543,
//  255^3 0.064 -> 0.066; fp32 1023^3 2.1 -> 2.3: the row form is the default for fp64 rows of 1024 only, tuning variant 34 forces it)
template <typename T> static bool jrow_ok(const mgk_geom *g) {
    if (!row_shape_ok<T>(g) || g->nx < 127) return false;
    return g_variant == 34 || (g_variant < 0 && sizeof(T) == 8 && g->nx + 1 >= 1024);
}
template <typename T, int PD, int FORM>
static void launch_rrrow(int w, unsigned nblk, hipStream_t s, const RRArgs<T> &a) {
    if (w <= 1) hipLaunchKernelGGL((k_rrrow<T, 1, PD, FORM>), dim3(nblk), dim3(64), 0, s, a);
    else if (w <= 2) hipLaunchKernelGGL((k_rrrow<T, 2, PD, FORM>), dim3(nblk), dim3(128), 0, s, a);
    else if (w <= 4) hipLaunchKernelGGL((k_rrrow<T, 4, PD, FORM>), dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_rrrow<T, 8, PD, FORM>), dim3(nblk), dim3(512), 0, s, a);
}
template <typename T, int PD, int FORM>
static void launch_pjrow(int w, unsigned nblk, hipStream_t s, const PJArgs<T> &a) {
    if (w <= 1) hipLaunchKernelGGL((k_pjrow<T, 1, PD, FORM>), dim3(nblk), dim3(64), 0, s, a);
    else if (w <= 2) hipLaunchKernelGGL((k_pjrow<T, 2, PD, FORM>), dim3(nblk), dim3(128), 0, s, a);
    else if (w <= 4) hipLaunchKernelGGL((k_pjrow<T, 4, PD, FORM>), dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_pjrow<T, 8, PD, FORM>), dim3(nblk), dim3(512), 0, s, a);
}
template <typename T>
static int residual_restrict(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                             const T *b, const T *u, T *bc, T *uc0, double dinv_c, double scale_c, int kcbeg, int kcend, void *stream,
                             const T *far_hi = nullptr) {
    constexpr int VX = 16 / sizeof(T);
    if (!c || !gf || !gc || !coef || !b || !u || !bc || gf->dim != 3)
        return fail(MGK_EINVAL, "mgk_residual_restrict: bad arguments (3-D only)");
    XferArgs x;
    int rc = xfer_args(gf, gc, x);
    if (rc) return rc;
    if (gf->nz != 2 * gc->nz + 1 && gf->nz != 2 * gc->nz)
        return fail(MGK_EINVAL, "mgk_residual_restrict: nzf must be 2 nzc + 1 (whole grid, last slab) or 2 nzc (inner slab: partial last plane)");
    if (gf->nx + 1 > 1024) return fail(MGK_EINVAL, "mgk_residual_restrict: nx + 1 > 1024 is not built");
    RRArgs<T> a; memset(&a, 0, sizeof(a));
    a.u = u + gf->org; a.b = b + gf->org; a.bc = bc + gc->org;
    a.nx = gf->nx; a.ny = gf->ny; a.nz = gf->nz; a.nxc = gc->nx; a.nyc = gc->ny; a.nzc = gc->nz;
    a.rs = gf->pitch; a.ms = gf->plane; a.crs = gc->pitch; a.cms = gc->plane;
    a.a0 = (T)coef[0]; a.a1 = (T)coef[1]; a.a2 = (T)coef[2]; a.a3 = (T)coef[3]; a.a4 = (T)coef[4]; a.a5 = (T)coef[5]; a.a6 = (T)coef[6];
    if (uc0 && gf->nz != 2 * gc->nz + 1) return fail(MGK_EINVAL, "mgk_residual_restrict_jz: whole grids only");
    a.uc0 = uc0 ? uc0 + gc->org : nullptr; a.dinv_c = (T)dinv_c; a.scale_c = (T)scale_c;
    a.far_hi = (gf->nz == 2 * gc->nz) ? far_hi : nullptr;         // only an inner slab has a plane nz + 1 to look at
    a.nty = (gf->ny - 1 + 3) / 4;                 // tiles of 5 rows at stride 4; ny = 2 nyc + 1
    if (a.nty < 1) a.nty = 1;
    if (kcbeg < 0 || kcend > gc->nz || kcbeg >= kcend) return fail(MGK_EINVAL, "mgk_residual_restrict: empty or out-of-range coarse plane range");
    if (uc0 && (kcbeg != 0 || kcend != gc->nz)) return fail(MGK_EINVAL, "mgk_residual_restrict_jz: whole grids only");
    a.kcbeg = kcbeg; a.kcend = kcend;
    const int nkc = kcend - kcbeg;
    const int w = (gf->nx + 1 + 64 * VX - 1) / (64 * VX);       // waves needed for a full row
    // blocks of <= 256 threads (fp32 rows) leave room for two per CU: cut z in two (fp32 at 1023^3: 2.78 -> 1.82 ms)
    // (rows of <= 2 waves: 1024 blocks -- 255^3 inside the 511^3 cycle 73 -> 52 us)
    long nch = (a.nty >= 256) ? ((w <= 4) ? 2 : 1) : ((w <= 2 && sizeof(T) == 8 ? 1024 : 512) + a.nty - 1) / a.nty;
    if (g_zchunk > 0) nch = (nkc + g_zchunk - 1) / g_zchunk;
    else if (c->chunk_planes > 1 && nch < (nkc + c->chunk_planes / 2 - 1) / (c->chunk_planes / 2)) nch = (nkc + c->chunk_planes / 2 - 1) / (c->chunk_planes / 2);
    int kcc = (int)((nkc + nch - 1) / nch);
    if (kcc < 4) kcc = 4;
    if (kcc > nkc) kcc = nkc;
    a.kcc = kcc;
    const long ntz = (nkc + kcc - 1) / kcc;
    const unsigned nblk = (unsigned)(a.nty * ntz);
    hipStream_t s = S(c, stream);
    // register / shuffle form (k_rrrow): tuning variants 31 / 32 force it with prefetch distance 1 / 2, 30 forces the LDS-tile form
    const int rowpd = row_shape_ok<T>(gf) ? row_form(w, sizeof(T)) : 0;
    if (rowpd == 34 || rowpd == 35) launch_rrrow<T, 1, 2>(w, nblk, s, a);      // DPP lane shifts
    else if (rowpd) launch_rrrow<T, 1, 0>(w, nblk, s, a);
    else if (w <= 1) hipLaunchKernelGGL((k_resrestrict<T, 1>), dim3(nblk), dim3(64), 0, s, a);
    else if (w <= 2) hipLaunchKernelGGL((k_resrestrict<T, 2>), dim3(nblk), dim3(128), 0, s, a);
    else if (w <= 4) hipLaunchKernelGGL((k_resrestrict<T, 4>), dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_resrestrict<T, 8>), dim3(nblk), dim3(512), 0, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_residual_restrict_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                                         const double *b, const double *u, double *bc, void *stream) {
    if (!gc) return fail(MGK_EINVAL, "mgk_residual_restrict_f64: bad arguments");
    return residual_restrict<double>(c, gf, gc, coef, b, u, bc, nullptr, 0.0, 0.0, 0, gc->nz, stream);
}
extern "C" int mgk_residual_restrict_range_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                                               const double *b, const double *u, double *bc, int kcbeg, int kcend, void *stream) {
    return residual_restrict<double>(c, gf, gc, coef, b, u, bc, nullptr, 0.0, 0.0, kcbeg, kcend, stream);
}
extern "C" int mgk_residual_restrict_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                                         const float *b, const float *u, float *bc, void *stream) {
    if (!gc) return fail(MGK_EINVAL, "mgk_residual_restrict_f32: bad arguments");
    return residual_restrict<float>(c, gf, gc, coef, b, u, bc, nullptr, 0.0, 0.0, 0, gc->nz, stream);
}
extern "C" int mgk_residual_restrict_range_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef,
                                               const float *b, const float *u, float *bc, int kcbeg, int kcend, void *stream) {
    return residual_restrict<float>(c, gf, gc, coef, b, u, bc, nullptr, 0.0, 0.0, kcbeg, kcend, stream);
}
// The same on a z-slab whose `far` field (geometry gfar = (nx, ny, 2), see mgk_jacobi2_slab_*) holds, in its hi ghost plane, plane 1
// of the rank above, and whose u and b have valid hi ghost planes: an inner slab then evaluates the residual of plane nz itself
// (same operands, same arithmetic as the owner: same bits) and completes its last coarse plane -- no partial plane to finish.
template <typename T>
static int residual_restrict_slab(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *coef,
                                  const T *b, const T *u, const T *far, int has_hi, T *bc, int kcbeg, int kcend, void *stream) {
    if (!gf || !gc) return fail(MGK_EINVAL, "mgk_residual_restrict_slab: bad arguments");
    const T *hi = nullptr;
    if (has_hi) {
        if (!gfar || !far || gfar->dim != 3 || gfar->nz != 2 || gfar->nx != gf->nx || gfar->ny != gf->ny || gfar->pitch != gf->pitch)
            return fail(MGK_EINVAL, "mgk_residual_restrict_slab: the far-plane field must have the geometry (nx, ny, 2) of the slab");
        if (gf->nz != 2 * gc->nz) return fail(MGK_EINVAL, "mgk_residual_restrict_slab: a slab with a rank above has nzf = 2 nzc");
        hi = far + gfar->org + 2 * gfar->plane;
    }
    return residual_restrict<T>(c, gf, gc, coef, b, u, bc, nullptr, 0.0, 0.0, kcbeg, kcend, stream, hi);
}
extern "C" int mgk_residual_restrict_slab_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *coef,
                                              const double *b, const double *u, const double *far, int has_hi, double *bc,
                                              int kcbeg, int kcend, void *stream) {
    return residual_restrict_slab<double>(c, gf, gc, gfar, coef, b, u, far, has_hi, bc, kcbeg, kcend, stream);
}
extern "C" int mgk_residual_restrict_slab_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const mgk_geom *gfar, const double *coef,
                                              const float *b, const float *u, const float *far, int has_hi, float *bc,
                                              int kcbeg, int kcend, void *stream) {
    return residual_restrict_slab<float>(c, gf, gc, gfar, coef, b, u, far, has_hi, bc, kcbeg, kcend, stream);
}
// the same, also writing the coarse level's first sweep from a zero guess (uc0 = scale_c * (bc * dinv_c), what mgk_jacobi_zero_*
// would compute from bc): saves that kernel's read of bc
extern "C" int mgk_residual_restrict_jz_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const double *b,
                                            const double *u, double *bc, double *uc0, double dinv_c, double scale_c, void *stream) {
    if (!uc0 || !gc) return fail(MGK_EINVAL, "mgk_residual_restrict_jz_f64: null uc0");
    return residual_restrict<double>(c, gf, gc, coef, b, u, bc, uc0, dinv_c, scale_c, 0, gc->nz, stream);
}
extern "C" int mgk_residual_restrict_jz_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const float *b,
                                            const float *u, float *bc, float *uc0, double dinv_c, double scale_c, void *stream) {
    if (!uc0 || !gc) return fail(MGK_EINVAL, "mgk_residual_restrict_jz_f32: null uc0");
    return residual_restrict<float>(c, gf, gc, coef, b, u, bc, uc0, dinv_c, scale_c, 0, gc->nz, stream);
}
template <typename T>
static int restrict_finish(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const T *r, T *bc, void *stream) {
    if (!c || !gf || !gc || !r || !bc || gf->dim != 3 || gf->nz != 2 * gc->nz)
        return fail(MGK_EINVAL, "mgk_restrict_finish: inner slabs only (nzf = 2 nzc)");
    XferArgs x;
    int rc = xfer_args(gf, gc, x);
    if (rc) return rc;
    dim3 block(256), grid((gc->nx + 255) / 256, gc->ny);
    // r's hi ghost plane (index nz) holds the neighbour's first residual plane; the last coarse plane is nzc - 1
    hipLaunchKernelGGL((k_restrict_finish<T>), grid, block, 0, S(c, stream), x, r + gf->org + (long)gf->nz * gf->plane,
                       bc + gc->org + (long)(gc->nz - 1) * gc->plane);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_restrict_finish_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *r, double *bc, void *stream) {
    return restrict_finish<double>(c, gf, gc, r, bc, stream);
}
extern "C" int mgk_restrict_finish_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const float *r, float *bc, void *stream) {
    return restrict_finish<float>(c, gf, gc, r, bc, stream);
}

// 2-D form of the fused prolongation sweep as independent waves (round 2; the LDS-tile k_stencil<..MODE_PJACOBI> ran at 4.0 TB/s at
// 4095^2): lane l holds the column pair x0 = 2 (62 tx + l - 1) and its parent column pair (l - 1 and l of the coarse row, the left one by
// a DPP shift); the corrected rows y-1, y, y+1 live in registers (the correction is added when a row arrives), lanes 1 .. 62 sweep and
// store: tiles overlap by two lanes.  Marches along y; two coarse rows are live (an even fine row has two parent rows).
struct PJ2dArgs {
    const double *u, *b, *uc;
    double *out;
    int nx, ny, nxc, nyc;
    long rs, crs;
    int ntx, yc;
    double a0, a2, a3, a4, a6, dinv, scale;
    const double *ctab, *dtab;      // optional (stretched meshes): 5 coefficients and 1/diag per grid row
};
__global__ void __launch_bounds__(256) k_pj2d(const PJ2dArgs a) {
    using VT = V16<double>;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // (row indices and row addresses on the scalar unit)
    const int tx = wid % a.ntx, cy = wid / a.ntx;
    const int y0 = cy * a.yc, y1 = min(y0 + a.yc, a.ny);
    if (y0 >= y1) return;                                     // whole wave
    const int pidx = tx * 62 + lane - 1;                      // pair index = coarse column of the pair's second (odd) fine column
    const int x0 = 2 * pidx;
    const bool xin = (x0 >= 0 && x0 < a.nx);
    const bool lastvec = (x0 + 2 > a.nx);
    const bool store = (lane >= 1 && lane <= 62 && xin);
    const int xc = min(max(x0, 0), a.nx - 1);
    const double *up_ = a.u + xc, *bp_ = a.b + xc;
    const double *cp_ = a.uc + min(max(pidx, -1), a.nxc);     // coarse column pidx (-1 and nxc are the ghost columns: zero)
    // coarse row ic of this lane's column and of the column to its left (rows -1 and nyc are ghost rows: zero)
    auto ldc = [&](int ic) -> double { return cp_[(long)min(max(ic, -1), a.nyc) * a.crs]; };
    // u + P uc on row y: y odd has ONE parent row (y-1)/2, y even two, y/2 - 1 and y/2; the even column x0 has the parent columns
    // pidx-1, pidx (weight 1/2 each), the odd column x0+1 the parent column pidx.  Terms in the order of the prolongation's row
    // (ascending coarse row, then column), weights w = wi * wj as there
    auto ldraw = [&](int y) -> VT { return *reinterpret_cast<const VT *>(up_ + (long)min(max(y, -1), a.ny) * a.rs); };
    auto correct = [&](VT v, int y, double cA, double cB) -> VT {      // cA: coarse row of the first parent, cB: of the second (even y)
        const bool two = ((y & 1) == 0);
        const double wi = two ? 0.5 : 1.0;
        const double wh = wi * 0.5, w1 = wi * 1.0;
        const double mA = lane_up<true>(cA), mB = lane_up<true>(cB);
        double s0 = 0.0, s1 = 0.0;
        s0 += wh * mA; s0 += wh * cA; s1 += w1 * cA;
        if (two) { s0 += wh * mB; s0 += wh * cB; s1 += w1 * cB; }
        v.v[0] = v.v[0] + s0; v.v[1] = v.v[1] + s1;
        if (!xin || y < 0 || y >= a.ny) { v.v[0] = 0.0; v.v[1] = 0.0; }
        if (lastvec) v.v[1] = 0.0;
        return v;
    };
    // parent rows of fine row y: (y-1)/2 alone (odd y; returned twice) or y/2 - 1 and y/2 (even y)
    auto pA = [&](int y) { return (y & 1) ? (y - 1) >> 1 : (y >> 1) - 1; };
    auto pB = [&](int y) { return (y & 1) ? (y - 1) >> 1 : (y >> 1); };
    // rows y0-1, y0, y0+1 corrected; row y0+2 and its parents in flight
    VT ua = correct(ldraw(y0 - 1), y0 - 1, ldc(pA(y0 - 1)), ldc(pB(y0 - 1)));
    VT ub = correct(ldraw(y0), y0, ldc(pA(y0)), ldc(pB(y0)));
    VT uc = correct(ldraw(y0 + 1), y0 + 1, ldc(pA(y0 + 1)), ldc(pB(y0 + 1)));
    VT ur = ldraw(y0 + 2);
    double cA = ldc(pA(y0 + 2)), cB = ldc(pB(y0 + 2));
    VT b0 = ldv_stream(bp_ + (long)y0 * a.rs, true);
    for (int y = y0; y < y1; y++) {
        // loads consumed in the next step: row y+3 with its parents, b of row y+1
        const VT ur2 = ldraw(y + 3);
        const double cA2 = ldc(pA(y + 3)), cB2 = ldc(pB(y + 3));
        const VT bn = ldv_stream(bp_ + (long)min(y + 1, a.ny - 1) * a.rs, true);
        // the row that arrived during the last step gets its correction
        const VT un = correct(ur, y + 2, cA, cB);
        const double Wv = lane_up<true>(ub.v[1]), Ev = lane_dn<true>(ub.v[0]);
        double k0 = a.a0, k2 = a.a2, k3 = a.a3, k4 = a.a4, k6 = a.a6, kd = a.dinv;
        if (a.ctab) { const CDBL4 *cr = (const CDBL4 *)(a.ctab + 5 * (long)y); k0 = cr[0]; k2 = cr[1]; k3 = cr[2]; k4 = cr[3]; k6 = cr[4]; kd = ((const CDBL4 *)(a.dtab + y))[0]; }
        VT o;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double wv = (e == 0) ? Wv : ub.v[0];
            const double ev = (e == 1) ? Ev : ub.v[1];
            double s = k0 * ua.v[e];
            s = s + k2 * wv;
            s = s + k3 * ub.v[e];
            s = s + k4 * ev;
            s = s + k6 * uc.v[e];
            const double res = (xin ? b0.v[e] : 0.0) - s;
            const double zz = res * kd;
            o.v[e] = ub.v[e] + a.scale * zz;
        }
        if (lastvec) o.v[1] = 0.0;
        if (store) stv_policy(a.out + (long)y * a.rs + x0, o, mgk_store_nt_2d(a.ny, a.rs));
        ua = ub; ub = uc; uc = un; ur = ur2; cA = cA2; cB = cB2; b0 = bn;
    }
}
static int prolong_jacobi_2d_waves(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                   const double *b, const double *ucoarse, const double *u, double *unew, void *stream,
                                   const double *ctab = nullptr, const double *dtab = nullptr) {
    PJ2dArgs a; memset(&a, 0, sizeof(a));
    a.ctab = ctab; a.dtab = dtab;
    a.u = u + gf->org; a.b = b + gf->org; a.uc = ucoarse + gc->org; a.out = unew + gf->org;
    a.nx = gf->nx; a.ny = gf->ny; a.nxc = gc->nx; a.nyc = gc->ny; a.rs = gf->pitch; a.crs = gc->pitch;
    if (coef) { a.a0 = coef[0]; a.a2 = coef[1]; a.a3 = coef[2]; a.a4 = coef[3]; a.a6 = coef[4]; }
    a.dinv = dinv; a.scale = scale;
    a.ntx = (gc->nx + 1 + 61) / 62;                           // pairs 0 .. nxc
    long nch = (4096 + a.ntx - 1) / a.ntx;                    // ~4096 waves (16 per CU); every chunk re-reads two fine rows
    if (g_zchunk > 0) nch = (gf->ny + g_zchunk - 1) / g_zchunk;
    int yc = (int)((gf->ny + nch - 1) / nch);
    if (yc < 16 && g_zchunk <= 0) yc = 16;
    if (yc > gf->ny) yc = gf->ny;
    a.yc = yc;
    const long waves = (long)a.ntx * ((gf->ny + yc - 1) / yc);
    hipLaunchKernelGGL(k_pj2d, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}

// K4 fused into the first post-smoothing sweep: unew = J(u + P uc)  (src/solver.c:1540-1542)
template <typename T>
static int prolong_jacobi(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                          const T *b, const T *ucoarse, const T *u, T *unew, int zbeg, int zend, void *stream) {
    if (!c || !gf || !gc || !coef || !b || !ucoarse || !u || !unew || u == unew || (gf->dim != 3 && sizeof(T) != 8))
        return fail(MGK_EINVAL, "mgk_prolong_jacobi: bad arguments (fp32: 3-D only)");
    XferArgs x;
    int rc = xfer_args(gf, gc, x);
    if (rc) return rc;
    StArgs<T> a; memset(&a, 0, sizeof(a));
    a.u = u + gf->org; a.b = b + gf->org; a.out = unew + gf->org;
    a.uc = ucoarse + gc->org; a.crs = gc->pitch; a.cms = (gf->dim == 3) ? gc->plane : gc->pitch;
    a.nxc = gc->nx; a.nyc = gc->ny; a.nzc = gc->nz;
    set_coef(a, gf, coef); a.dinv = (T)dinv; a.scale = (T)scale;
    const int nm = (gf->dim == 3) ? gf->nz : gf->ny;
    if (zbeg < 0 || zend > nm || zbeg >= zend) return fail(MGK_EINVAL, "mgk_prolong_jacobi: empty or out-of-range plane range");
    a.zbeg = zbeg; a.zend = zend;
    if constexpr (sizeof(T) == 8) {
        // 2-D, whole grid: independent waves from 2047^2 on (a wave marches >= 16 rows: on the small levels of the 4097^2 cycle this form
        // takes 14-18 us where the LDS-tile kernel takes 5-8); tuning variant 30 keeps the LDS-tile kernel, 38 forces the waves
        if (gf->dim == 2 && zbeg == 0 && zend == gf->ny && ((gf->nx >= 2047 && g_variant != 30) || (g_variant == 38 && gf->nx >= 3)))
            return prolong_jacobi_2d_waves(c, gf, gc, coef, dinv, scale, (const double *)b, (const double *)ucoarse, (const double *)u, (double *)unew, stream);
    }
    constexpr int VX = 16 / sizeof(T);
    const int w = (gf->nx + 1 + 64 * VX - 1) / (64 * VX);        // waves per full row
    const int rowpd = row_shape_ok<T>(gf) ? row_form(w, sizeof(T)) : 0;
    if (rowpd) {
        // register / shuffle form (k_pjrow): full-row tiles of 4 rows; one 512-thread block per CU at 1023^3 (256 tiles, one
        // chunk), smaller blocks two per CU
        PJArgs<T> q; memset(&q, 0, sizeof(q));
        q.u = a.u; q.b = a.b; q.uc = a.uc; q.out = a.out;
        q.nx = gf->nx; q.ny = gf->ny; q.nz = gf->nz; q.nxc = gc->nx; q.nyc = gc->ny; q.nzc = gc->nz;
        q.rs = gf->pitch; q.ms = gf->plane; q.crs = gc->pitch; q.cms = gc->plane;
        q.a0 = a.a0; q.a1 = a.a1; q.a2 = a.a2; q.a3 = a.a3; q.a4 = a.a4; q.a5 = a.a5; q.a6 = a.a6; q.dinv = a.dinv; q.scale = a.scale;
        q.zbeg = zbeg; q.zend = zend;
        q.nty = (gf->ny + 3) / 4;
        const int nzr = zend - zbeg;
        const long target = (w > 4) ? 256 : (w <= 2 && sizeof(T) == 8 ? 1024 : 512);       // (255^3 inside the 511^3 cycle: 76 -> 67 us)
        long nch = (q.nty >= target) ? 1 : (target + q.nty - 1) / q.nty;
        if (g_zchunk > 0) nch = (nzr + g_zchunk - 1) / g_zchunk;
        else if (c->chunk_planes > 0 && nch < (nzr + c->chunk_planes - 1) / c->chunk_planes) nch = (nzr + c->chunk_planes - 1) / c->chunk_planes;
        int zc = (int)((nzr + nch - 1) / nch);
        if (zc < 4) zc = 4;
        if (zc > nzr) zc = nzr;
        q.zc = zc;
        const unsigned nblk = (unsigned)(q.nty * ((nzr + zc - 1) / zc));
        if (rowpd == 33) launch_pjrow<T, 1, 1>(w, nblk, S(c, stream), q);          // unconditional loads
        else if (rowpd == 34) launch_pjrow<T, 1, 3>(w, nblk, S(c, stream), q);     // + DPP lane shifts
        else if (rowpd == 35) launch_pjrow<T, 1, 2>(w, nblk, S(c, stream), q);     // predicated loads, DPP lane shifts
        else launch_pjrow<T, 1, 0>(w, nblk, S(c, stream), q);
        HIPCHK(hipGetLastError());
        return 0;
    }
    return dispatch_st<MODE_PJACOBI>(c, gf, a, S(c, stream), nullptr);
}
extern "C" int mgk_prolong_jacobi_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv,
                                      double scale, const double *b, const double *uc, const double *u, double *unew, void *stream) {
    if (!gf) return fail(MGK_EINVAL, "mgk_prolong_jacobi_f64: bad arguments");
    return prolong_jacobi<double>(c, gf, gc, coef, dinv, scale, b, uc, u, unew, 0, gf->dim == 3 ? gf->nz : gf->ny, stream);
}
extern "C" int mgk_prolong_jacobi_range_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv,
                                            double scale, const double *b, const double *uc, const double *u, double *unew,
                                            int zbeg, int zend, void *stream) {
    return prolong_jacobi<double>(c, gf, gc, coef, dinv, scale, b, uc, u, unew, zbeg, zend, stream);
}
extern "C" int mgk_prolong_jacobi_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv,
                                      double scale, const float *b, const float *uc, const float *u, float *unew, void *stream) {
    if (!gf) return fail(MGK_EINVAL, "mgk_prolong_jacobi_f32: bad arguments");
    return prolong_jacobi<float>(c, gf, gc, coef, dinv, scale, b, uc, u, unew, 0, gf->nz, stream);
}
extern "C" int mgk_prolong_jacobi_range_f32(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv,
                                            double scale, const float *b, const float *uc, const float *u, float *unew,
                                            int zbeg, int zend, void *stream) {
    return prolong_jacobi<float>(c, gf, gc, coef, dinv, scale, b, uc, u, unew, zbeg, zend, stream);
}




// ------------------------------------------------------------------------------------------
// The tail of the hierarchy in ONE kernel: the levels with n <= 15 (3-D) / n <= 63 (2-D) hold at most a few thousand
// unknowns each; their part of a V-cycle is ~50 launches of ~4.5 us of which the arithmetic is a rounding error.  One
// 1024-lane workgroup keeps u, its ping-pong partner and b of EVERY tail level in LDS (ghost rings included: 139 KB in 3-D,
// 137 KB in 2-D) and walks the reference's loop (src/solver.c:1533-1544) over them with a workgroup barrier between
// operations.  Each operation evaluates exactly the expressions of the kernels it replaces (k_stencil / k_jacobi_zero /
// k_restrict / k_prolong_add: same terms, same order, no FMA), so the cycle stays bit-identical.
// In: b of the first tail level (global, padded layout; produced by the restriction from the level above).
// Out: u of that level after its post-smoothing (global), ready for the prolongation to the level above.
// ------------------------------------------------------------------------------------------
#define MGK_TAIL_MAXLEV 8
#define MGK_TAIL_LDS_BYTES 143360          /* 140 KiB of the 160 KiB */
template <typename T>
struct TailArgs {
    int dim, nlev, v0, v1;
    int n[MGK_TAIL_MAXLEV];
    int off[MGK_TAIL_MAXLEV];              // offset (elements) of the level's three arrays in LDS: A0, A1 (u ping-pong), B; each (n+2)^dim
    T coef[MGK_TAIL_MAXLEV][7], dinv[MGK_TAIL_MAXLEV];
    T scale, cscale;                       // the Richardson damping factor; cscale: on the COARSEST level (PCMG's exact solve of a 1 x 1 system is one undamped sweep)
    const T *b_in;
    T *u_out;                              // both at the interior origin of the first tail level's padded field
    long rs, ms;                           // its row / plane strides
    int total;                             // elements of LDS in use
    const T *ctab[MGK_TAIL_MAXLEV], *dtab[MGK_TAIL_MAXLEV];   // optional (2-D stretched meshes): per-row coefficients / 1/diag of every tail level
    long long *stamps;                     // profiling aid (mgk_debug_tail_stamps; null in production): lane 0 deposits s_memrealtime / s_memtime pairs at every barrier
};

template <typename T, int DIM>
__device__ __forceinline__ int tail_idx(int m, int k, int i, int j) {      // (k,i,j) interior coordinates -> index in an (n+2)^DIM array
    return (DIM == 3) ? ((k + 1) * m + (i + 1)) * m + (j + 1) : (i + 1) * m + (j + 1);
}
// Every interior point (k, i, j) of an n^DIM level once, dealt over the lanes of the workgroup.  With n + 1 a power of two (every level of
// a hierarchy with npts = 2^k + 1) the coordinates are shifts and masks of the lane number instead of two divisions and two remainders
// by n (~70 of the ~100 vector instructions per point; round 3, under rocprofv3: k_tail<double, 3> 43 -> 33 us, <double, 2> 45 -> 42 us).
// Which lane evaluates a point does not change its value.  What bounds the kernel (tools/tail_phases.py, stamps at every barrier): a step on
// a tiny level costs 0.36 us (850 clocks at 2.4 GHz: scalar loads of the level's constants, LDS round trip, the dependent fp64 chain, the
// barrier), a sweep of the largest level 1.7-2.3 us -- the instruction and LDS throughput of ONE CU for 4 points per lane (evaluating the
// four points of a lane before storing any of them, so that their latencies overlap, changed nothing: it is throughput, not latency).
template <int DIM, typename F>
__device__ __forceinline__ void tail_points(int n, F &&f) {
    const int t = threadIdx.x, B = blockDim.x;
    const int lg = 31 - __clz(n + 1);
    if (((n + 1) & n) == 0 && (B >> (DIM == 3 ? 2 * lg : lg)) >= 1) {
        const int j = t & n;
        if (DIM == 2) {
            const int step = B >> lg;
            if (j < n) for (int i = t >> lg; i < n; i += step) f(0, i, j);
        } else {
            const int i = (t >> lg) & n, step = B >> (2 * lg);
            if (j < n && i < n) for (int k = t >> (2 * lg); k < n; k += step) f(k, i, j);
        }
    } else {
        const int N = (DIM == 3) ? n * n * n : n * n;
        for (int p = t; p < N; p += B) f((DIM == 3) ? p / (n * n) : 0, (p / n) % n, p % n);
    }
}
// mode 0: out = u + scale*((b - A u)*dinv); mode 1: out = b - A u; mode 2 (zero guess): out = scale*(b*dinv)
template <typename T, int DIM>
__device__ void tail_stencil(int mode, int n, const T *cf_, T dinv_, T scale, const T *u, const T *b, T *out, const T *ctab = nullptr, const T *dtab = nullptr) {
    const int m = n + 2, sk = m * m;
    tail_points<DIM>(n, [&](int k, int i, int j) {
        const int q = tail_idx<T, DIM>(m, k, i, j);
        const T *cf = (DIM == 2 && ctab) ? ctab + 5 * i : cf_;
        const T dinv = (DIM == 2 && dtab) ? dtab[i] : dinv_;
        if (mode == 2) { const T zx = b[q] * dinv; out[q] = scale * zx; return; }
        T t;
        if (DIM == 3) {
            t = cf[0] * u[q - sk];
            t = t + cf[1] * u[q - m];
            t = t + cf[2] * u[q - 1];
            t = t + cf[3] * u[q];
            t = t + cf[4] * u[q + 1];
            t = t + cf[5] * u[q + m];
            t = t + cf[6] * u[q + sk];
        } else {
            t = cf[0] * u[q - m];
            t = t + cf[1] * u[q - 1];
            t = t + cf[2] * u[q];
            t = t + cf[3] * u[q + 1];
            t = t + cf[4] * u[q + m];
        }
        const T res = b[q] - t;
        if (mode == 1) { out[q] = res; return; }
        const T zz = res * dinv;
        out[q] = u[q] + scale * zz;
    });
}
// bc = R r (k_restrict: ascending fine index (dk, di, dj), weights w1[dk]*w2[di][dj])
template <typename T, int DIM>
__device__ void tail_restrict(int nf, int nc, const T *r, T *bc) {
    const int mf = nf + 2, mc = nc + 2;
    const T w2[3][3] = {{(T)0.0625, (T)0.125, (T)0.0625}, {(T)0.125, (T)0.25, (T)0.125}, {(T)0.0625, (T)0.125, (T)0.0625}};
    const T w1[3] = {(T)0.25, (T)0.5, (T)0.25};
    tail_points<DIM>(nc, [&](int kc, int ic, int jc) {
        T sum = (T)0;
        if (DIM == 3) {
#pragma unroll
            for (int dk = 0; dk < 3; dk++)
#pragma unroll
                for (int di = 0; di < 3; di++) {
                    const T *row = r + tail_idx<T, DIM>(mf, 2 * kc + dk, 2 * ic + di, 2 * jc);
                    sum += (w1[dk] * w2[di][0]) * row[0];
                    sum += (w1[dk] * w2[di][1]) * row[1];
                    sum += (w1[dk] * w2[di][2]) * row[2];
                }
        } else {
#pragma unroll
            for (int di = 0; di < 3; di++) {
                const T *row = r + tail_idx<T, DIM>(mf, 0, 2 * ic + di, 2 * jc);
                sum += w2[di][0] * row[0];
                sum += w2[di][1] * row[1];
                sum += w2[di][2] * row[2];
            }
        }
        bc[tail_idx<T, DIM>(mc, kc, ic, jc)] = sum;
    });
}
// uf += P uc (k_prolong_add / prolong_one: parents in ascending coarse index, weight wk*(wi*wj) each)
template <typename T, int DIM>
__device__ void tail_prolong_add(int nf, int nc, const T *uc, T *uf) {
    const int mf = nf + 2, mc = nc + 2;
    tail_points<DIM>(nf, [&](int k3, int i, int x) {
        const int k = (DIM == 3) ? k3 : 1;
        const int iodd = i & 1, kodd = k & 1, xodd = x & 1;
        const int ic0 = iodd ? (i - 1) / 2 : i / 2 - 1, nic = iodd ? 1 : 2;
        const int kc0 = (DIM == 3) ? (kodd ? (k - 1) / 2 : k / 2 - 1) : 0, nkc = (DIM == 3) ? (kodd ? 1 : 2) : 1;
        const int jc0 = xodd ? (x - 1) / 2 : x / 2 - 1, njc = xodd ? 1 : 2;
        const T wi = iodd ? (T)1 : (T)0.5, wk = (DIM == 3) ? (kodd ? (T)1 : (T)0.5) : (T)1, wj = xodd ? (T)1 : (T)0.5;
        const T w = (DIM == 3) ? wk * (wi * wj) : wi * wj;
        T s = (T)0;
        for (int qk = 0; qk < nkc; qk++)
            for (int qi = 0; qi < nic; qi++)
                for (int qj = 0; qj < njc; qj++) s += w * uc[tail_idx<T, DIM>(mc, kc0 + qk, ic0 + qi, jc0 + qj)];
        const int q = tail_idx<T, DIM>(mf, (DIM == 3) ? k : 0, i, x);
        uf[q] = uf[q] + s;
    });
}

template <typename T, int DIM>
__global__ void __launch_bounds__(1024) k_tail(const TailArgs<T> a) {
    __shared__ __attribute__((aligned(16))) unsigned char raw[MGK_TAIL_LDS_BYTES];
    T *lds = reinterpret_cast<T *>(raw);
    int nstamp = 0;
    auto stamp = [&]() { if (a.stamps && threadIdx.x == 0 && nstamp < 128) { a.stamps[2 * nstamp] = wall_clock64(); a.stamps[2 * nstamp + 1] = clock64(); nstamp++; } };
    stamp();
    for (int q = threadIdx.x; q < a.total; q += blockDim.x) lds[q] = (T)0;        // ghost rings stay 0 (homogeneous Dirichlet)
    __syncthreads();
    stamp();
    int cur[MGK_TAIL_MAXLEV];                     // which of A0 / A1 holds u of the level
    auto A = [&](int l, int which) -> T * { const int m = a.n[l] + 2; const int sz = (DIM == 3) ? m * m * m : m * m; return lds + a.off[l] + which * sz; };
    auto Bv = [&](int l) -> T * { return A(l, 2); };
    {   // b of the first tail level from global memory
        const int n = a.n[0], m = n + 2, N = (DIM == 3) ? n * n * n : n * n;
        T *b0 = Bv(0);
        for (int p = threadIdx.x; p < N; p += blockDim.x) {
            const int j = p % n, i = (p / n) % n, k = (DIM == 3) ? p / (n * n) : 0;
            b0[tail_idx<T, DIM>(m, k, i, j)] = a.b_in[(long)k * a.ms + (long)i * a.rs + j];
        }
    }
    __syncthreads(); stamp();
    // KSPSolve from a zero guess, `sweeps` Richardson+Jacobi sweeps (src/solver.c:1536): the first is scale*(b*dinv)
    auto smooth0 = [&](int l, int sweeps) {
        cur[l] = 0;
        if (sweeps < 1) return;                   // KSPSolve zero-fills: A0 is still all zeros
        const T sc = (l == a.nlev - 1) ? a.cscale : a.scale;
        tail_stencil<T, DIM>(2, a.n[l], a.coef[l], a.dinv[l], sc, A(l, 0), Bv(l), A(l, 0), a.ctab[l], a.dtab[l]);
        __syncthreads(); stamp();
        for (int it = 1; it < sweeps; it++) {
            tail_stencil<T, DIM>(0, a.n[l], a.coef[l], a.dinv[l], sc, A(l, cur[l]), Bv(l), A(l, cur[l] ^ 1), a.ctab[l], a.dtab[l]);
            __syncthreads(); stamp();
            cur[l] ^= 1;
        }
    };
    smooth0(0, a.nlev == 1 ? a.v1 : a.v0);
    for (int l = 1; l < a.nlev; l++) {            // :1534-1537
        tail_stencil<T, DIM>(1, a.n[l - 1], a.coef[l - 1], a.dinv[l - 1], a.scale, A(l - 1, cur[l - 1]), Bv(l - 1), A(l - 1, cur[l - 1] ^ 1), a.ctab[l - 1], a.dtab[l - 1]);
        __syncthreads(); stamp();
        tail_restrict<T, DIM>(a.n[l - 1], a.n[l], A(l - 1, cur[l - 1] ^ 1), Bv(l));
        __syncthreads(); stamp();
        // the zero-guess sweep writes A0 of level l, which is all zeros only in the first cycle... it is overwritten in full
        smooth0(l, l == a.nlev - 1 ? a.v1 : a.v0);
    }
    for (int l = a.nlev - 2; l >= 0; l--) {       // :1540-1542
        tail_prolong_add<T, DIM>(a.n[l], a.n[l + 1], A(l + 1, cur[l + 1]), A(l, cur[l]));
        __syncthreads(); stamp();
        for (int it = 0; it < a.v0; it++) {
            tail_stencil<T, DIM>(0, a.n[l], a.coef[l], a.dinv[l], a.scale, A(l, cur[l]), Bv(l), A(l, cur[l] ^ 1), a.ctab[l], a.dtab[l]);
            __syncthreads(); stamp();
            cur[l] ^= 1;
        }
    }
    {
        const int n = a.n[0], m = n + 2, N = (DIM == 3) ? n * n * n : n * n;
        const T *u0 = A(0, cur[0]);
        for (int p = threadIdx.x; p < N; p += blockDim.x) {
            const int j = p % n, i = (p / n) % n, k = (DIM == 3) ? p / (n * n) : 0;
            a.u_out[(long)k * a.ms + (long)i * a.rs + j] = u0[tail_idx<T, DIM>(m, k, i, j)];
        }
    }
    stamp();
    if (a.stamps && threadIdx.x == 0) a.stamps[256] = nstamp;
}

static thread_local long long *g_tail_stamps = nullptr;
// profiling aid: the tail kernels launched by this thread deposit (s_memrealtime, s_memtime) at each of their barriers into dev[0 .. 255]
// and the number of pairs into dev[256] (257 long longs of device memory; null switches it off)
extern "C" void mgk_debug_tail_stamps(long long *dev) { g_tail_stamps = dev; }
template <typename T>
static int tail_cycle(mgk_ctx *c, const mgk_geom *g0, int nlev, const int *n, const double *coef7, const double *dinv, double scale,
                      int v0, int v1, const T *b, T *u, void *stream, const T *const *ctab = nullptr, const T *const *dtab = nullptr, const double *cscale = nullptr) {
    if (!c || !g0 || !n || (!coef7 && !ctab) || (!dinv && !dtab) || !b || !u || nlev < 1 || nlev > MGK_TAIL_MAXLEV || v0 < 0 || v1 < 0 ||
        (ctab && (!dtab || g0->dim != 2)))
        return fail(MGK_EINVAL, "mgk_tail_cycle: bad arguments");
    if (n[0] != g0->nx || g0->ny != g0->nx || (g0->dim == 3 && g0->nz != g0->nx))
        return fail(MGK_EINVAL, "mgk_tail_cycle: the first tail level must be a whole cube / square of n[0] unknowns per side");
    TailArgs<T> a; memset(&a, 0, sizeof(a));
    a.dim = g0->dim; a.nlev = nlev; a.v0 = v0; a.v1 = v1; a.scale = (T)scale; a.cscale = (T)(cscale ? *cscale : scale);
    long off = 0;
    for (int l = 0; l < nlev; l++) {
        if (n[l] < 1 || (l > 0 && n[l - 1] != 2 * n[l] + 1)) return fail(MGK_EINVAL, "mgk_tail_cycle: need n[l-1] = 2 n[l] + 1");
        const long m = n[l] + 2, sz = (g0->dim == 3) ? m * m * m : m * m;
        a.n[l] = n[l]; a.off[l] = (int)off; off += 3 * sz;
        for (int q = 0; q < 7; q++) a.coef[l][q] = coef7 ? (T)coef7[7 * l + q] : (T)0;
        a.dinv[l] = dinv ? (T)dinv[l] : (T)1;
        a.ctab[l] = ctab ? ctab[l] : nullptr; a.dtab[l] = dtab ? dtab[l] : nullptr;
        if (ctab && (!ctab[l] || !dtab[l])) return fail(MGK_EINVAL, "mgk_tail_cycle: null coefficient table");
    }
    if (off * (long)sizeof(T) > MGK_TAIL_LDS_BYTES) return fail(MGK_EINVAL, "mgk_tail_cycle: the levels do not fit in LDS");
    a.total = (int)off;
    a.b_in = b + g0->org; a.u_out = u + g0->org; a.rs = g0->pitch; a.ms = g0->plane;
    a.stamps = g_tail_stamps;
    if (g0->dim == 3) hipLaunchKernelGGL((k_tail<T, 3>), dim3(1), dim3(1024), 0, S(c, stream), a);
    else hipLaunchKernelGGL((k_tail<T, 2>), dim3(1), dim3(1024), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_tail_cycle_f64(mgk_ctx *c, const mgk_geom *g0, int nlev, const int *n, const double *coef7, const double *dinv,
                                  double scale, int v0, int v1, const double *b, double *u, void *stream) {
    return tail_cycle<double>(c, g0, nlev, n, coef7, dinv, scale, v0, v1, b, u, stream);
}
extern "C" int mgk_tail_cycle_f32(mgk_ctx *c, const mgk_geom *g0, int nlev, const int *n, const double *coef7, const double *dinv,
                                  double scale, int v0, int v1, const float *b, float *u, void *stream) {
    return tail_cycle<float>(c, g0, nlev, n, coef7, dinv, scale, v0, v1, b, u, stream);
}
// 2-D stretched meshes: ctab[l] / dtab[l] are the device tables (n[l] x 5 and n[l] doubles) of tail level l
extern "C" int mgk_tail_cycle_rowcoef_f64(mgk_ctx *c, const mgk_geom *g0, int nlev, const int *n, const double *const *ctab, const double *const *dtab,
                                          double scale, int v0, int v1, const double *b, double *u, void *stream) {
    if (!ctab || !dtab) return fail(MGK_EINVAL, "mgk_tail_cycle_rowcoef_f64: null tables");
    return tail_cycle<double>(c, g0, nlev, n, nullptr, nullptr, scale, v0, v1, b, u, stream, ctab, dtab);
}
// ... with another damping factor on the COARSEST level (2-D, fp64; constant coefficients or row tables: pass either coef7 + dinv or ctab + dtab):
// PCMG's default coarse solver is exact, and on a 1 x 1 grid the exact solve IS one undamped Jacobi sweep from the zero guess (v1 = 1, coarse_scale = 1)
extern "C" int mgk_tail_cycle_cs_f64(mgk_ctx *c, const mgk_geom *g0, int nlev, const int *n, const double *coef7, const double *dinv,
                                     const double *const *ctab, const double *const *dtab, double scale, double coarse_scale, int v0, int v1,
                                     const double *b, double *u, void *stream) {
    if ((ctab == nullptr) != (dtab == nullptr)) return fail(MGK_EINVAL, "mgk_tail_cycle_cs_f64: ctab and dtab go together");
    return tail_cycle<double>(c, g0, nlev, n, ctab ? nullptr : coef7, ctab ? nullptr : dinv, scale, v0, v1, b, u, stream, ctab, dtab, &coarse_scale);
}
extern "C" int mgk_tail_max_n(int dim) { return dim == 3 ? 15 : 63; }


// ------------------------------------------------------------------------------------------
// 2-D fused residual + full weighting: b_c = R (b - A u), the fine residual is never written (src/solver.c:1534-1535).
// Every WAVE is independent (no LDS, no barrier): it owns 64 adjacent column pairs of the fine grid -- lanes 1 .. 62 produce the
// coarse columns centred on their pairs, lane 0 only supplies the west neighbour of lane 1 and lane 63 the residual east of lane
// 62 (tiles overlap by two lanes) -- and marches along y over a chunk of coarse rows with the rows y-1, y, y+1 of u in a register
// ring loaded PD rows ahead; x neighbours by DPP lane shifts.  The running sums follow the row of res exactly: (di, dj)
// ascending, an even fine row being di = 2 of the coarse row below and di = 0 of the one above (equal weights: one product).
// Traffic: 16 B per fine unknown read + 2 B written instead of 24 + 10.
// ------------------------------------------------------------------------------------------
struct RR2dArgs {
    const double *u, *b;
    double *bc, *uc0;
    int nx, ny, nxc, nyc;
    long rs, crs;
    int ntx, ycc;
    double a0, a2, a3, a4, a6, dinv_c, scale_c;
    const double *ctab, *dtab_c;    // optional (stretched meshes): coefficients per FINE grid row, 1/diag per COARSE grid row
};
template <int PD>
__global__ void __launch_bounds__(256) k_rr2d(const RR2dArgs a) {
    constexpr int NP = PD + 3;
    using VT = V16<double>;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // (row indices and row addresses on the scalar unit)
    const int tx = wid % a.ntx, cy = wid / a.ntx;
    const int ic0 = cy * a.ycc, ic1 = min(ic0 + a.ycc, a.nyc);
    if (ic0 >= ic1) return;                                   // whole wave
    const int p = tx * 62 + lane - 1;                         // pair index = coarse column of this lane
    const int x0 = 2 * p;
    const bool xok = (x0 >= 0 && x0 < a.nx);
    const bool store = (lane >= 1 && lane <= 62 && p < a.nxc);
    const int xc = min(max(x0, 0), a.nx - 1);                 // clamped: every lane loads from inside the row, invalid lanes are zeroed
    const double *up_ = a.u + xc, *bp_ = a.b + xc;
    const int y0 = 2 * ic0, y1 = 2 * ic1 + 1;                 // fine rows y0 .. y1-1 are processed, row y1 is the last one read
    auto ldu = [&](int y) -> VT {
        VT v = *reinterpret_cast<const VT *>(up_ + (long)min(y, y1) * a.rs);
        if (!xok) { v.v[0] = 0.0; v.v[1] = 0.0; }
        return v;
    };
    auto ldb = [&](int y) -> VT {
        VT v = ldv_stream(bp_ + (long)min(y, y1 - 1) * a.rs, true);
        if (!xok) { v.v[0] = 0.0; v.v[1] = 0.0; }
        return v;
    };
    const double w2[3][3] = {{0.0625, 0.125, 0.0625}, {0.125, 0.25, 0.125}, {0.0625, 0.125, 0.0625}};
    VT U[NP], B[NP];
#pragma unroll
    for (int q = -1; q <= PD; q++) U[(q + NP) % NP] = ldu(y0 + q);
#pragma unroll
    for (int q = 0; q < PD; q++) B[q] = ldb(y0 + q);
    double acc = 0.0, accn = 0.0;
    for (int yb = y0; yb < y1; yb += NP) {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int y = yb + k;
            if (y < y1) {
                const int cm = (k + NP - 1) % NP, cc = k, cp = (k + 1) % NP;
                U[(k + PD + 1) % NP] = ldu(y + PD + 1);
                B[(k + PD) % NP] = ldb(y + PD);
                const VT &c = U[cc];
                const double Wv = lane_up<true>(c.v[1]), Ev = lane_dn<true>(c.v[0]);
                double k0 = a.a0, k2 = a.a2, k3 = a.a3, k4 = a.a4, k6 = a.a6;
                if (a.ctab) { const CDBL4 *cr = (const CDBL4 *)(a.ctab + 5 * (long)y); k0 = cr[0]; k2 = cr[1]; k3 = cr[2]; k4 = cr[3]; k6 = cr[4]; }
                VT r;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const double wv = (e == 0) ? Wv : c.v[0];
                    const double ev = (e == 1) ? Ev : c.v[1];
                    double t = k0 * U[cm].v[e];
                    t = t + k2 * wv;
                    t = t + k3 * c.v[e];
                    t = t + k4 * ev;
                    t = t + k6 * U[cp].v[e];
                    r.v[e] = B[cc].v[e] - t;
                    if (!xok || x0 + e >= a.nx) r.v[e] = 0.0;
                }
                const double rn = lane_dn<true>(r.v[0]);
                const bool even = ((y & 1) == 0);
                const int di = even ? 0 : 1;
                double pr = w2[di][0] * r.v[0];
                acc += pr; if (even) accn += pr;
                pr = w2[di][1] * r.v[1];
                acc += pr; if (even) accn += pr;
                pr = w2[di][2] * rn;
                acc += pr; if (even) accn += pr;
                if (even) {
                    const int ic = y / 2 - 1;                 // completed coarse row
                    if (ic >= ic0 && store) {
                        const long oc = (long)ic * a.crs + p;
                        a.bc[oc] = acc;
                        if (a.uc0) { const double zq = acc * (a.dtab_c ? a.dtab_c[ic] : a.dinv_c); a.uc0[oc] = a.scale_c * zq; }
                    }
                    acc = accn; accn = 0.0;
                }
            }
        }
    }
}
static int residual_restrict_2d(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const double *ctab, const double *dtab_c,
                                const double *b, const double *u, double *bc, double *uc0, double dinv_c, double scale_c, void *stream) {
    if (!c || !gf || !gc || (!coef && !ctab) || !b || !u || !bc || gf->dim != 2 || gc->dim != 2 || (uc0 && ctab && !dtab_c))
        return fail(MGK_EINVAL, "mgk_residual_restrict_2d_f64: bad arguments (2-D)");
    if (gf->nx != 2 * gc->nx + 1 || gf->ny != 2 * gc->ny + 1) return fail(MGK_EINVAL, "mgk_residual_restrict_2d_f64: need nf = 2 nc + 1");
    RR2dArgs a; memset(&a, 0, sizeof(a));
    a.u = u + gf->org; a.b = b + gf->org; a.bc = bc + gc->org; a.uc0 = uc0 ? uc0 + gc->org : nullptr;
    a.nx = gf->nx; a.ny = gf->ny; a.nxc = gc->nx; a.nyc = gc->ny; a.rs = gf->pitch; a.crs = gc->pitch;
    if (coef) { a.a0 = coef[0]; a.a2 = coef[1]; a.a3 = coef[2]; a.a4 = coef[3]; a.a6 = coef[4]; }
    a.dinv_c = dinv_c; a.scale_c = scale_c; a.ctab = ctab; a.dtab_c = dtab_c;
    a.ntx = (gc->nx + 61) / 62;
    long nch = (4096 + a.ntx - 1) / a.ntx;                    // ~4096 waves (16 per CU); every chunk re-reads two fine rows
    if (g_zchunk > 0) nch = (gc->ny + g_zchunk - 1) / g_zchunk;
    int ycc = (int)((gc->ny + nch - 1) / nch);
    if (ycc < 4) ycc = 4;
    if (ycc > gc->ny) ycc = gc->ny;
    // round 3: the levels that fit the caches (rows of <= 1024) lack waves, not bandwidth: chunks of TWO coarse rows whose seven fine rows
    // of u (and five of b) are all requested before the first residual (PD = 5: the register ring holds the whole chunk), as the short
    // form of k_jacobi3_2d.  Tuning variants 55 / 56 force the marching / the short form.
    const bool shortf = (g_variant == 56) || (g_variant != 55 && g_zchunk <= 0 && gf->nx + 1 <= 1024);
    if (shortf) ycc = gc->ny < 2 ? gc->ny : 2;
    a.ycc = ycc;
    const long waves = (long)a.ntx * ((gc->ny + ycc - 1) / ycc);
    if (shortf) hipLaunchKernelGGL((k_rr2d<5>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, S(c, stream), a);
    else hipLaunchKernelGGL((k_rr2d<2>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_residual_restrict_2d_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, const double *b,
                                            const double *u, double *bc, double *uc0, double dinv_c, double scale_c, void *stream) {
    return residual_restrict_2d(c, gf, gc, coef, nullptr, nullptr, b, u, bc, uc0, dinv_c, scale_c, stream);
}
extern "C" int mgk_residual_restrict_2d_rowcoef_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *ctab_f, const double *b,
                                                    const double *u, double *bc, double *uc0, const double *dtab_c, double scale_c, void *stream) {
    if (!ctab_f) return fail(MGK_EINVAL, "mgk_residual_restrict_2d_rowcoef_f64: null table");
    return residual_restrict_2d(c, gf, gc, nullptr, ctab_f, dtab_c, b, u, bc, uc0, 1.0, scale_c, stream);
}

// 2-D form of mgk_sweep_residual_restrict_f64: out = J(u); bc = R (b - A out) [; uc0 = scale_c * (bc * dinv_c)] in one pass.
// Like k_rr2d every WAVE is independent (no LDS, no barrier): lane l holds the column pair x0 = 2 (61 tx + l - 1); all lanes make
// the sweep (the outer column of lanes 0 / 63 is not used), lanes 1 .. 62 the residual, lanes 1 .. 61 full weighting and the stores:
// tiles overlap by three lanes.  Marches along y over a chunk of coarse rows; the chunk re-reads four fine rows.
struct SRR2dArgs {
    const double *u, *b;
    double *out, *bc, *uc0;
    int nx, ny, nxc, nyc;
    long rs, crs;
    int ntx, ycc;
    double a0, a2, a3, a4, a6, dinv, scale, dinv_c, scale_c;
    const double *ctab, *dtab, *dtab_c;     // optional (stretched meshes): coefficients and 1/diag per FINE grid row, 1/diag per COARSE grid row
};
__global__ void __launch_bounds__(256) k_srr2d(const SRR2dArgs a) {
    using VT = V16<double>;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // (row indices and row addresses on the scalar unit)
    const int tx = wid % a.ntx, cy = wid / a.ntx;
    const int ic0 = cy * a.ycc, ic1 = min(ic0 + a.ycc, a.nyc);
    if (ic0 >= ic1) return;                                   // whole wave
    const int pidx = tx * 61 + lane - 1;                      // pair index = coarse column of this lane
    const int x0 = 2 * pidx;
    const bool xin = (x0 >= 0 && x0 < a.nx);
    const bool lastvec = (x0 + 2 > a.nx);                     // its second column is the ghost column x = nx
    const bool store = (lane >= 1 && lane <= 61 && pidx < a.nxc + 1 && xin);       // stores of the swept field (pair inside the grid)
    const bool cstore = (lane >= 1 && lane <= 61 && pidx < a.nxc);
    const int xc = min(max(x0, 0), a.nx - 1);                 // clamped: every lane loads from inside the row, invalid lanes are zeroed
    const double *up_ = a.u + xc, *bp_ = a.b + xc;
    const int y0 = 2 * ic0, y1 = min(2 * ic1 + 1, a.ny);      // rows whose residual this chunk forms: [y0, y1)
    const int ys1 = (ic1 == a.nyc) ? a.ny : 2 * ic1;          // rows of the swept field this chunk stores: [y0, ys1)
    auto ldu = [&](int r) -> VT {
        VT v = *reinterpret_cast<const VT *>(up_ + (long)min(max(r, -1), a.ny) * a.rs);
        if (!xin) { v.v[0] = 0.0; v.v[1] = 0.0; }
        return v;
    };
    auto ldb = [&](int r) -> VT {
        VT v = ldv_stream(bp_ + (long)min(max(r, 0), a.ny - 1) * a.rs, true);
        if (!xin) { v.v[0] = 0.0; v.v[1] = 0.0; }
        return v;
    };
    const double w2[3][3] = {{0.0625, 0.125, 0.0625}, {0.125, 0.25, 0.125}, {0.0625, 0.125, 0.0625}};
    const int t0 = y0 - 2;                                    // first step: the sweep of row y0 - 1
    VT ua = ldu(t0), ub = ldu(t0 + 1), uc = ldu(t0 + 2), ud;
    VT b1 = ldb(t0 + 1), bn, b0 = v16_zero<double>();
    VT wm = b0, wc = b0, wp = b0;                             // swept rows t-1, t, t+1
    double acc = 0.0, accn = 0.0;
    for (int t = t0; t < y1; t++) {
        const int p = t + 1;
        ud = ldu(t + 3);
        bn = ldb(t + 2);
        // ---- the sweep of row p ----
        {
            const double Wv = lane_up<true>(ub.v[1]), Ev = lane_dn<true>(ub.v[0]);
            const bool pin = (p >= 0 && p < a.ny);
            double k0 = a.a0, k2 = a.a2, k3 = a.a3, k4 = a.a4, k6 = a.a6, kd = a.dinv;
            if (a.ctab) { const int pr_ = min(max(p, 0), a.ny - 1); const CDBL4 *cr = (const CDBL4 *)(a.ctab + 5 * (long)pr_); k0 = cr[0]; k2 = cr[1]; k3 = cr[2]; k4 = cr[3]; k6 = cr[4]; kd = ((const CDBL4 *)(a.dtab + pr_))[0]; }
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const double wv = (e == 0) ? Wv : ub.v[0];
                const double ev = (e == 1) ? Ev : ub.v[1];
                double s = k0 * ua.v[e];
                s = s + k2 * wv;
                s = s + k3 * ub.v[e];
                s = s + k4 * ev;
                s = s + k6 * uc.v[e];
                const double res = b1.v[e] - s;
                const double zz = res * kd;
                wp.v[e] = ub.v[e] + a.scale * zz;
                if (!pin || !xin || (lastvec && e == 1)) wp.v[e] = 0.0;
            }
            if (store && p >= y0 && p < ys1) stv_policy(a.out + (long)p * a.rs + x0, wp, mgk_store_nt_2d(a.ny, a.rs));
        }
        // ---- residual of the swept row t, full weighting ----
        if (t >= y0) {
            const double Wv = lane_up<true>(wc.v[1]), Ev = lane_dn<true>(wc.v[0]);
            double k0 = a.a0, k2 = a.a2, k3 = a.a3, k4 = a.a4, k6 = a.a6;
            if (a.ctab) { const CDBL4 *cr = (const CDBL4 *)(a.ctab + 5 * (long)t); k0 = cr[0]; k2 = cr[1]; k3 = cr[2]; k4 = cr[3]; k6 = cr[4]; }
            VT r;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const double wv = (e == 0) ? Wv : wc.v[0];
                const double ev = (e == 1) ? Ev : wc.v[1];
                double s = k0 * wm.v[e];
                s = s + k2 * wv;
                s = s + k3 * wc.v[e];
                s = s + k4 * ev;
                s = s + k6 * wp.v[e];
                r.v[e] = b0.v[e] - s;
                if (!xin || (lastvec && e == 1)) r.v[e] = 0.0;
            }
            const double rn = lane_dn<true>(r.v[0]);
            const bool even = ((t & 1) == 0);
            const int di = even ? 0 : 1;
            double pr = w2[di][0] * r.v[0];
            acc += pr; if (even) accn += pr;
            pr = w2[di][1] * r.v[1];
            acc += pr; if (even) accn += pr;
            pr = w2[di][2] * rn;
            acc += pr; if (even) accn += pr;
            if (even) {
                const int ic = t / 2 - 1;                 // completed coarse row
                if (ic >= ic0 && cstore) {
                    const long oc = (long)ic * a.crs + pidx;
                    a.bc[oc] = acc;
                    if (a.uc0) { const double zq = acc * (a.dtab_c ? a.dtab_c[ic] : a.dinv_c); a.uc0[oc] = a.scale_c * zq; }
                }
                acc = accn; accn = 0.0;
            }
        }
        b0 = b1; b1 = bn; wm = wc; wc = wp; ua = ub; ub = uc; uc = ud;
    }
}
static int sweep_residual_restrict_2d(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                      const double *ctab, const double *dtab, const double *dtab_c,
                                      const double *b, const double *u, double *unew, double *bc, double *uc0,
                                      double dinv_c, double scale_c, void *stream) {
    if (!c || !gf || !gc || (!coef && !ctab) || (ctab && !dtab) || (ctab && uc0 && !dtab_c) || !b || !u || !unew || u == unew || !bc || gf->dim != 2 || gc->dim != 2)
        return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_2d_f64: bad arguments (2-D)");
    if (gf->nx != 2 * gc->nx + 1 || gf->ny != 2 * gc->ny + 1) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_2d_f64: need nf = 2 nc + 1");
    SRR2dArgs a; memset(&a, 0, sizeof(a));
    a.u = u + gf->org; a.b = b + gf->org; a.out = unew + gf->org; a.bc = bc + gc->org; a.uc0 = uc0 ? uc0 + gc->org : nullptr;
    a.nx = gf->nx; a.ny = gf->ny; a.nxc = gc->nx; a.nyc = gc->ny; a.rs = gf->pitch; a.crs = gc->pitch;
    if (coef) { a.a0 = coef[0]; a.a2 = coef[1]; a.a3 = coef[2]; a.a4 = coef[3]; a.a6 = coef[4]; }
    a.dinv = dinv; a.scale = scale; a.dinv_c = dinv_c; a.scale_c = scale_c;
    a.ctab = ctab; a.dtab = dtab; a.dtab_c = dtab_c;
    a.ntx = (gc->nx + 1 + 60) / 61;                           // pairs 0 .. nxc (the last one holds the last fine column and the ghost column)
    long nch = (4096 + a.ntx - 1) / a.ntx;                    // ~4096 waves (16 per CU); every chunk re-reads four fine rows
    if (g_zchunk > 0) nch = (gc->ny + g_zchunk - 1) / g_zchunk;
    int ycc = (int)((gc->ny + nch - 1) / nch);
    if (ycc < 8 && g_zchunk <= 0) ycc = 8;
    if (ycc > gc->ny) ycc = gc->ny;
    a.ycc = ycc;
    const long waves = (long)a.ntx * ((gc->ny + ycc - 1) / ycc);
    hipLaunchKernelGGL(k_srr2d, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_sweep_residual_restrict_2d_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                                  const double *b, const double *u, double *unew, double *bc, double *uc0,
                                                  double dinv_c, double scale_c, void *stream) {
    if (!coef) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_2d_f64: bad arguments (2-D)");
    return sweep_residual_restrict_2d(c, gf, gc, coef, dinv, scale, nullptr, nullptr, nullptr, b, u, unew, bc, uc0, dinv_c, scale_c, stream);
}
// stretched meshes: ctab_f / dtab_f of the FINE level, dtab_c (1/diag per grid row) of the COARSE level (needed only with uc0)
extern "C" int mgk_sweep_residual_restrict_2d_rowcoef_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *ctab_f, const double *dtab_f,
                                                          double scale, const double *b, const double *u, double *unew, double *bc, double *uc0,
                                                          const double *dtab_c, double scale_c, void *stream) {
    if (!ctab_f) return fail(MGK_EINVAL, "mgk_sweep_residual_restrict_2d_rowcoef_f64: null table");
    return sweep_residual_restrict_2d(c, gf, gc, nullptr, 1.0, scale, ctab_f, dtab_f, dtab_c, b, u, unew, bc, uc0, 1.0, scale_c, stream);
}

// ------------------------------------------------------------------------------------------
// 2-D operators whose five coefficients depend on the grid row only (the reference's stretched meshes,
// -mesh 1/2: metrics are functions of y, src/mesh.c:45-107, src/problem.c:3-22).  Same marching kernel; the
// coefficients of a marching step come from a device table instead of launch constants.
//   ctab: ny x 5 doubles {(i-1), (j-1), C, (j+1), (i+1)} per grid row;  dtab: ny doubles 1/diag (PCJACOBI)
// mode: 0 Jacobi sweep, 1 residual, 4 apply
// ------------------------------------------------------------------------------------------
extern "C" int mgk_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, int mode, const double *ctab, const double *dtab, double scale,
                               const double *b, const double *u, double *out, void *stream) {
    if (!c || !g || g->dim != 2 || !ctab || !u || !out || u == out || (mode != MODE_APPLY && !b) || (mode == MODE_JACOBI && !dtab))
        return fail(MGK_EINVAL, "mgk_rowcoef_f64: bad arguments (2-D only)");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = (b ? b : u) + g->org; a.out = out + g->org;
    a.ctab = ctab; a.dtab = dtab; a.scale = scale; a.dinv = 1.0;
    hipStream_t s = S(c, stream);
    if (mode == MODE_JACOBI) return dispatch_st<MODE_JACOBI>(c, g, a, s, nullptr);
    if (mode == MODE_RESIDUAL) return dispatch_st<MODE_RESIDUAL>(c, g, a, s, nullptr);
    if (mode == MODE_APPLY) return dispatch_st<MODE_APPLY>(c, g, a, s, nullptr);
    return fail(MGK_EINVAL, "mgk_rowcoef_f64: unknown mode");
}
// Chebyshev step on the row-table operator (stretched meshes, 2-D): pkp1 = (c_km1 * pkm1 + c_k * pk) + c_z * ((b - A pk) * dtab[i])
extern "C" int mgk_cheby_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *dtab, double c_km1, double c_k, double c_z,
                                     const double *b, const double *pk, const double *pkm1, double *pkp1, void *stream) {
    if (!c || !g || g->dim != 2 || !ctab || !dtab || !b || !pk || !pkm1 || !pkp1 || pk == pkp1 || pkm1 == pkp1)
        return fail(MGK_EINVAL, "mgk_cheby_rowcoef_f64: bad arguments (2-D only)");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = pk + g->org; a.b = b + g->org; a.aux = pkm1 + g->org; a.out = pkp1 + g->org;
    a.ctab = ctab; a.dtab = dtab; a.dinv = 1.0; a.ckm1 = c_km1; a.ck = c_k; a.cz = c_z;
    return dispatch_st<MODE_CHEBY>(c, g, a, S(c, stream), nullptr);
}
// stretched meshes (2-D): the fused forms of the cycle on the row-table operator -- k_stencil reads its coefficients per marching
// step in every mode, so these are the constant-coefficient entry points with the tables in place of the constants
extern "C" int mgk_jacobi_sumsq_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *dtab, double scale,
                                            const double *b, const double *u, double *unew, double *sumsq_host, void *stream) {
    if (!c || !g || g->dim != 2 || !ctab || !dtab || !b || !u || !unew || u == unew || !sumsq_host) return fail(MGK_EINVAL, "mgk_jacobi_sumsq_rowcoef_f64: bad arguments (2-D)");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org; a.partials = c->partials;
    a.ctab = ctab; a.dtab = dtab; a.scale = scale; a.dinv = 1.0;
    int nblk = 0;
    int rc = dispatch_st<MODE_JNORM>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}
extern "C" int mgk_residual_sumsq_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *ctab, const double *b, const double *u,
                                              double *sumsq_host, void *stream) {
    if (!c || !g || g->dim != 2 || !ctab || !b || !u || !sumsq_host) return fail(MGK_EINVAL, "mgk_residual_sumsq_rowcoef_f64: bad arguments (2-D)");
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.partials = c->partials;
    a.ctab = ctab; a.dinv = 1.0;
    int nblk = 0;
    int rc = dispatch_st<MODE_RESNORM>(c, g, a, S(c, stream), &nblk);
    if (rc) return rc;
    return finish_to_host(c, nblk, 1, S(c, stream), sumsq_host);
}
extern "C" int mgk_prolong_jacobi_rowcoef_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *ctab, const double *dtab, double scale,
                                              const double *b, const double *uc, const double *u, double *unew, void *stream) {
    if (!c || !gf || !gc || gf->dim != 2 || !ctab || !dtab || !b || !uc || !u || !unew || u == unew) return fail(MGK_EINVAL, "mgk_prolong_jacobi_rowcoef_f64: bad arguments (2-D)");
    XferArgs x;
    int rc = xfer_args(gf, gc, x);
    if (rc) return rc;
    if ((gf->nx >= 2047 && g_variant != 30) || (g_variant == 38 && gf->nx >= 3))          // independent waves, as mgk_prolong_jacobi_f64 in 2-D
        return prolong_jacobi_2d_waves(c, gf, gc, nullptr, 1.0, scale, b, uc, u, unew, stream, ctab, dtab);
    StArgs<double> a; memset(&a, 0, sizeof(a));
    a.u = u + gf->org; a.b = b + gf->org; a.out = unew + gf->org;
    a.uc = uc + gc->org; a.crs = gc->pitch; a.cms = gc->pitch;
    a.nxc = gc->nx; a.nyc = gc->ny; a.nzc = gc->nz;
    a.ctab = ctab; a.dtab = dtab; a.scale = scale; a.dinv = 1.0;
    a.zbeg = 0; a.zend = gf->ny;
    return dispatch_st<MODE_PJACOBI>(c, gf, a, S(c, stream), nullptr);
}
extern "C" int mgk_jacobi_zero_rowcoef_f64(mgk_ctx *c, const mgk_geom *g, const double *dtab, double scale,
                                           const double *b, double *unew, void *stream) {
    if (!c || !g || g->dim != 2 || !dtab || !b || !unew) return fail(MGK_EINVAL, "mgk_jacobi_zero_rowcoef_f64: bad arguments");
    RowArgs a = row_args(g);
    dim3 grid, block;
    row_grid(a, a.npairs, grid, block, 1024);
    hipLaunchKernelGGL(k_jacobi_zero<double>, grid, block, 0, S(c, stream), a, 1.0, scale, b + g->org, unew + g->org, dtab);
    HIPCHK(hipGetLastError());
    return 0;
}

