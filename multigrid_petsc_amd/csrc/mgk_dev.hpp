// mgk_dev.hpp -- what the translation units of libmgk.so share: context, error reporting, device helpers (16-byte lane vectors,
// streaming accesses, wavefront reductions and DPP lane shifts, buffer-descriptor accesses).  Moved out of mgk_kernels.hip in round 3
// so that new kernels live in a file of their own (mgk_kernels3.hip) and compile in seconds.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include "mgk.h"

extern thread_local char g_err[512];
int fail(int code, const char *what);
#define HIPCHK(call)                                                             \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((int)e_, #call); } while (0)

struct mgk_ctx {
    int device;
    hipStream_t compute, comm;
    double *partials;      // reduction scratch (device)
    double *result_dev;    // 8 doubles (device)
    double *result_host;   // 8 doubles (pinned host)
    int max_partials;
    hipEvent_t ev[32];     // ring of dependency events for mgk_stream_wait (no create/destroy on the hot path)
    int ev_next;
    double *defer_slot;    // non-null: the next single-value reductions deposit here (device) instead of syncing to the host
    int chunk_planes;      // > 0: the 3-D marching kernels launched on this context cut z into chunks of about this many planes
                           // (slab ranks: short blocks, so that the exchange kernels of the comm stream find CUs beside them)
};
static inline hipStream_t S(mgk_ctx *c, void *s) { return s ? (hipStream_t)s : c->compute; }
// a double in the CONSTANT address space: data that is read-only for the life of a kernel (the row tables of the stretched meshes) read through
// such a pointer at a wave-uniform address becomes a scalar load; through an ordinary pointer of the argument struct it stays a vector load
typedef double __attribute__((address_space(4))) CDBL4;
// per THREAD tuning knobs (mgk_set_tuning) and the fixed-order finish of per-block / per-wave partial sums (mgk_kernels.hip)
extern thread_local int g_variant, g_zchunk;
int finish_to_host(mgk_ctx *c, int nparts, int nslots, hipStream_t s, double *host_out);
int mgk_preload_kernels3();      // forces the code object of mgk_kernels3.hip to load (mgk_ctx_create)

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double2 ld2(const double *p, bool ok) {
    return ok ? *reinterpret_cast<const double2 *>(p) : make_double2(0.0, 0.0);
}
__device__ __forceinline__ double ld1(const double *p, bool ok) { return ok ? *p : 0.0; }
// streaming (read-once / write-once) accesses: non-temporal so that they do not displace the u planes
// and halo lines that neighbouring tiles re-read from L2 / Infinity Cache
#ifndef MGK_NT
#define MGK_NT 3
#endif
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2_stream(const double *p, bool ok) {
#if MGK_NT & 1
    if (!ok) return make_double2(0.0, 0.0);
    d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
    return make_double2(v.x, v.y);
#else
    return ld2(p, ok);
#endif
}
__device__ __forceinline__ void st2_stream(double *p, double2 v) {
#if MGK_NT & 2
    d2v t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<d2v *>(p));
#else
    *reinterpret_cast<double2 *>(p) = v;
#endif
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
// block-wide sum, result valid in thread 0.  red: >= 16 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) for (int q = 0; q < nw; q++) s += red[q];
    return s;
}
__device__ __forceinline__ double block_max(double v, double *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) for (int q = 0; q < nw; q++) s = fmax(s, red[q]);
    return s;
}


// 16-byte lane vector: 2 doubles or 4 floats
template <typename T> struct alignas(16) V16 { T v[16 / sizeof(T)]; };
typedef float f4v __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ V16<T> v16_zero() {
    V16<T> r;
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); e++) r.v[e] = (T)0;
    return r;
}
template <typename T> __device__ __forceinline__ V16<T> ldv(const T *p, bool ok) {
    return ok ? *reinterpret_cast<const V16<T> *>(p) : v16_zero<T>();
}
__device__ __forceinline__ V16<double> ldv_stream(const double *p, bool ok) {
    double2 t = ld2_stream(p, ok);
    V16<double> r; r.v[0] = t.x; r.v[1] = t.y;
    return r;
}
__device__ __forceinline__ V16<float> ldv_stream(const float *p, bool ok) {
    V16<float> r = v16_zero<float>();
    if (ok) {
#if MGK_NT & 1
        f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
#else
        r = *reinterpret_cast<const V16<float> *>(p);
#endif
    }
    return r;
}
__device__ __forceinline__ void stv_stream(double *p, const V16<double> &v) { st2_stream(p, make_double2(v.v[0], v.v[1])); }
// The store of a 2-D pass: non-temporal only when the field is larger than the 256 MB Infinity Cache.  Measured at 4095^2 (134 MB per field,
// tools/exp_incycle_2d.py): a pass that READS what its predecessor wrote with non-temporal stores takes 83-85 us, with ordinary stores 73 us
// (launched alone, reading fields nobody wrote: 70 / 71 us) -- and in a multigrid cycle every pass reads what the previous one wrote.  The
// 3-D kernels keep their non-temporal stores: at 511^3 and 1023^3 they are worth 5-9 %, at 255^3 (133 MB) the two policies measured the same.
__device__ __forceinline__ bool mgk_store_nt_2d(int ny, long rs) { return (long)ny * rs * 8 > (256L << 20); }
__device__ __forceinline__ void stv_policy(double *p, const V16<double> &v, bool nt) {
    if (nt) stv_stream(p, v); else *reinterpret_cast<V16<double> *>(p) = v;
}
__device__ __forceinline__ void stv_stream(float *p, const V16<float> &v) {
#if MGK_NT & 2
    f4v t; t.x = v.v[0]; t.y = v.v[1]; t.z = v.v[2]; t.w = v.v[3];
    __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p));
#else
    *reinterpret_cast<V16<float> *>(p) = v;
#endif
}

// lane i <- lane i-1 / lane i+1 of the wavefront.  DPP form: whole-wavefront shifts (wave_shr:1 / wave_shl:1) on the vector ALU
// instead of ds_bpermute on the LDS pipe; lane 0 / 63 keep their own value (the callers replace it by the wave-edge value).
template <bool DPP> __device__ __forceinline__ double lane_up(double v) {
    if (!DPP) return __shfl_up(v, 1, 64);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <bool DPP> __device__ __forceinline__ double lane_dn(double v) {
    if (!DPP) return __shfl_down(v, 1, 64);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <bool DPP> __device__ __forceinline__ float lane_up(float v) {
    if (!DPP) return __shfl_up(v, 1, 64);
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
template <bool DPP> __device__ __forceinline__ float lane_dn(float v) {
    if (!DPP) return __shfl_down(v, 1, 64);
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// lane shifts that deliver `old` to the lane without a source (lane 0 / lane 63): the DPP shift keeps the old operand there
__device__ __forceinline__ double lane_up_old(double v, double old) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_dn_old(double v, double old) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_up_old(float v, float old) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_dn_old(float v, float old) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
// loads / stores through a buffer descriptor (one per plane, built on the scalar unit): the row offset is a 32-bit SGPR operand, the lane
// offset one constant VGPR -- no 64-bit vector address arithmetic
typedef unsigned int mgk_u4v __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ V16<T> bufld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    mgk_u4v v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(V16<T>, v);
}
template <typename T> __device__ __forceinline__ V16<T> bufld_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    mgk_u4v v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
    return __builtin_bit_cast(V16<T>, v);
}
template <typename T> __device__ __forceinline__ void bufst_nt(const V16<T> &x, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mgk_u4v, x), r, voff, soff, (MGK_NT & 2) ? 2 : 0);
}
