"""ctypes binding of include/mgk.h (kernel-level C ABI).  Test/bench plumbing only."""
import ctypes as C
import numpy as np
from ._lib import load_mgk

c_dp = C.POINTER(C.c_double)


class Geom(C.Structure):
    _fields_ = [("dim", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("pitch", C.c_int), ("plane", C.c_long), ("org", C.c_long), ("total", C.c_long)]


class MgkError(RuntimeError):
    pass


def _sigs(L):
    vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    G = C.POINTER(Geom)
    S = {
        "mgk_geom_init": (i, [G, i, i, i, i]),
        "mgk_device_count": (i, []),
        "mgk_set_device": (i, [i]),
        "mgk_ctx_create": (i, [C.POINTER(vp), i]),
        "mgk_ctx_destroy": (None, [vp]),
        "mgk_last_error": (C.c_char_p, []),
        "mgk_stream_compute": (vp, [vp]),
        "mgk_stream_comm": (vp, [vp]),
        "mgk_malloc": (i, [vp, C.POINTER(vp), sz]),
        "mgk_free": (i, [vp, vp]),
        "mgk_memset0": (i, [vp, vp, sz, vp]),
        "mgk_h2d": (i, [vp, vp, vp, sz]),
        "mgk_d2h": (i, [vp, vp, vp, sz]),
        "mgk_d2d": (i, [vp, vp, vp, sz, vp]),
        "mgk_sync": (i, [vp, vp]),
        "mgk_timer_create": (i, [vp, C.POINTER(vp)]),
        "mgk_timer_start": (i, [vp, vp, vp]),
        "mgk_timer_stop": (i, [vp, vp, vp]),
        "mgk_timer_elapsed_ms": (i, [vp, vp, C.POINTER(d)]),
        "mgk_timer_destroy": (None, [vp, vp]),
        "mgk_stream_wait": (i, [vp, vp, vp]),
        "mgk_pack_f64": (i, [vp, G, vp, vp, vp]),
        "mgk_unpack_f64": (i, [vp, G, vp, vp, vp]),
        "mgk_jacobi_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp]),
        "mgk_jacobi_zero_f64": (i, [vp, G, d, d, vp, vp, vp]),
        "mgk_cheby_f64": (i, [vp, G, c_dp, d, d, d, d, vp, vp, vp, vp, vp]),
        "mgk_residual_f64": (i, [vp, G, c_dp, vp, vp, vp, vp]),
        "mgk_residual_sumsq_f64": (i, [vp, G, c_dp, vp, vp, C.POINTER(d), vp]),
        "mgk_jacobi2_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp]),
        "mgk_jacobi2_f32": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp]),
        "mgk_jacobi2_2d_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp]),
        "mgk_jacobi2_2d_rowcoef_f64": (i, [vp, G, vp, vp, d, vp, vp, vp, vp]),
        "mgk_jacobi_sumsq_rowcoef_f64": (i, [vp, G, vp, vp, d, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_jacobi_sumsq_store_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_apply_add_f64": (i, [vp, G, c_dp, vp, vp, vp]),
        "mgk_window_add_f64": (i, [vp, G, G, i, vp, vp, vp, vp]),
        "mgk_dense_mult_f64": (i, [vp, i, i, vp, vp, vp, vp]),
        "mgk_residual_sumsq_rowcoef_f64": (i, [vp, G, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_prolong_jacobi_rowcoef_f64": (i, [vp, G, G, vp, vp, d, vp, vp, vp, vp, vp]),
        "mgk_residual_restrict_2d_rowcoef_f64": (i, [vp, G, G, vp, vp, vp, vp, vp, vp, d, vp]),
        "mgk_jacobi2_2d_sumsq_rowcoef_f64": (i, [vp, G, vp, vp, d, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_sweep_residual_restrict_2d_rowcoef_f64": (i, [vp, G, G, vp, vp, d, vp, vp, vp, vp, vp, vp, d, vp]),
        "mgk_tail_cycle_rowcoef_f64": (i, [vp, G, i, C.POINTER(i), C.POINTER(vp), C.POINTER(vp), d, i, i, vp, vp, vp]),
        "mgk_jacobi_sumsq_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_restrict_fw_f64": (i, [vp, G, G, vp, vp, vp]),
        "mgk_prolong_add_f64": (i, [vp, G, G, vp, vp, vp]),
        "mgk_sumsq_f64": (i, [vp, G, vp, C.POINTER(d), vp]),
        "mgk_fill_separable_f64": (i, [vp, G, vp, vp, vp, vp, vp]),
        "mgk_error_sums_f64": (i, [vp, G, vp, vp, vp, vp, c_dp, vp]),
        "mgk_set_tuning": (None, [i, i]),
        "mgk_geom_init_f32": (i, [G, i, i, i, i]),
        "mgk_jacobi_f32": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp]),
        "mgk_jacobi_zero_f32": (i, [vp, G, d, d, vp, vp, vp]),
        "mgk_residual_f32": (i, [vp, G, c_dp, vp, vp, vp, vp]),
        "mgk_restrict_fw_f32": (i, [vp, G, G, vp, vp, vp]),
        "mgk_prolong_add_f32": (i, [vp, G, G, vp, vp, vp]),
        "mgk_residual_f64_to_f32": (i, [vp, G, G, c_dp, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_correct_f64_from_f32": (i, [vp, G, G, vp, vp, vp]),
        "mgk_correct_residual_f64_f32": (i, [vp, G, G, c_dp, vp, vp, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_pack_f32": (i, [vp, G, vp, vp, vp]),
        "mgk_unpack_f32": (i, [vp, G, vp, vp, vp]),
        "mgk_apply_f64": (i, [vp, G, c_dp, vp, vp, vp]),
        "mgk_residual_restrict_f64": (i, [vp, G, G, c_dp, vp, vp, vp, vp]),
        "mgk_residual_restrict_f32": (i, [vp, G, G, c_dp, vp, vp, vp, vp]),
        "mgk_residual_restrict_jz_f64": (i, [vp, G, G, c_dp, vp, vp, vp, vp, d, d, vp]),
        "mgk_residual_restrict_jz_f32": (i, [vp, G, G, c_dp, vp, vp, vp, vp, d, d, vp]),
        "mgk_residual_range_f64": (i, [vp, G, c_dp, vp, vp, vp, i, i, vp]),
        "mgk_residual_range_f32": (i, [vp, G, c_dp, vp, vp, vp, i, i, vp]),
        "mgk_restrict_finish_f64": (i, [vp, G, G, vp, vp, vp]),
        "mgk_restrict_finish_f32": (i, [vp, G, G, vp, vp, vp]),
        "mgk_flat_dot": (i, [vp, C.c_long, vp, vp, C.POINTER(d), vp]),
        "mgk_stream_triad_f64": (i, [vp, C.c_long, vp, vp, vp, d, i, i, vp]),
        "mgk_flat_axpy": (i, [vp, C.c_long, d, vp, vp, vp]),
        "mgk_flat_scale": (i, [vp, C.c_long, d, vp, vp]),
        "mgk_flat_fill": (i, [vp, C.c_long, d, vp, vp]),
        "mgk_flat_aypx": (i, [vp, C.c_long, d, vp, vp, vp]),
        "mgk_flat_axpbypcz": (i, [vp, C.c_long, d, d, d, vp, vp, vp, vp]),
        "mgk_flat_pointwise_mult": (i, [vp, C.c_long, vp, vp, vp, vp]),
        "mgk_prolong_jacobi_f64": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, vp]),
        "mgk_prolong_jacobi_f32": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, vp]),
        "mgk_prolong_jacobi_range_f64": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, i, i, vp]),
        "mgk_prolong_jacobi_range_f32": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, i, i, vp]),
        "mgk_residual_restrict_range_f64": (i, [vp, G, G, c_dp, vp, vp, vp, i, i, vp]),
        "mgk_residual_restrict_range_f32": (i, [vp, G, G, c_dp, vp, vp, vp, i, i, vp]),
        "mgk_residual_restrict_slab_f64": (i, [vp, G, G, G, c_dp, vp, vp, vp, i, vp, i, i, vp]),
        "mgk_residual_restrict_slab_f32": (i, [vp, G, G, G, c_dp, vp, vp, vp, i, vp, i, i, vp]),
        "mgk_residual_restrict_2d_f64": (i, [vp, G, G, c_dp, vp, vp, vp, vp, d, d, vp]),
        "mgk_sweep_residual_restrict_f64": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, vp, d, d, vp]),
        "mgk_sweep_residual_restrict_ok_f64": (i, [G, G]),
        "mgk_jacobi2_sumsq_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_jacobi2_sumsq_ok_f64": (i, [G]),
        "mgk_jacobi2_sumsq_mid_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_prolong_jacobi2_ok_f64": (i, [G, G]),
        "mgk_prolong_jacobi2_f64": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, vp]),
        "mgk_jacobi2_zero_ok_f64": (i, [G]),
        "mgk_jacobi2_zero_ok_f32": (i, [G]),
        "mgk_jacobi2_zero_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp]),
        "mgk_jacobi2_zero_f32": (i, [vp, G, c_dp, d, d, vp, vp, vp]),
        "mgk_jacobi2_2d_sumsq_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_jacobi3_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp]),
        "mgk_jacobi3_sumsq_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_jacobi3_2d_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp, vp, vp]),
        "mgk_jacobi3_2d_sumsq_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_jacobi3_2d_zero_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp, vp]),
        "mgk_jacobi3_2d_sumsq_store_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, vp, vp, vp, C.POINTER(d), vp]),
        "mgk_prolong_jacobi3_2d_f64": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, vp, vp, vp]),
        "mgk_sweep_residual_restrict_2d_f64": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, vp, d, d, vp]),
        "mgk_jacobi2_sumsq_slab_f64": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, i, i, i, i, i, C.POINTER(i), vp]),
        "mgk_jacobi2_sumsq_mid_slab_f64": (i, [vp, G, G, c_dp, d, d, vp, vp, vp, vp, i, i, i, i, i, C.POINTER(i), vp]),
        "mgk_prolong_jacobi2_slab_ok_f64": (i, [G, G, i]),
        "mgk_prolong_jacobi2_slab_f64": (i, [vp, G, G, G, G, c_dp, d, d, vp, vp, vp, vp, vp, vp, i, i, i, i, vp]),
        "mgk_sweep_residual_restrict_slab_ok_f64": (i, [G, G]),
        "mgk_sweep_residual_restrict_slab_f64": (i, [vp, G, G, G, c_dp, d, d, vp, vp, vp, vp, vp, vp, i, i, vp, i, i, vp]),
        "mgk_ctx_set_chunk_planes": (i, [vp, i]),
        "mgk_residual_f64_to_f32_jz": (i, [vp, G, G, c_dp, vp, vp, vp, vp, d, d, C.POINTER(d), vp]),
        "mgk_correct_residual_f64_f32_jz": (i, [vp, G, G, c_dp, vp, vp, vp, vp, vp, vp, d, d, C.POINTER(d), vp]),
        "mgk_tail_cycle_f64": (i, [vp, G, i, C.POINTER(i), c_dp, c_dp, d, i, i, vp, vp, vp]),
        "mgk_tail_cycle_cs_f64": (i, [vp, G, i, C.POINTER(i), c_dp, c_dp, C.POINTER(vp), C.POINTER(vp), d, d, i, i, vp, vp, vp]),
        "mgk_tail_cycle_f32": (i, [vp, G, i, C.POINTER(i), c_dp, c_dp, d, i, i, vp, vp, vp]),
        "mgk_tail_max_n": (i, [i]),
        "mgk_debug_tail_stamps": (None, [vp]),
        "mgk_jacobi_sumsq_range_f64": (i, [vp, G, c_dp, d, d, vp, vp, vp, i, i, i, C.POINTER(i), vp]),
        "mgk_partials_finish": (i, [vp, i, C.POINTER(d), vp]),
        "mgk_host_alloc": (i, [vp, C.POINTER(vp), sz]),
        "mgk_host_free": (i, [vp, vp]),
        "mgk_d2h_async": (i, [vp, vp, vp, sz, vp]),
        "mgk_h2d_async": (i, [vp, vp, vp, sz, vp]),
        "mgk_delay_us": (i, [vp, d, vp]),
        "mgk_paced_copy": (i, [vp, vp, vp, sz, d, i, vp]),
    }
    for name, (res, args) in S.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    return S


class Mgk:
    """Thin object wrapper: one context on one device."""

    def __init__(self, device=0):
        self.L = load_mgk()
        self.symbols = _sigs(self.L)
        self.ctx = C.c_void_p()
        self._chk(self.L.mgk_ctx_create(C.byref(self.ctx), device))

    def _chk(self, rc):
        if rc != 0:
            raise MgkError(f"mgk call failed: rc={rc}: {self.L.mgk_last_error().decode()}")

    def close(self):
        if self.ctx:
            self.L.mgk_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    # -- geometry / memory --
    def geom(self, dim, nx, ny=None, nz=None):
        g = Geom()
        ny = nx if ny is None else ny
        nz = (nx if dim == 3 else 1) if nz is None else nz
        self._chk(self.L.mgk_geom_init(C.byref(g), dim, nx, ny, nz))
        return g

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.mgk_malloc(self.ctx, C.byref(p), nbytes))
        return p

    def free(self, p):
        self._chk(self.L.mgk_free(self.ctx, p))

    def field(self, g):
        return self.alloc(8 * g.total)

    def sync(self):
        self._chk(self.L.mgk_sync(self.ctx, None))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        p = self.alloc(arr.nbytes)
        self._chk(self.L.mgk_h2d(self.ctx, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return p

    def download(self, p, n):
        out = np.empty(n, dtype=np.float64)
        self._chk(self.L.mgk_d2h(self.ctx, out.ctypes.data_as(C.c_void_p), p, out.nbytes))
        return out

    def to_field(self, g, compact):
        """compact lexicographic numpy array -> new padded device field"""
        tmp = self.upload(np.asarray(compact, dtype=np.float64).ravel())
        f = self.field(g)
        self._chk(self.L.mgk_pack_f64(self.ctx, C.byref(g), tmp, f, None))
        self.sync()
        self.free(tmp)
        return f

    def from_field(self, g, f):
        n = g.nx * g.ny * g.nz
        tmp = self.alloc(8 * n)
        self._chk(self.L.mgk_unpack_f64(self.ctx, C.byref(g), f, tmp, None))
        self.sync()
        out = self.download(tmp, n)
        self.free(tmp)
        return out

    # -- fp32 fields (mixed-precision cycle) --
    def geom32(self, n):
        g = Geom()
        self._chk(self.L.mgk_geom_init_f32(C.byref(g), 3, n, n, n))
        return g

    def to_field32(self, g32, compact):
        tmp = self.upload(np.asarray(compact, dtype=np.float64).ravel())
        f = self.alloc(4 * g32.total)
        self._chk(self.L.mgk_pack_f32(self.ctx, C.byref(g32), tmp, f, None))
        self.sync()
        self.free(tmp)
        return f

    def from_field32(self, g32, f):
        n = g32.nx * g32.ny * g32.nz
        tmp = self.alloc(8 * n)
        self._chk(self.L.mgk_unpack_f32(self.ctx, C.byref(g32), f, tmp, None))
        self.sync()
        out = self.download(tmp, n)
        self.free(tmp)
        return out.astype(np.float32)

    def raw_field(self, g, f):
        """whole padded allocation (ghosts included) as a flat numpy array"""
        return self.download(f, g.total)

    @staticmethod
    def coef(vals):
        return (C.c_double * len(vals))(*vals)
