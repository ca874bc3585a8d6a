"""ctypes binding of the CPU oracle (oracle/libmgo.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

# bound the oracle's OpenMP team: a GPU box shows every hardware thread of the host but a job only
# gets a share of them (16 per GPU)
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, os.cpu_count() or 1))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "libmgo.so")


class VcycleCfg(C.Structure):
    _fields_ = [("dim", C.c_int), ("npts", C.c_int), ("levels", C.c_int), ("v0", C.c_int), ("v1", C.c_int),
                ("maxiter", C.c_int), ("ksp_type", C.c_int), ("scale", C.c_double), ("emin", C.c_double),
                ("emax", C.c_double), ("use_csr", C.c_int), ("fixed_cycles", C.c_int), ("rtol", C.c_double), ("mesh", C.c_int)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    def __init__(self):
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = self.L = C.CDLL(_LIB)
        L.mgo_mesh_h.restype = C.c_double
        L.mgo_mesh_h.argtypes = [C.c_int, C.c_int]
        L.mgo_norm2.restype = C.c_double
        L.mgo_norm2.argtypes = [C.c_void_p, C.c_long]
        L.mgo_sumsq.restype = C.c_double
        L.mgo_sumsq.argtypes = [C.c_void_p, C.c_long]
        L.mgo_ffunc.restype = C.c_double
        L.mgo_ffunc.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        L.mgo_solfunc.restype = C.c_double
        L.mgo_solfunc.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        for f in ("mgo_build_A", "mgo_build_R", "mgo_build_P"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_int, C.c_int, C.c_int]
        L.mgo_csr_free.argtypes = [C.c_void_p]
        for f in ("mgo_csr_nrows", "mgo_csr_ncols", "mgo_csr_nnz"):
            getattr(L, f).restype = C.c_long
            getattr(L, f).argtypes = [C.c_void_p]
        L.mgo_csr_mult.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mgo_csr_row.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.mgo_csr_diag_inv.argtypes = [C.c_void_p, C.c_void_p]
        L.mgo_richardson_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.mgo_chebyshev_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.mgo_residual_csr.argtypes = [C.c_void_p] * 4
        L.mgo_st_apply.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.mgo_st_jacobi.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double] + [C.c_void_p] * 5 + [C.c_int]
        L.mgo_st_cheby_step.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_double] * 3 + [C.c_void_p]
        L.mgo_st_residual.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.mgo_st_restrict.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mgo_st_prolong_add.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.mgo_vcycle.restype = C.c_int
        L.mgo_vcycle.argtypes = [C.POINTER(VcycleCfg), C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mgo_level_stencil.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        L.mgo_rhs.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.mgo_error_norms.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.mgo_coords_uniform.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.mgo_get_ranges.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.mgo_mapping_2d.restype = C.c_int
        L.mgo_mapping_2d.argtypes = [C.c_int] * 6 + [C.c_void_p] * 3
        L.mgo_level_total_2d.restype = C.c_int
        L.mgo_level_total_2d.argtypes = [C.c_int] * 4
        L.mgo_grid_n.restype = C.c_int
        L.mgo_grid_n.argtypes = [C.c_int, C.c_int]
        L.mgo_grid_ids.restype = C.c_int
        L.mgo_grid_ids.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.mgo_restriction_stencil.argtypes = [C.c_void_p]
        L.mgo_prolongation_stencil.argtypes = [C.c_void_p]
        L.mgo_num_threads.restype = C.c_int
        L.mgo_error_norms_mesh.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.mgo_st_jacobi_f32.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.mgo_st_residual_f32.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.mgo_st_restrict_f32.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.mgo_st_prolong_add_f32.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.mgo_pcmg.restype = C.c_int
        L.mgo_pcmg.argtypes = [C.POINTER(VcycleCfg), C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mgo_icycle.restype = C.c_int
        L.mgo_icycle.argtypes = [C.POINTER(VcycleCfg), C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.mgo_vcycle_mixed.restype = C.c_int
        L.mgo_vcycle_mixed.argtypes = [C.POINTER(VcycleCfg), C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]

    # ---- conveniences ----
    def level_stencil(self, dim, npts, l):
        As = np.zeros(7)
        h = C.c_double()
        self.L.mgo_level_stencil(dim, npts, l, _p(As), C.byref(h))
        return As[:5 if dim == 2 else 7].copy(), h.value

    def rhs(self, dim, npts):
        b = np.zeros((npts - 2) ** dim)
        self.L.mgo_rhs(dim, npts, _p(b))
        return b

    def coords(self, npts):
        c = np.zeros(npts)
        self.L.mgo_coords_uniform(npts, 0, _p(c))
        return c

    def error_norms_mesh(self, npts, mesh, u):
        e = np.zeros(3)
        u = np.ascontiguousarray(u)
        self.L.mgo_error_norms_mesh(npts, mesh, _p(u), _p(e))
        return e

    def error_norms(self, dim, npts, u):
        e = np.zeros(3)
        u = np.ascontiguousarray(u)
        self.L.mgo_error_norms(dim, npts, _p(u), _p(e))
        return e

    def jacobi(self, dim, n, As, scale, b, u, zero_guess=False, nz=None, zlo=None, zhi=None):
        nz = (n if dim == 3 else 1) if nz is None else nz
        out = np.zeros_like(b)
        As = np.ascontiguousarray(As)
        self.L.mgo_st_jacobi(dim, n, nz, _p(As), scale, _p(b), _p(u), _p(zlo), _p(zhi), _p(out), int(zero_guess))
        return out

    def cheby_step(self, dim, n, As, b, pk, pkm1, ckm1, ck, cz, nz=None, zlo=None, zhi=None):
        nz = (n if dim == 3 else 1) if nz is None else nz
        out = np.zeros_like(b)
        As = np.ascontiguousarray(As)
        self.L.mgo_st_cheby_step(dim, n, nz, _p(As), _p(b), _p(pk), _p(zlo), _p(zhi), _p(pkm1), ckm1, ck, cz, _p(out))
        return out

    def residual(self, dim, n, As, b, u, nz=None, zlo=None, zhi=None):
        nz = (n if dim == 3 else 1) if nz is None else nz
        out = np.zeros_like(b)
        As = np.ascontiguousarray(As)
        self.L.mgo_st_residual(dim, n, nz, _p(As), _p(b), _p(u), _p(zlo), _p(zhi), _p(out))
        return out

    def apply(self, dim, n, As, x, nz=None, zlo=None, zhi=None):
        nz = (n if dim == 3 else 1) if nz is None else nz
        out = np.zeros_like(x)
        As = np.ascontiguousarray(As)
        self.L.mgo_st_apply(dim, n, nz, _p(As), _p(x), _p(zlo), _p(zhi), _p(out))
        return out

    def restrict(self, dim, nf, rf, nzf=None, nzc=None, fzhi=None):
        nc = (nf - 1) // 2
        nzf = (nf if dim == 3 else 1) if nzf is None else nzf
        nzc = (nc if dim == 3 else 1) if nzc is None else nzc
        out = np.zeros(nc * nc * (nzc if dim == 3 else 1))
        self.L.mgo_st_restrict(dim, nf, nzf, nzc, _p(rf), _p(fzhi), _p(out))
        return out

    def prolong_add(self, dim, nf, uc, uf, nzf=None, nzc=None, czlo=None, czhi=None):
        nc = (nf - 1) // 2
        nzf = (nf if dim == 3 else 1) if nzf is None else nzf
        nzc = (nc if dim == 3 else 1) if nzc is None else nzc
        out = np.array(uf, dtype=np.float64, copy=True)
        self.L.mgo_st_prolong_add(dim, nf, nzf, nzc, _p(uc), _p(czlo), _p(czhi), _p(out))
        return out

    def sumsq(self, x):
        x = np.ascontiguousarray(x)
        return self.L.mgo_sumsq(_p(x), x.size)

    def vcycle(self, dim, npts, levels, v0=3, v1=3, maxiter=1000, ksp_type=0, scale=1.0, emin=0.0, emax=0.0,
               use_csr=0, fixed_cycles=0, rtol=0.0, want_u=True, mesh=0):
        cfg = VcycleCfg(dim, npts, levels, v0, v1, maxiter, ksp_type, scale, emin, emax, use_csr, fixed_cycles, rtol, mesh)
        rn = np.zeros(max(maxiter, fixed_cycles) + 1)
        u = np.zeros((npts - 2) ** dim) if want_u else None
        bn, sec = C.c_double(), C.c_double()
        it = self.L.mgo_vcycle(C.byref(cfg), _p(rn), _p(u), C.byref(bn), C.byref(sec))
        return {"iters": it, "rnorm": rn[:it + 1].copy(), "u": u, "bnorm": bn.value, "seconds": sec.value}

    def pcmg(self, dim, npts, levels, v0=3, v1=3, maxiter=1000, ksp_type=0, scale=1.0, emin=0.0, emax=0.0, use_csr=0):
        """-cycle 8 restatement (outer Richardson + PCMG V-cycle), see oracle/mgo.c: mgo_pcmg"""
        cfg = VcycleCfg(dim, npts, levels, v0, v1, maxiter, ksp_type, scale, emin, emax, use_csr, 0, 0.0, 0)
        rn = np.zeros(maxiter + 1)
        u = np.zeros((npts - 2) ** dim)
        bn, sec = C.c_double(), C.c_double()
        it = self.L.mgo_pcmg(C.byref(cfg), _p(rn), _p(u), C.byref(bn), C.byref(sec))
        return {"iters": it, "rnorm": rn[:it + 1].copy(), "u": u, "bnorm": bn.value, "seconds": sec.value}

    def icycle(self, dim, npts, maxiter=1000, scale=1.0, use_csr=0):
        """-cycle 1, one grid: monitored Richardson + Jacobi (oracle/mgo.c: mgo_icycle)"""
        cfg = VcycleCfg(dim, npts, 1, 0, 0, maxiter, 0, scale, 0.0, 0.0, use_csr, 0, 0.0, 0)
        rn = np.zeros(maxiter + 1)
        u = np.zeros((npts - 2) ** dim)
        bn = C.c_double()
        it = self.L.mgo_icycle(C.byref(cfg), _p(rn), _p(u), C.byref(bn))
        return {"iters": it, "rnorm": rn[:it + 1].copy(), "u": u, "bnorm": bn.value}

    # ---- fp32 leg ----
    def jacobi32(self, n, As, scale, b, u, zero_guess=False):
        As32 = np.asarray(As, dtype=np.float32)
        dinv = np.float32(1.0 / As[3])
        out = np.zeros(n ** 3, dtype=np.float32)
        self.L.mgo_st_jacobi_f32(n, _p(As32), dinv, np.float32(scale), _p(b), _p(u), _p(out), int(zero_guess))
        return out

    def residual32(self, n, As, b, u):
        As32 = np.asarray(As, dtype=np.float32)
        out = np.zeros(n ** 3, dtype=np.float32)
        self.L.mgo_st_residual_f32(n, _p(As32), _p(b), _p(u), _p(out))
        return out

    def restrict32(self, nf, rf):
        nc = (nf - 1) // 2
        out = np.zeros(nc ** 3, dtype=np.float32)
        self.L.mgo_st_restrict_f32(nf, _p(rf), _p(out))
        return out

    def prolong_add32(self, nf, uc, uf):
        out = np.array(uf, dtype=np.float32, copy=True)
        self.L.mgo_st_prolong_add_f32(nf, _p(uc), _p(out))
        return out

    def vcycle_mixed(self, npts, levels, v0=3, v1=3, maxiter=100, scale=1.0, fixed_cycles=0):
        cfg = VcycleCfg(3, npts, levels, v0, v1, maxiter, 0, scale, 0.0, 0.0, 0, fixed_cycles, 0.0, 0)
        rn = np.zeros(max(maxiter, fixed_cycles) + 1)
        u = np.zeros((npts - 2) ** 3)
        bn, sec = C.c_double(), C.c_double()
        it = self.L.mgo_vcycle_mixed(C.byref(cfg), _p(rn), _p(u), C.byref(bn), C.byref(sec))
        return {"iters": it, "rnorm": rn[:it + 1].copy(), "u": u, "bnorm": bn.value, "seconds": sec.value}

    # CSR handles
    def build(self, which, dim, npts, l):
        return {"A": self.L.mgo_build_A, "R": self.L.mgo_build_R, "P": self.L.mgo_build_P}[which](dim, npts, l)

    def csr_mult(self, m, x):
        y = np.zeros(self.L.mgo_csr_nrows(m))
        x = np.ascontiguousarray(x)
        self.L.mgo_csr_mult(m, _p(x), _p(y))
        return y

    def csr_rows(self, m):
        rows = []
        cols = np.zeros(64, dtype=np.int32)
        vals = np.zeros(64)
        n = C.c_int()
        for r in range(self.L.mgo_csr_nrows(m)):
            self.L.mgo_csr_row(m, r, C.byref(n), _p(cols), _p(vals))
            rows.append((cols[:n.value].copy(), vals[:n.value].copy()))
        return rows
