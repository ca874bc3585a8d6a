"""ctypes binding of include/mgsolve.h -- the host-side V-cycle driver (C99) over the HIP kernels.

Names follow the reference driver (src/poisson.c:27-138): SetUp -> Assemble (set_rhs_problem) ->
Solve -> Postprocessing (error_norms, solution).  Plumbing only; all work happens in the C library.
"""
import ctypes as C

import numpy as np

from ._lib import load_mgpetsc


class MgConfig(C.Structure):
    _fields_ = [("dim", C.c_int), ("npts", C.c_int), ("levels", C.c_int), ("v", C.c_int * 2),
                ("maxiter", C.c_int), ("ksp_type", C.c_int), ("scale", C.c_double),
                ("emin", C.c_double), ("emax", C.c_double), ("rtol", C.c_double),
                ("device", C.c_int), ("precision", C.c_int), ("rank", C.c_int), ("nranks", C.c_int),
                ("dist_min_n", C.c_int), ("fuse", C.c_int), ("overlap", C.c_int), ("graph", C.c_int),
                ("pair_min_n", C.c_int), ("slab_chunk", C.c_int), ("mesh", C.c_int)]


class MgError(RuntimeError):
    pass


_KSP = {"richardson": 0, "chebyshev": 1}


def _lib():
    L = load_mgpetsc()
    if getattr(L, "_mg_sigs", False):
        return L
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    L.mg_config_default.argtypes = [C.POINTER(MgConfig)]
    L.mg_solver_create.restype = i
    L.mg_solver_create.argtypes = [C.POINTER(vp), C.POINTER(MgConfig), vp]
    L.mg_solver_destroy.argtypes = [vp]
    L.mg_last_error.restype = C.c_char_p
    for f in ("mg_solver_set_rhs_problem", "mg_solver_reset", "mg_solver_solve", "mg_solver_sync"):
        getattr(L, f).restype = i
        getattr(L, f).argtypes = [vp]
    L.mg_solver_set_rhs_host.restype = i
    L.mg_solver_set_rhs_host.argtypes = [vp, vp]
    L.mg_solver_cycles.restype = i
    L.mg_solver_cycles.argtypes = [vp, i]
    L.mg_solver_iterations.restype = i
    L.mg_solver_iterations.argtypes = [vp]
    L.mg_solver_bnorm.restype = d
    L.mg_solver_bnorm.argtypes = [vp]
    L.mg_solver_rnorm.restype = C.POINTER(d)
    L.mg_solver_rnorm.argtypes = [vp]
    L.mg_solver_solve_seconds.restype = d
    L.mg_solver_solve_seconds.argtypes = [vp]
    L.mg_solver_num_levels.restype = i
    L.mg_solver_num_levels.argtypes = [vp]
    L.mg_solver_level_n.restype = i
    L.mg_solver_level_n.argtypes = [vp, i]
    L.mg_solver_level_local_planes.restype = i
    L.mg_solver_level_local_planes.argtypes = [vp, i, C.POINTER(i)]
    L.mg_solver_local_unknowns.restype = C.c_long
    L.mg_solver_local_unknowns.argtypes = [vp]
    L.mg_solver_dof_updates_per_cycle.restype = d
    L.mg_solver_dof_updates_per_cycle.argtypes = [vp]
    L.mg_solver_get_solution.restype = i
    L.mg_solver_get_solution.argtypes = [vp, vp]
    L.mg_solver_error_norms.restype = i
    L.mg_solver_error_norms.argtypes = [vp, C.POINTER(d)]
    L.mg_solver_profile.restype = i
    L.mg_solver_profile.argtypes = [vp, i]
    L.mg_solver_profile_read.restype = i
    L.mg_solver_profile_read.argtypes = [vp, C.POINTER(d), C.POINTER(i)]
    L.mg_solver_profile_read_kind.restype = i
    L.mg_solver_profile_read_kind.argtypes = [vp, i, C.POINTER(d), C.POINTER(i)]
    L.mg_get_ranges.argtypes = [i, i, vp]
    L.mg_grid_n.restype = i
    L.mg_grid_n.argtypes = [i, i]
    L.mg_grid_to_global.restype = C.c_long
    L.mg_grid_to_global.argtypes = [i, i, i, i, i]
    L.mg_global_to_grid.argtypes = [i, i, C.c_long, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.mg_slab_range.restype = i
    L.mg_slab_range.argtypes = [i, i, i, i, i, C.POINTER(i), C.POINTER(i)]
    L._mg_sigs = True
    return L


def get_ranges(totaln, procs):
    r = np.zeros(procs + 1, dtype=np.int32)
    _lib().mg_get_ranges(totaln, procs, r.ctypes.data_as(C.c_void_p))
    return r


def slab_range(npts, levels_dist, level, rank, nranks):
    a, b = C.c_int(), C.c_int()
    rc = _lib().mg_slab_range(npts, levels_dist, level, rank, nranks, C.byref(a), C.byref(b))
    if rc:
        raise MgError(f"mg_slab_range rc={rc}")
    return a.value, b.value


class Solver:
    """One rank's multigrid solver (whole grid when nranks == 1)."""

    def __init__(self, dim, npts, levels, v=(3, 3), maxiter=100000, ksp_type="richardson", scale=1.0,
                 eigenvalues=(0.0, 0.0), rtol=1.0e-7, device=0, rank=0, nranks=1, comm=None,
                 dist_min_n=0, fuse=-1, overlap=-1, precision="fp64", graph=-1, pair_min_n=0, mesh=0, slab_chunk=-1):
        self.L = _lib()
        cfg = MgConfig()
        self.L.mg_config_default(C.byref(cfg))
        cfg.dim, cfg.npts, cfg.levels = dim, npts, levels
        cfg.v[0], cfg.v[1] = v
        cfg.maxiter = maxiter
        cfg.ksp_type = _KSP[ksp_type]
        cfg.scale = scale
        cfg.emin, cfg.emax = eigenvalues
        cfg.rtol = rtol
        cfg.device, cfg.rank, cfg.nranks = device, rank, nranks
        cfg.precision = {"fp64": 0, "mixed": 1}[precision]
        cfg.dist_min_n, cfg.fuse, cfg.overlap, cfg.graph = dist_min_n, fuse, overlap, graph
        cfg.pair_min_n = pair_min_n
        cfg.slab_chunk = slab_chunk
        cfg.mesh = mesh
        self.cfg = cfg
        self.h = C.c_void_p()
        self._chk(self.L.mg_solver_create(C.byref(self.h), C.byref(cfg), comm))

    def _chk(self, rc):
        if rc:
            raise MgError(f"rc={rc}: {self.L.mg_last_error().decode()}")

    def close(self):
        if self.h:
            self.L.mg_solver_destroy(self.h)
            self.h = C.c_void_p()

    def set_rhs_problem(self):
        self._chk(self.L.mg_solver_set_rhs_problem(self.h))

    def set_rhs(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert b.size == self.local_unknowns
        self._chk(self.L.mg_solver_set_rhs_host(self.h, b.ctypes.data_as(C.c_void_p)))

    def reset(self):
        self._chk(self.L.mg_solver_reset(self.h))

    def solve(self):
        self._chk(self.L.mg_solver_solve(self.h))
        return self.iterations

    def cycles(self, n):
        self._chk(self.L.mg_solver_cycles(self.h, n))

    def sync(self):
        self._chk(self.L.mg_solver_sync(self.h))

    @property
    def iterations(self):
        return self.L.mg_solver_iterations(self.h)

    @property
    def bnorm(self):
        return self.L.mg_solver_bnorm(self.h)

    @property
    def rnorm(self):
        """absolute residual norms rnorm[0..iterations]"""
        p = self.L.mg_solver_rnorm(self.h)
        return np.array([p[q] for q in range(self.iterations + 1)])

    @property
    def solve_seconds(self):
        return self.L.mg_solver_solve_seconds(self.h)

    @property
    def local_unknowns(self):
        return self.L.mg_solver_local_unknowns(self.h)

    @property
    def dof_updates_per_cycle(self):
        return self.L.mg_solver_dof_updates_per_cycle(self.h)

    def level_n(self, l):
        return self.L.mg_solver_level_n(self.h, l)

    def level_planes(self, l):
        z0 = C.c_int()
        nz = self.L.mg_solver_level_local_planes(self.h, l, C.byref(z0))
        return z0.value, nz

    def solution(self):
        u = np.empty(self.local_unknowns)
        self._chk(self.L.mg_solver_get_solution(self.h, u.ctypes.data_as(C.c_void_p)))
        return u

    def error_norms(self):
        e = (C.c_double * 3)()
        self._chk(self.L.mg_solver_error_norms(self.h, e))
        return np.array(list(e))

    def profile(self, on=True):
        self._chk(self.L.mg_solver_profile(self.h, int(on)))

    def profile_read(self, kind=0):
        """(total ms, launches) of the fine-level launches timed since profile(True): kind 0 plain sweeps,
        kind 1 two-sweeps-in-one-pass launches"""
        ms, n = C.c_double(), C.c_int()
        self._chk(self.L.mg_solver_profile_read_kind(self.h, kind, C.byref(ms), C.byref(n)))
        return ms.value, n.value
