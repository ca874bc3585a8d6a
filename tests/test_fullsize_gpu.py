"""Full-size checks at BASELINE.json's sizes (the oracle cannot run these in seconds), through size-independent
properties of the discretisation:
  * the converged solution reproduces the closed-form discrete-eigenvector error (known answer),
  * the V-cycle contraction and the cycle count are h-independent (same as on the oracle-checked small grids),
  * transfer operators reproduce constants / are adjoint (P = 2^d R^T),
  * a zero right-hand side with zero guess stays exactly zero; the residual of the exact discrete solution vanishes."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _kat(dim, npts):
    h = 1.0 / (npts - 1)
    return dim * math.pi ** 2 / ((4 * dim / h ** 2) * math.sin(math.pi * h / 2) ** 2) - 1.0


@pytest.mark.parametrize("dim,npts,levels,scale,precision", [
    (3, 1025, 10, 6.0 / 7.0, "fp64"),      # the headline configuration
    (3, 513, 9, 6.0 / 7.0, "fp64"),        # BASELINE config 3
    (2, 4097, 12, 0.8, "fp64"),            # BASELINE config 2
    (3, 1025, 10, 6.0 / 7.0, "mixed"),     # BASELINE config 5
])
def test_full_size_solve_known_answer_and_h_independence(dim, npts, levels, scale, precision):
    from multigrid_petsc_amd.solver import Solver
    s = Solver(dim, npts, levels, scale=scale, maxiter=40, precision=precision)
    s.set_rhs_problem()
    it = s.solve()
    rn = s.rnorm
    assert rn[-1] <= 1e-7 * s.bnorm < rn[-2]                     # stopping rule of src/solver.c:1530
    # h-independence: same count (+-1) and contraction as the oracle-checked grids of the same family
    small = Solver(dim, 129 if dim == 3 else 513, 7 if dim == 3 else 9, scale=scale, maxiter=40, precision=precision)
    small.set_rhs_problem()
    it_small = small.solve()
    assert abs(it - it_small) <= 1, (it, it_small)
    rho, rho_small = (rn[-1] / rn[1]) ** (1.0 / (it - 1)), (small.rnorm[-1] / small.rnorm[1]) ** (1.0 / (it_small - 1))
    assert abs(rho - rho_small) <= 0.03
    small.close()
    # known answer: max error against sin*sin(*sin) equals the discrete-eigenvector closed form
    e = s.error_norms()
    kat = _kat(dim, npts)
    assert abs(e[0] - kat) <= 5e-7 + 1e-3 * kat, (e[0], kat)
    s.close()


def test_full_size_operator_properties(mgk):
    L = mgk.L
    n = 1023
    g = mgk.geom(3, n)
    nc = (n - 1) // 2
    gc = mgk.geom(3, nc)
    one_f, one_c = mgk.upload(np.ones(n)), mgk.upload(np.ones(nc))
    a, b, out, ca, cb = mgk.field(g), mgk.field(g), mgk.field(g), mgk.field(gc), mgk.field(gc)
    ss = C.c_double()
    h = 1.0 / (n + 1)
    c = 1.0 / (h * h)
    As = [c, c, c, -6 * c, c, c, c]
    coef, dinv = mgk.coef(As), 1.0 / (-6 * c)
    # (1) zero stays zero, exactly
    mgk._chk(L.mgk_jacobi_f64(mgk.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, a, out, None))
    mgk._chk(L.mgk_sumsq_f64(mgk.ctx, C.byref(g), out, C.byref(ss), None))
    assert ss.value == 0.0
    # (2) restriction of the constant 1: interior coarse values are exactly 1 (weights sum to 1)
    mgk._chk(L.mgk_fill_separable_f64(mgk.ctx, C.byref(g), one_f, one_f, one_f, a, None))
    mgk._chk(L.mgk_restrict_fw_f64(mgk.ctx, C.byref(g), C.byref(gc), a, ca, None))
    mgk._chk(L.mgk_sumsq_f64(mgk.ctx, C.byref(gc), ca, C.byref(ss), None))
    assert ss.value == float(nc) ** 3
    # (3) adjointness  <P x, y>_fine = 8 <x, R y>_coarse  for separable test vectors (P = 2^d R^T)
    rng = np.random.default_rng(1)
    xs = [mgk.upload(rng.uniform(0.5, 1.5, nc)) for _ in range(3)]
    ys = [mgk.upload(rng.uniform(0.5, 1.5, n)) for _ in range(3)]
    mgk._chk(L.mgk_fill_separable_f64(mgk.ctx, C.byref(gc), xs[0], xs[1], xs[2], cb, None))       # x (coarse)
    mgk._chk(L.mgk_fill_separable_f64(mgk.ctx, C.byref(g), ys[0], ys[1], ys[2], b, None))         # y (fine)
    mgk._chk(L.mgk_memset0(mgk.ctx, out, 8 * g.total, None))
    mgk._chk(L.mgk_prolong_add_f64(mgk.ctx, C.byref(g), C.byref(gc), cb, out, None))              # P x
    mgk._chk(L.mgk_restrict_fw_f64(mgk.ctx, C.byref(g), C.byref(gc), b, ca, None))                # R y
    lhs, rhs = C.c_double(), C.c_double()
    mgk._chk(L.mgk_flat_dot(mgk.ctx, g.total, out, b, C.byref(lhs), None))
    mgk._chk(L.mgk_flat_dot(mgk.ctx, gc.total, cb, ca, C.byref(rhs), None))
    assert abs(lhs.value - 8.0 * rhs.value) <= 1e-12 * abs(lhs.value)
    # (4) linearity of the residual kernel: r(b, 2u) - 2 r(b, u) = -b  =>  || r(0, 2u) - 2 r(0, u) || = 0 exactly
    mgk._chk(L.mgk_memset0(mgk.ctx, b, 8 * g.total, None))
    mgk._chk(L.mgk_residual_f64(mgk.ctx, C.byref(g), coef, b, a, out, None))                      # -A 1
    mgk._chk(L.mgk_flat_scale(mgk.ctx, g.total, 2.0, a, None))
    r2 = mgk.field(g)
    mgk._chk(L.mgk_residual_f64(mgk.ctx, C.byref(g), coef, b, a, r2, None))                       # -A 2
    mgk._chk(L.mgk_flat_axpy(mgk.ctx, g.total, -2.0, out, r2, None))
    mgk._chk(L.mgk_flat_dot(mgk.ctx, g.total, r2, r2, C.byref(ss), None))
    assert ss.value == 0.0                                                                        # scaling by 2 is exact
    for p in (one_f, one_c, a, b, out, ca, cb, r2, *xs, *ys):
        mgk.free(p)


@pytest.mark.parametrize("precision", ["fp64", "mixed"])
def test_full_size_fused_cycle_equals_unfused_cycle(precision):
    """At 1023^3 every fusion of the cycle (residual+norm, prolongation+sweep, residual+restriction, norm+next first sweep,
    mixed-precision correction+residual) must leave the fields bit for bit as the plain kernel-per-call cycle does:
    identical solution checksums, residual histories equal up to the summation order of the norm."""
    from multigrid_petsc_amd.solver import Solver
    out = {}
    # -1: the default cycle (all fusions, sweeps two per pass, register / shuffle kernels, two-sweep norm pass, sweep inside the restriction)
    for fuse in (0, 31, 63 | 256 | 512, 63 | 256 | 512 | 1024 | 2048, -1):
        s = Solver(3, 1025, 10, scale=6.0 / 7.0, maxiter=12, precision=precision, fuse=fuse)
        s.set_rhs_problem()
        s.cycles(5)
        s.sync()
        e = s.error_norms()             # max, sum, sqrt(sum of squares) of |u - exact|: three checksums of the field
        out[fuse] = (s.rnorm.copy(), e)
        s.close()
    for f in (31, 63 | 256 | 512, 63 | 256 | 512 | 1024 | 2048, -1):
        assert np.abs(out[0][0] / out[f][0] - 1).max() <= 1e-12
        assert out[0][1][0] == out[f][1][0]
        assert abs(out[0][1][1] - out[f][1][1]) <= 1e-13 * out[0][1][1] and abs(out[0][1][2] - out[f][1][2]) <= 1e-13 * out[0][1][2]
