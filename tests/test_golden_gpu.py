"""GPU parity against the COMMITTED golden vectors (tests/golden/vcycle_golden.npz), without the oracle.

The vectors come from tests/golden/make_golden.py, a scipy.sparse restatement of the reference cycle
(src/solver.c:1414-1575) run in the development container; see that file for what they pin and what
they cannot (PETSc is unavailable: parity is unpinned by the reference itself).
Tolerance: BASELINE.json north_star, 1e-12 relative on fp64 residual norms; fields are expected and
observed bit-identical because the kernels keep the ascending-column, unfused arithmetic."""
import ctypes as C

import numpy as np
import pytest

from golden_cases import GOLD, CYCLE_KEYS, MESH_KEYS, cycle_case, mesh_case

pytestmark = pytest.mark.gpu
RTOL = 1e-12


@pytest.mark.parametrize("key", CYCLE_KEYS)
def test_solver_reproduces_committed_cycle_vectors(key):
    from multigrid_petsc_amd.solver import Solver
    g = cycle_case(key)
    s = Solver(g["dim"], g["npts"], g["levels"], v=(g["v0"], g["v1"]), maxiter=g["maxiter"], scale=g["scale"])
    s.set_rhs_problem()
    it = s.solve()
    assert it == g["iters"]
    assert abs(s.bnorm - g["bnorm"]) <= RTOL * g["bnorm"]
    rn = s.rnorm
    assert rn.shape == g["rnorm"].shape
    assert np.abs(rn - g["rnorm"]).max() <= RTOL * g["rnorm"][0]
    assert np.abs(rn / g["rnorm"] - 1).max() <= 1e-9
    u = s.solution()
    assert np.abs(u - g["u"]).max() <= RTOL * np.abs(g["u"]).max()
    assert np.abs(np.asarray(s.error_norms()) / g["err"] - 1).max() <= 1e-11
    s.close()


def _coef(dim, n):
    q = float((n + 1) ** 2)                      # 1/h^2, h = 1/(n+1): src/matbuild.c:99-104, src/problem.c:3-22
    return [q] * dim + [-2.0 * dim * q] + [q] * dim


@pytest.mark.parametrize("dim,nf", [(2, 31), (3, 15)])
def test_kernels_reproduce_committed_operator_vectors(mgk, dim, nf):
    x, xc, y = GOLD["xfer_d%d_fine" % dim], GOLD["xfer_d%d_coarse" % dim], GOLD["xfer_d%d_base" % dim]
    nc = (nf - 1) // 2
    gf, gc = mgk.geom(dim, nf), mgk.geom(dim, nc)
    coef = mgk.coef(_coef(dim, nf))
    dx, dxc, dy, dout, dc = mgk.to_field(gf, x), mgk.to_field(gc, xc), mgk.to_field(gf, y), mgk.field(gf), mgk.field(gc)
    L = mgk.L
    mgk._chk(L.mgk_restrict_fw_f64(mgk.ctx, C.byref(gf), C.byref(gc), dx, dc, None))
    assert np.array_equal(mgk.from_field(gc, dc), GOLD["xfer_d%d_restricted" % dim])
    mgk._chk(L.mgk_apply_f64(mgk.ctx, C.byref(gf), coef, dx, dout, None))
    assert np.array_equal(mgk.from_field(gf, dout), GOLD["xfer_d%d_applied" % dim])
    mgk._chk(L.mgk_residual_f64(mgk.ctx, C.byref(gf), coef, dy, dx, dout, None))
    assert np.array_equal(mgk.from_field(gf, dout), GOLD["xfer_d%d_residual" % dim])
    if dim == 3:
        want = GOLD["xfer_d3_residual"]          # fused residual + restriction == restriction of the residual vector
        mgk._chk(L.mgk_restrict_fw_f64(mgk.ctx, C.byref(gf), C.byref(gc), dout, dc, None))
        two_step = mgk.from_field(gc, dc)
        mgk._chk(L.mgk_memset0(mgk.ctx, dc, 8 * gc.total, None))
        mgk._chk(L.mgk_residual_restrict_f64(mgk.ctx, C.byref(gf), C.byref(gc), coef, dy, dx, dc, None))
        assert np.array_equal(mgk.from_field(gc, dc), two_step) and want.size == nf ** 3
    mgk._chk(L.mgk_prolong_add_f64(mgk.ctx, C.byref(gf), C.byref(gc), dxc, dy, None))
    assert np.array_equal(mgk.from_field(gf, dy), GOLD["xfer_d%d_prolonged" % dim])
    for p in (dx, dxc, dy, dout, dc):
        mgk.free(p)


@pytest.mark.parametrize("dim,npts", [(2, 17), (2, 33), (2, 129), (3, 9), (3, 17), (3, 33)])
def test_rhs_fill_reproduces_committed_vectors(dim, npts):
    from multigrid_petsc_amd.solver import Solver
    s = Solver(dim, npts, 1, v=(1, 1), maxiter=0, scale=1.0)
    s.set_rhs_problem()
    assert s.solve() == 0
    want = GOLD["b0_d%d_n%d" % (dim, npts)]
    wn = float(np.sqrt(np.dot(want, want)))
    assert abs(s.bnorm - wn) <= RTOL * wn and abs(s.rnorm[0] - wn) <= RTOL * wn      # u = 0: r = b
    s.close()


@pytest.mark.parametrize("key", MESH_KEYS)
def test_solver_reproduces_committed_stretched_mesh_vectors(key):
    from multigrid_petsc_amd.solver import Solver
    g = mesh_case(key)
    s = Solver(2, g["npts"], g["levels"], v=(g["v0"], g["v1"]), maxiter=g["maxiter"], scale=g["scale"], mesh=g["mesh"])
    s.set_rhs_problem()
    assert s.solve() == g["iters"]
    assert np.abs(s.rnorm - g["rnorm"]).max() <= 1e-11 * g["rnorm"][0]
    assert np.abs(s.solution() - g["u"]).max() <= 1e-11 * np.abs(g["u"]).max()
    assert np.abs(np.asarray(s.error_norms()) / g["err"] - 1).max() <= 1e-9
    s.close()
