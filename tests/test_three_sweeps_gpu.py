"""Round 3: three Richardson+Jacobi sweeps in one pass (csrc/mgk_kernels3.hip) against the CPU oracle on the same seeded inputs -- fields
bit for bit (np.array_equal), sums of squares to 1e-13.  2-D: mgk_jacobi3_2d_f64 / _sumsq / _zero, mgk_prolong_jacobi3_2d_f64, on
constant stencils (oracle/mgo.c) and on row tables (the canonical term order restated in numpy, as for the other row-table kernels).
Reference operations: KSPSolve with max_it = 3, src/solver.c:1531, :1536, :1542; MatMult(pro) + VecAXPY :1540-1541; VecNorm :1546."""
import ctypes as C

import numpy as np
import pytest

from oracle import Oracle
from test_kernels_gpu import _rt_apply, _rt_jacobi, _rt_tables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    return Oracle()


# 1 .. 7: smaller than a wave tile; 119 / 121 / 239 / 241: one and two tiles of 60 column pairs, exactly and one pair over; big levels
SIZES_2D = [1, 3, 5, 7, 31, 63, 119, 121, 239, 241, 255, 511, 1023, 2047, 4095]


@pytest.mark.parametrize("n", SIZES_2D)
def test_three_sweeps_2d_bit_exact(mgk, orc, n):
    rng = np.random.default_rng(31000 + n)
    q = float((n + 1) ** 2)
    As = [q, q, -4.0 * q, q, q]
    dinv = 1.0 / As[2]
    N = n * n
    u, b = rng.uniform(-1, 1, N), rng.uniform(-1, 1, N)
    g = mgk.geom(2, n)
    du, db, dout = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g)
    J = lambda x, zg=False: orc.jacobi(2, n, As, 0.8, b, x, zero_guess=zg)
    j3 = J(J(J(u)))
    z3 = J(J(J(np.zeros(N), True)))
    r0 = orc.residual(2, n, As, b, u)
    ss = C.c_double()
    L, coef = mgk.L, mgk.coef(As)
    # default choice; the marching form (50) and the short-chunk forms (51: 4 rows, 52: 8 rows) forced; chunk seams everywhere
    # (tuning variants: 58 marches the odd chunks downwards, 59 every chunk; neither is a default)
    for var, zc in ((-1, -1), (50, -1), (51, -1), (52, -1), (-1, 1), (-1, 5), (-1, 12), (-1, 64), (58, -1), (58, 5), (58, 12), (57, -1), (59, -1), (59, 5)):
        L.mgk_set_tuning(var, zc)
        mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(L.mgk_jacobi3_2d_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, None, None, db, du, dout, None))
        got = mgk.from_field(g, dout)
        assert np.array_equal(got, j3), f"variant={var} zc={zc}: three sweeps, max diff {np.abs(got - j3).max()}"
        raw = mgk.raw_field(g, dout)
        assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * max(np.abs(got).sum(), 1e-300)      # ghosts / padding stay zero
        mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(L.mgk_jacobi3_2d_sumsq_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, None, None, db, du, dout, C.byref(ss), None))
        assert np.array_equal(mgk.from_field(g, dout), j3), f"zc={zc}: three sweeps + norm"
        want = orc.sumsq(r0)
        assert abs(ss.value - want) <= 1e-13 * want, f"zc={zc}: norm {ss.value} vs {want}"
        mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(L.mgk_jacobi3_2d_zero_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, None, None, db, dout, None))
        got = mgk.from_field(g, dout)
        assert np.array_equal(got, z3), f"variant={var} zc={zc}: three sweeps from the zero guess, max diff {np.abs(got - z3).max()}"
    L.mgk_set_tuning(-1, -1)
    assert np.array_equal(mgk.from_field(g, du), u) and np.array_equal(mgk.from_field(g, db), b)
    # in place on u is refused, as for the other multi-sweep passes
    assert L.mgk_jacobi3_2d_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, None, None, db, du, du, None) != 0
    for p in (du, db, dout):
        mgk.free(p)


@pytest.mark.parametrize("nf", [3, 7, 63, 239, 243, 247, 255, 1023, 2047, 4095])   # (nf = 2 nc + 1 with nc odd)
def test_prolongation_and_three_sweeps_2d_bit_exact(mgk, orc, nf):
    rng = np.random.default_rng(32000 + nf)
    nc = (nf - 1) // 2
    q = float((nf + 1) ** 2)
    As = [q, q, -4.0 * q, q, q]
    dinv = 1.0 / As[2]
    u, b, uc = rng.uniform(-1, 1, nf * nf), rng.uniform(-1, 1, nf * nf), rng.uniform(-1, 1, nc * nc)
    gf, gc = mgk.geom(2, nf), mgk.geom(2, nc)
    du, db, duc, dout = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.to_field(gc, uc), mgk.field(gf)
    x = orc.prolong_add(2, nf, uc, u)
    for _ in range(3):
        x = orc.jacobi(2, nf, As, 0.8, b, x)
    for var, zc in ((-1, -1), (50, -1), (51, -1), (52, -1), (-1, 1), (-1, 7), (-1, 64), (58, -1), (58, 7), (58, 12), (59, 7)):
        mgk.L.mgk_set_tuning(var, zc)
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * gf.total, None))
        mgk._chk(mgk.L.mgk_prolong_jacobi3_2d_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, None, None, db, duc, du, dout, None))
        got = mgk.from_field(gf, dout)
        assert np.array_equal(got, x), f"variant={var} zc={zc}: max diff {np.abs(got - x).max()}"
        raw = mgk.raw_field(gf, dout)
        assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * np.abs(got).sum()
    mgk.L.mgk_set_tuning(-1, -1)
    assert np.array_equal(mgk.from_field(gf, du), u) and np.array_equal(mgk.from_field(gc, duc), uc)
    for p in (du, db, duc, dout):
        mgk.free(p)


@pytest.mark.parametrize("n", [3, 7, 63, 255, 509, 1023, 2047])
def test_three_sweeps_2d_on_row_tables(mgk, orc, n):
    """the same four entry points with per-row coefficient tables (stretched meshes): against the canonical term order in numpy"""
    rng = np.random.default_rng(33000 + n)
    ct, dt = _rt_tables(rng, n)
    u, b = rng.uniform(-1, 1, (n, n)), rng.uniform(-1, 1, (n, n))
    g = mgk.geom(2, n)
    du, db, dout = mgk.to_field(g, u.ravel()), mgk.to_field(g, b.ravel()), mgk.field(g)
    dct, ddt = mgk.upload(ct.ravel()), mgk.upload(dt)
    J = lambda x: _rt_jacobi(ct, b, x, 0.8)
    j3 = J(J(J(u)))
    z1 = 0.8 * (b * dt[:, None])
    z3 = J(J(z1))
    rr = b - _rt_apply(ct, u)
    ss = C.c_double()
    L = mgk.L
    for var, zc in ((-1, -1), (50, -1), (51, -1), (52, -1), (-1, 3), (-1, 16), (58, -1), (58, 16), (59, 16)):
        L.mgk_set_tuning(var, zc)
        mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(L.mgk_jacobi3_2d_sumsq_f64(mgk.ctx, C.byref(g), None, 1.0, 0.8, dct, ddt, db, du, dout, C.byref(ss), None))
        assert np.array_equal(mgk.from_field(g, dout).reshape(n, n), j3), f"zc={zc}"
        assert abs(ss.value - float((rr * rr).sum())) <= 1e-12 * float((rr * rr).sum())
        mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(L.mgk_jacobi3_2d_f64(mgk.ctx, C.byref(g), None, 1.0, 0.8, dct, ddt, db, du, dout, None))
        assert np.array_equal(mgk.from_field(g, dout).reshape(n, n), j3), f"zc={zc}"
        mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(L.mgk_jacobi3_2d_zero_f64(mgk.ctx, C.byref(g), None, 1.0, 0.8, dct, ddt, db, dout, None))
        assert np.array_equal(mgk.from_field(g, dout).reshape(n, n), z3), f"zc={zc}"
    if n >= 3 and ((n - 1) // 2) % 2 == 1:                 # (a coarse grid exists: nc odd)
        nc = (n - 1) // 2
        uc = rng.uniform(-1, 1, nc * nc)
        gc = mgk.geom(2, nc)
        duc = mgk.to_field(gc, uc)
        x = orc.prolong_add(2, n, uc, u.ravel()).reshape(n, n)
        want = J(J(J(x)))
        for var, zc in ((-1, -1), (50, -1), (51, -1), (52, -1), (-1, 5), (58, 5), (58, -1)):
            L.mgk_set_tuning(var, zc)
            mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
            mgk._chk(L.mgk_prolong_jacobi3_2d_f64(mgk.ctx, C.byref(g), C.byref(gc), None, 1.0, 0.8, dct, ddt, db, duc, du, dout, None))
            assert np.array_equal(mgk.from_field(g, dout).reshape(n, n), want), f"zc={zc}"
        mgk.free(duc)
    L.mgk_set_tuning(-1, -1)
    for p in (du, db, dout, dct, ddt):
        mgk.free(p)


@pytest.mark.parametrize("n", [7, 255, 1023, 2047])
def test_three_sweeps_with_stored_residual_2d(mgk, orc, n):
    """mgk_jacobi3_2d_sumsq_store_f64 (the drop-in's KSPBuildResidual + VecNorm + the next KSPSolve's three sweeps): r is the residual
    kernel's, unew three sweeps, bit for bit; the sum is ||r||^2; ghosts and padding of both outputs stay zero"""
    rng = np.random.default_rng(34000 + n)
    q = float((n + 1) ** 2)
    As = [q, q, -4.0 * q, q, q]
    dinv = 1.0 / As[2]
    u, b = rng.uniform(-1, 1, n * n), rng.uniform(-1, 1, n * n)
    g = mgk.geom(2, n)
    du, db, dout, dr = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g), mgk.field(g)
    res = orc.residual(2, n, As, b, u)
    j3 = u
    for _ in range(3):
        j3 = orc.jacobi(2, n, As, 0.8, b, j3)
    ss = C.c_double()
    for var in (-1, 50, 51, 52):
        mgk.L.mgk_set_tuning(var, -1)
        for f in (dout, dr):
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, f, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_jacobi3_2d_sumsq_store_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, None, None, db, du, dout, dr, C.byref(ss), None))
        assert np.array_equal(mgk.from_field(g, dr), res), f"variant={var}: stored residual"
        assert np.array_equal(mgk.from_field(g, dout), j3), f"variant={var}: three sweeps"
        assert abs(ss.value - orc.sumsq(res)) <= 1e-13 * orc.sumsq(res)
        for f, v in ((dr, res), (dout, j3)):
            raw = mgk.raw_field(g, f)
            assert abs(np.abs(raw).sum() - np.abs(v).sum()) <= 1e-9 * np.abs(v).sum()
    mgk.L.mgk_set_tuning(-1, -1)
    assert mgk.L.mgk_jacobi3_2d_sumsq_store_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, None, None, db, du, dout, dout, C.byref(ss), None) != 0
    for p in (du, db, dout, dr):
        mgk.free(p)


# ---- 3-D ----
@pytest.mark.parametrize("nx,ny,nz", [(1, 1, 1), (3, 3, 3), (7, 7, 7), (31, 31, 31), (63, 63, 63), (119, 9, 11), (121, 5, 7), (127, 127, 127),
                                      (239, 13, 9), (255, 255, 33), (511, 31, 17), (1023, 1023, 9), (1023, 21, 40)])
def test_three_sweeps_3d_bit_exact(mgk, orc, nx, ny, nz):
    """mgk_jacobi3_f64 / _sumsq_f64 == three sweeps of the oracle (cubes) / of mgk_jacobi_f64, itself pinned to the oracle (boxes: the
    oracle's stencil operators want nx = ny), bit for bit; tile heights 2 / 3 / 4 and several z chunkings"""
    rng = np.random.default_rng(35000 + nx + 7 * ny + 13 * nz)
    q = float((nx + 1) ** 2)
    As = [q, q, q, -6.0 * q, q, q, q]
    dinv = 1.0 / As[3]
    N = nx * ny * nz
    u, b = rng.uniform(-1, 1, N), rng.uniform(-1, 1, N)
    g = mgk.geom(3, nx, ny, nz)
    du, db, dout, d1, d2 = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g), mgk.field(g), mgk.field(g)
    L, coef = mgk.L, mgk.coef(As)
    if nx == ny:
        J = lambda x: orc.jacobi(3, nx, As, 0.8, b, x, nz=nz)
        want = J(J(J(u)))
        r0 = orc.residual(3, nx, As, b, u, nz=nz)
    else:
        mgk._chk(L.mgk_jacobi_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, db, du, d1, None))
        mgk._chk(L.mgk_jacobi_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, db, d1, d2, None))
        mgk._chk(L.mgk_jacobi_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, db, d2, d1, None))
        want = mgk.from_field(g, d1)
        mgk._chk(L.mgk_residual_f64(mgk.ctx, C.byref(g), coef, db, du, d2, None))
        r0 = mgk.from_field(g, d2)
    ss = C.c_double()
    for var, zc in ((-1, -1), (62, -1), (63, -1), (64, -1), (-1, 1), (-1, 5), (63, 7), (53, 16)):      # 64: the row-by-row form
        L.mgk_set_tuning(var, zc)
        mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(L.mgk_jacobi3_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, db, du, dout, None))
        got = mgk.from_field(g, dout)
        assert np.array_equal(got, want), f"variant={var} zc={zc}: max diff {np.abs(got - want).max()}"
        raw = mgk.raw_field(g, dout)
        assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * max(np.abs(got).sum(), 1e-300)
        mgk._chk(L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(L.mgk_jacobi3_sumsq_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, db, du, dout, C.byref(ss), None))
        assert np.array_equal(mgk.from_field(g, dout), want), f"variant={var} zc={zc}: with the norm"
        assert abs(ss.value - orc.sumsq(r0)) <= 1e-13 * orc.sumsq(r0)
    L.mgk_set_tuning(-1, -1)
    assert np.array_equal(mgk.from_field(g, du), u)
    for p in (du, db, dout, d1, d2):
        mgk.free(p)
