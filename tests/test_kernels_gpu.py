"""GPU parity: every HIP kernel (through the C ABI, include/mgk.h) against the CPU oracle on the same
seeded inputs.  Field values must be BIT-IDENTICAL (the kernels implement the canonical arithmetic of
oracle/mgo.c: ascending-column sums, no FMA); reductions (sums of squares) agree to 1e-13 relative
(summation order is not part of the contract; north_star tolerance for fp64 residual norms is 1e-12)."""
import ctypes as C

import numpy as np
import pytest

from oracle import Oracle

pytestmark = pytest.mark.gpu
RED_RTOL = 1e-13


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def _rand(rng, n):
    return rng.uniform(-1.0, 1.0, n)


def _stencil(orc, dim, n):
    # coefficients of the level whose grid has n unknowns per side: npts = n + 2, level 0
    return orc.level_stencil(dim, n + 2, 0)[0]


CASES_3D = [(3, n, v) for n in (1, 3, 7, 31) for v in (0, 2)] + [(3, 127, 34), (3, 255, 34), (3, 255, -1), (3, 63, 34)] + \
           [(3, 63, v) for v in (0, 1, 2, 3, 6, 9, 12, 13)] + [(3, 127, v) for v in (1, 2, 3, 6, 9, 12, 13)]
CASES_2D = [(2, n, v) for n in (1, 3, 15, 127) for v in (0, 1)] + [(2, 255, v) for v in (0, 1, 2)] + \
           [(2, 1023, 2), (2, 2047, 2)]


@pytest.mark.parametrize("dim,n,variant", CASES_3D + CASES_2D)
def test_stencil_modes_bit_exact(mgk, orc, dim, n, variant):
    rng = np.random.default_rng(1000 * dim + n + variant)
    N = n ** dim
    As = _stencil(orc, dim, n)
    dinv = 1.0 / As[3 if dim == 3 else 2]
    u, b, pm = _rand(rng, N), _rand(rng, N), _rand(rng, N)
    g = mgk.geom(dim, n)
    du, db, dpm, dout = mgk.to_field(g, u), mgk.to_field(g, b), mgk.to_field(g, pm), mgk.field(g)
    coef = mgk.coef(As)
    L = mgk.L
    for zchunk in (-1, 5):
        L.mgk_set_tuning(variant, zchunk)
        # Jacobi sweep, two scales
        for scale in (1.0, 0.8):
            mgk._chk(L.mgk_jacobi_f64(mgk.ctx, C.byref(g), coef, dinv, scale, db, du, dout, None))
            got = mgk.from_field(g, dout)
            want = orc.jacobi(dim, n, As, scale, b, u)
            assert np.array_equal(got, want), f"jacobi scale={scale} maxdiff={np.abs(got - want).max()}"
        # residual
        mgk._chk(L.mgk_residual_f64(mgk.ctx, C.byref(g), coef, db, du, dout, None))
        got = mgk.from_field(g, dout)
        want = orc.residual(dim, n, As, b, u)
        assert np.array_equal(got, want)
        # fused residual + sum of squares
        ss = C.c_double()
        mgk._chk(L.mgk_residual_sumsq_f64(mgk.ctx, C.byref(g), coef, db, du, C.byref(ss), None))
        ref = orc.sumsq(want)
        assert abs(ss.value - ref) <= RED_RTOL * ref
        # Chebyshev recurrence step
        ck1, ck, cz = -0.37, 1.37, 0.21
        mgk._chk(L.mgk_cheby_f64(mgk.ctx, C.byref(g), coef, dinv, ck1, ck, cz, db, du, dpm, dout, None))
        got = mgk.from_field(g, dout)
        want = orc.cheby_step(dim, n, As, b, u, pm, ck1, ck, cz)
        assert np.array_equal(got, want)
    L.mgk_set_tuning(-1, -1)
    # ghosts of the output stay zero (Dirichlet ring is never written)
    raw = mgk.raw_field(g, dout)
    inter = mgk.from_field(g, dout)
    assert abs(np.abs(raw).sum() - np.abs(inter).sum()) <= 1e-9 * max(1.0, np.abs(inter).sum())
    for p in (du, db, dpm, dout):
        mgk.free(p)


@pytest.mark.parametrize("dim,n", [(2, 1), (2, 7), (2, 255), (3, 1), (3, 7), (3, 63)])
def test_zero_guess_sweep_and_sumsq(mgk, orc, dim, n):
    rng = np.random.default_rng(7 + n)
    N = n ** dim
    As = _stencil(orc, dim, n)
    dinv = 1.0 / As[3 if dim == 3 else 2]
    b = _rand(rng, N)
    g = mgk.geom(dim, n)
    db, dout = mgk.to_field(g, b), mgk.field(g)
    mgk._chk(mgk.L.mgk_jacobi_zero_f64(mgk.ctx, C.byref(g), dinv, 0.8, db, dout, None))
    got = mgk.from_field(g, dout)
    want = orc.jacobi(dim, n, As, 0.8, b, np.zeros(N), zero_guess=True)
    assert np.array_equal(got, want)
    # identical to a full sweep from u = 0
    assert np.array_equal(want, orc.jacobi(dim, n, As, 0.8, b, np.zeros(N)))
    ss = C.c_double()
    mgk._chk(mgk.L.mgk_sumsq_f64(mgk.ctx, C.byref(g), db, C.byref(ss), None))
    ref = orc.sumsq(b)
    assert abs(ss.value - ref) <= RED_RTOL * ref
    mgk.free(db)
    mgk.free(dout)


@pytest.mark.parametrize("dim,nf", [(2, 3), (2, 7), (2, 127), (2, 1023), (3, 3), (3, 7), (3, 31), (3, 127)])
def test_transfer_bit_exact(mgk, orc, dim, nf):
    rng = np.random.default_rng(99 + nf)
    nc = (nf - 1) // 2
    rf, uc, uf = _rand(rng, nf ** dim), _rand(rng, nc ** dim), _rand(rng, nf ** dim)
    gf, gc = mgk.geom(dim, nf), mgk.geom(dim, nc)
    drf, duc, duf, dbc = mgk.to_field(gf, rf), mgk.to_field(gc, uc), mgk.to_field(gf, uf), mgk.field(gc)
    mgk._chk(mgk.L.mgk_restrict_fw_f64(mgk.ctx, C.byref(gf), C.byref(gc), drf, dbc, None))
    got = mgk.from_field(gc, dbc)
    want = orc.restrict(dim, nf, rf)
    assert np.array_equal(got, want)
    mgk._chk(mgk.L.mgk_prolong_add_f64(mgk.ctx, C.byref(gf), C.byref(gc), duc, duf, None))
    got = mgk.from_field(gf, duf)
    want = orc.prolong_add(dim, nf, uc, uf)
    assert np.array_equal(got, want)
    # ghosts untouched
    raw = mgk.raw_field(gf, duf)
    assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * np.abs(got).sum()
    for p in (drf, duc, duf, dbc):
        mgk.free(p)


def test_transfer_matches_assembled_matrices(orc):
    """matrix-free transfer == MatMult with the assembled res/pro (runs on the GPU box too, CPU only)"""
    rng = np.random.default_rng(5)
    for dim, npts in ((2, 17), (3, 9)):
        nf, nc = npts - 2, (npts - 3) // 2
        R, P = orc.build("R", dim, npts, 0), orc.build("P", dim, npts, 0)
        rf, uc = _rand(rng, nf ** dim), _rand(rng, nc ** dim)
        assert np.array_equal(orc.csr_mult(R, rf), orc.restrict(dim, nf, rf))
        assert np.array_equal(orc.csr_mult(P, uc), orc.prolong_add(dim, nf, uc, np.zeros(nf ** dim)))


@pytest.mark.parametrize("dim,npts", [(2, 33), (3, 17)])
def test_rhs_fill_and_error_sums(mgk, orc, dim, npts):
    n = npts - 2
    c = orc.coords(npts)
    PI = 3.14159265358979323846
    s = np.sin(PI * c[1:-1])
    cx = ((-2 * PI * PI) if dim == 2 else (-3 * PI * PI)) * s     # ((-d*PI)*PI)*sin(PI*x), left to right
    g = mgk.geom(dim, n)
    dcx, ds, db = mgk.upload(cx), mgk.upload(s), mgk.field(g)
    mgk._chk(mgk.L.mgk_fill_separable_f64(mgk.ctx, C.byref(g), dcx, ds, ds, db, None))
    got = mgk.from_field(g, db)
    want = orc.rhs(dim, npts)
    assert np.array_equal(got, want)
    # error sums against the exact solution
    rng = np.random.default_rng(3)
    u = _rand(rng, n ** dim)
    du = mgk.to_field(g, u)
    e = np.zeros(3)
    mgk._chk(mgk.L.mgk_error_sums_f64(mgk.ctx, C.byref(g), du, ds, ds, ds, e.ctypes.data_as(C.POINTER(C.c_double)), None))
    ref = orc.error_norms(dim, npts, u)
    assert e[0] == ref[0]
    assert abs(e[1] - ref[1]) <= 1e-12 * ref[1]
    assert abs(np.sqrt(e[2]) - ref[2]) <= 1e-12 * ref[2]
    for p in (dcx, ds, db, du):
        mgk.free(p)


@pytest.mark.parametrize("nf,variant", [(3, -1), (7, 0), (31, 1), (63, 2), (63, 6), (127, 3), (127, 12), (255, -1), (63, 13), (255, 13), (255, 9),
                                        (3, 31), (7, 31), (15, 34), (31, 31), (63, 31), (63, 33), (127, 31), (127, 34), (255, 31), (255, 35), (511, 31)])
def test_fused_prolong_jacobi_bit_exact(mgk, orc, nf, variant):
    """unew = Jacobi(u + P uc) in one pass == prolong_add followed by a sweep (src/solver.c:1540-1542)"""
    rng = np.random.default_rng(400 + nf)
    nc = (nf - 1) // 2
    As = _stencil(orc, 3, nf)
    dinv = 1.0 / As[3]
    u, b, uc = _rand(rng, nf ** 3), _rand(rng, nf ** 3), _rand(rng, max(nc, 1) ** 3)
    gf, gc = mgk.geom(3, nf), mgk.geom(3, nc)
    du, db, duc, dout = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.to_field(gc, uc), mgk.field(gf)
    want = orc.jacobi(3, nf, As, 0.8, b, orc.prolong_add(3, nf, uc, u))
    for zc in (-1, 3):
        mgk.L.mgk_set_tuning(variant, zc)
        mgk._chk(mgk.L.mgk_prolong_jacobi_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, duc, du, dout, None))
        got = mgk.from_field(gf, dout)
        assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"
    mgk.L.mgk_set_tuning(-1, -1)
    assert np.array_equal(mgk.from_field(gf, du), u)          # the input is left untouched
    for p in (du, db, duc, dout):
        mgk.free(p)


@pytest.mark.parametrize("nf", [3, 7, 15, 63, 127, 255])
def test_fused_residual_restrict_bit_exact(mgk, orc, nf):
    """b_c = R (b - A u) in one pass == residual followed by full weighting (src/solver.c:1534-1535)"""
    rng = np.random.default_rng(700 + nf)
    nc = (nf - 1) // 2
    As = _stencil(orc, 3, nf)
    u, b = _rand(rng, nf ** 3), _rand(rng, nf ** 3)
    gf, gc = mgk.geom(3, nf), mgk.geom(3, nc)
    du, db, dbc = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.field(gc)
    want = orc.restrict(3, nf, orc.residual(3, nf, As, b, u))
    for variant in (30, 31, 34):            # LDS-tile form, register / shuffle form (ds_bpermute / DPP lane shifts)
        for zc in (-1, 5):
            mgk.L.mgk_set_tuning(variant, zc)
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dbc, 8 * gc.total, None))
            mgk._chk(mgk.L.mgk_residual_restrict_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), db, du, dbc, None))
            got = mgk.from_field(gc, dbc)
            assert np.array_equal(got, want), f"variant={variant} zc={zc} max diff {np.abs(got - want).max()}"
            raw = mgk.raw_field(gc, dbc)
            assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * np.abs(got).sum()
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, dbc):
        mgk.free(p)


@pytest.mark.parametrize("dim,n,variant", [(3, 7, 0), (3, 31, 2), (3, 127, 1), (3, 255, 6), (2, 127, 0), (2, 1023, 2),
                                            (3, 127, 34), (3, 255, -1), (3, 255, 34), (3, 63, 34)])
def test_sweep_with_input_residual_norm_bit_exact(mgk, orc, dim, n, variant):
    """mgk_jacobi_sumsq_f64: the sweep is the plain Jacobi sweep, the sum is ||b - A u||^2 of the INPUT field"""
    rng = np.random.default_rng(4000 + n)
    As = _stencil(orc, dim, n)
    dinv = 1.0 / As[3 if dim == 3 else 2]
    u, b = _rand(rng, n ** dim), _rand(rng, n ** dim)
    g = mgk.geom(dim, n)
    du, db, dout = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g)
    ss = C.c_double()
    for zc in (-1, 5):
        mgk.L.mgk_set_tuning(variant, zc)
        mgk._chk(mgk.L.mgk_jacobi_sumsq_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, du, dout, C.byref(ss), None))
        assert np.array_equal(mgk.from_field(g, dout), orc.jacobi(dim, n, As, 0.8, b, u))
        want = orc.sumsq(orc.residual(dim, n, As, b, u))
        assert abs(ss.value - want) <= 1e-13 * want
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, dout):
        mgk.free(p)


def _slab_field(mgk, g, whole, n, z0):
    """padded fp64 field of the z-slab [z0, z0 + g.nz) of a whole n^3 grid, ghost planes taken from the neighbours"""
    w = whole.reshape(n, n, n)
    pad = np.zeros(g.total)
    for k in range(-1, g.nz + 1):
        kz = z0 + k
        if 0 <= kz < n:
            for i in range(n):
                o = g.org + k * g.plane + i * g.pitch
                pad[o:o + n] = w[kz, i]
    return mgk.upload(pad)


@pytest.mark.parametrize("n,cut", [(31, 7), (15, 1), (63, 20)])
def test_fused_residual_restrict_on_slabs_bit_exact(mgk, orc, n, cut):
    """two z-slabs (coarse planes [0, cut) and [cut, nc)): the inner slab's fused kernel leaves its last coarse plane partial,
    mgk_restrict_finish_f64 closes it with the neighbour's first residual plane; the result is the whole-grid restriction"""
    rng = np.random.default_rng(77 + n)
    nc = (n - 1) // 2
    As = _stencil(orc, 3, n)
    u, b = _rand(rng, n ** 3), _rand(rng, n ** 3)
    want = orc.restrict(3, n, orc.residual(3, n, As, b, u)).reshape(nc, nc, nc)
    L = mgk.L
    coef = mgk.coef(As)
    got = []
    # slab 1 (last rank) first: its residual plane 0 is what slab 0 needs as hi ghost
    g1, gc1 = mgk.geom(3, n, n, n - 2 * cut), mgk.geom(3, nc, nc, nc - cut)
    u1, b1, r1, bc1 = _slab_field(mgk, g1, u, n, 2 * cut), _slab_field(mgk, g1, b, n, 2 * cut), mgk.field(g1), mgk.field(gc1)
    mgk._chk(L.mgk_residual_range_f64(mgk.ctx, C.byref(g1), coef, b1, u1, r1, 0, 1, None))
    mgk._chk(L.mgk_residual_restrict_f64(mgk.ctx, C.byref(g1), C.byref(gc1), coef, b1, u1, bc1, None))
    r1_plane0 = mgk.from_field(g1, r1).reshape(g1.nz, n, n)[0]
    assert np.array_equal(r1_plane0, orc.residual(3, n, As, b, u).reshape(n, n, n)[2 * cut])
    # slab 0 (inner): partial + finish
    g0, gc0 = mgk.geom(3, n, n, 2 * cut), mgk.geom(3, nc, nc, cut)
    u0, b0, bc0 = _slab_field(mgk, g0, u, n, 0), _slab_field(mgk, g0, b, n, 0), mgk.field(gc0)
    rpad = np.zeros(g0.total)
    for i in range(n):
        o = g0.org + g0.nz * g0.plane + i * g0.pitch
        rpad[o:o + n] = r1_plane0[i]
    r0 = mgk.upload(rpad)
    mgk._chk(L.mgk_residual_restrict_f64(mgk.ctx, C.byref(g0), C.byref(gc0), coef, b0, u0, bc0, None))
    mgk._chk(L.mgk_restrict_finish_f64(mgk.ctx, C.byref(g0), C.byref(gc0), r0, bc0, None))
    got0 = mgk.from_field(gc0, bc0).reshape(cut, nc, nc)
    got1 = mgk.from_field(gc1, bc1).reshape(nc - cut, nc, nc)
    assert np.array_equal(got0, want[:cut]) and np.array_equal(got1, want[cut:])
    for p in (u1, b1, r1, bc1, u0, b0, bc0, r0):
        mgk.free(p)


@pytest.mark.parametrize("nf,variant", [(3, -1), (7, 0), (127, 0), (255, 1), (1023, 2), (1023, 0), (2047, 30), (3, 38), (7, 38), (127, 38), (243, 38),
                                        (255, 38), (1023, 38), (2047, -1), (4095, -1)])
def test_fused_prolong_jacobi_2d_bit_exact(mgk, orc, nf, variant):
    """(from 2047^2 on the independent-wave kernel k_pj2d runs; variant 30 keeps the LDS-tile kernel, 38 forces the waves)"""
    rng = np.random.default_rng(500 + nf)
    nc = (nf - 1) // 2
    As = _stencil(orc, 2, nf)
    dinv = 1.0 / As[2]
    u, b, uc = _rand(rng, nf ** 2), _rand(rng, nf ** 2), _rand(rng, nc ** 2)
    gf, gc = mgk.geom(2, nf), mgk.geom(2, nc)
    du, db, duc, dout = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.to_field(gc, uc), mgk.field(gf)
    want = orc.jacobi(2, nf, As, 0.8, b, orc.prolong_add(2, nf, uc, u))
    for zc in (-1, 3):
        mgk.L.mgk_set_tuning(variant, zc)
        mgk._chk(mgk.L.mgk_prolong_jacobi_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, duc, du, dout, None))
        got = mgk.from_field(gf, dout)
        assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, duc, dout):
        mgk.free(p)


@pytest.mark.parametrize("n", [1, 3, 5, 7, 15, 31, 63, 127, 255])
def test_two_sweeps_in_one_pass_bit_exact(mgk, orc, n):
    """mgk_jacobi2_f64 (temporal blocking, u' lives in LDS only) == two mgk_jacobi_f64 sweeps, bit for bit"""
    rng = np.random.default_rng(6000 + n)
    As = _stencil(orc, 3, n) if n > 2 else [float((n + 1) ** 2)] * 3 + [-6.0 * (n + 1) ** 2] + [float((n + 1) ** 2)] * 3
    dinv = 1.0 / As[3]
    u, b = _rand(rng, n ** 3), _rand(rng, n ** 3)
    g = mgk.geom(3, n)
    du, db, dout = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g)
    want = orc.jacobi(3, n, As, 0.8, b, orc.jacobi(3, n, As, 0.8, b, u))
    for var, zc in [(v, z) for v in (-1, 1, 2) for z in (-1, 8, 13)]:      # default choice, ring variant, one-barrier variant
        mgk.L.mgk_set_tuning(var, zc)
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_jacobi2_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, du, dout, None))
        got = mgk.from_field(g, dout)
        assert np.array_equal(got, want), f"variant={var} zc={zc} max diff {np.abs(got - want).max()}"
        raw = mgk.raw_field(g, dout)
        assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * np.abs(got).sum()      # ghosts / padding stay zero
    mgk.L.mgk_set_tuning(-1, -1)
    assert np.array_equal(mgk.from_field(g, du), u)
    for p in (du, db, dout):
        mgk.free(p)


@pytest.mark.parametrize("n", [1, 3, 7, 63, 127, 255, 507, 509, 511, 1023, 2047])
def test_two_sweeps_in_one_pass_2d_bit_exact(mgk, orc, n):
    rng = np.random.default_rng(8000 + n)
    q = float((n + 1) ** 2)
    As = [q, q, -4.0 * q, q, q]
    dinv = 1.0 / As[2]
    u, b = _rand(rng, n ** 2), _rand(rng, n ** 2)
    g = mgk.geom(2, n)
    du, db, dout = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g)
    want = orc.jacobi(2, n, As, 0.8, b, orc.jacobi(2, n, As, 0.8, b, u))
    for zc in (-1, 16, 37):
        mgk.L.mgk_set_tuning(-1, zc)
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_jacobi2_2d_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, du, dout, None))
        got = mgk.from_field(g, dout)
        assert np.array_equal(got, want), f"zc={zc} max diff {np.abs(got - want).max()}"
        raw = mgk.raw_field(g, dout)
        assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * np.abs(got).sum()
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, dout):
        mgk.free(p)


@pytest.mark.parametrize("nf", [7, 63, 255])
def test_fused_residual_restrict_with_coarse_first_sweep(mgk, orc, nf):
    """mgk_residual_restrict_jz_f64: b_c as before, plus the coarse level's first sweep from a zero guess (what
    mgk_jacobi_zero_f64 computes from b_c)"""
    rng = np.random.default_rng(9100 + nf)
    nc = (nf - 1) // 2
    As, Asc = _stencil(orc, 3, nf), _stencil(orc, 3, nc)
    u, b = _rand(rng, nf ** 3), _rand(rng, nf ** 3)
    gf, gc = mgk.geom(3, nf), mgk.geom(3, nc)
    du, db, dbc, duc = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.field(gc), mgk.field(gc)
    mgk._chk(mgk.L.mgk_memset0(mgk.ctx, duc, 8 * gc.total, None))
    dinvc = 1.0 / Asc[3]
    mgk._chk(mgk.L.mgk_residual_restrict_jz_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), db, du, dbc, duc, dinvc, 0.8, None))
    bc = orc.restrict(3, nf, orc.residual(3, nf, As, b, u))
    assert np.array_equal(mgk.from_field(gc, dbc), bc)
    assert np.array_equal(mgk.from_field(gc, duc), orc.jacobi(3, nc, Asc, 0.8, bc, np.zeros(nc ** 3), zero_guess=True))
    for p in (du, db, dbc, duc):
        mgk.free(p)


@pytest.mark.parametrize("nf,variant,cuts", [(31, 0, (2, 30)), (63, 2, (2, 62)), (63, 13, (3, 40)), (127, 9, (2, 126)), (255, 9, (2, 254)),
                                             (63, 6, (5, 6)), (31, 31, (2, 30)), (63, 33, (3, 40)), (63, 31, (5, 6)), (127, 31, (2, 126)),
                                             (255, 34, (2, 254)), (127, 35, (7, 8))])
def test_fused_prolong_jacobi_plane_ranges(mgk, orc, nf, variant, cuts):
    """mgk_prolong_jacobi_range_f64: the slab solver sweeps the inner planes while the ghost planes travel, then the boundary
    planes; any cut into plane ranges (even and odd starts) must give the whole-launch result bit for bit"""
    rng = np.random.default_rng(4400 + nf)
    nc = (nf - 1) // 2
    As = _stencil(orc, 3, nf)
    dinv = 1.0 / As[3]
    u, b, uc = _rand(rng, nf ** 3), _rand(rng, nf ** 3), _rand(rng, nc ** 3)
    gf, gc = mgk.geom(3, nf), mgk.geom(3, nc)
    du, db, duc, dout = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.to_field(gc, uc), mgk.field(gf)
    want = orc.jacobi(3, nf, As, 0.8, b, orc.prolong_add(3, nf, uc, u))
    a, bnd = cuts
    mgk.L.mgk_set_tuning(variant, -1)
    for z0, z1 in ((a, bnd), (0, a), (bnd, nf)):          # inner planes first, as the solver does
        mgk._chk(mgk.L.mgk_prolong_jacobi_range_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, duc, du, dout, z0, z1, None))
    mgk.L.mgk_set_tuning(-1, -1)
    got = mgk.from_field(gf, dout)
    assert np.array_equal(got, want), f"max diff {np.abs(got - want).max()}"
    assert mgk.L.mgk_prolong_jacobi_range_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, duc, du, dout, 3, 3, None) != 0
    for p in (du, db, duc, dout):
        mgk.free(p)


@pytest.mark.parametrize("nf", [15, 31, 63, 127])
def test_fused_residual_restrict_coarse_plane_ranges(mgk, orc, nf):
    """mgk_residual_restrict_range_f64 over [1, mid), [mid, nc-1), [0, 1), [nc-1, nc) -- the order of the slab solver -- equals
    the whole launch and the oracle"""
    rng = np.random.default_rng(4700 + nf)
    nc = (nf - 1) // 2
    As = _stencil(orc, 3, nf)
    u, b = _rand(rng, nf ** 3), _rand(rng, nf ** 3)
    gf, gc = mgk.geom(3, nf), mgk.geom(3, nc)
    du, db, dbc = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.field(gc)
    want = orc.restrict(3, nf, orc.residual(3, nf, As, b, u))
    mid = nc // 2
    for variant in (30, 31, 34):
        for zc in (-1, 4):
            mgk.L.mgk_set_tuning(variant, zc)
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dbc, 8 * gc.total, None))
            for k0, k1 in ((1, mid), (mid, nc - 1), (0, 1), (nc - 1, nc)):
                mgk._chk(mgk.L.mgk_residual_restrict_range_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), db, du, dbc, k0, k1, None))
            got = mgk.from_field(gc, dbc)
            assert np.array_equal(got, want), f"variant={variant} zc={zc} max diff {np.abs(got - want).max()}"
    mgk.L.mgk_set_tuning(-1, -1)
    assert mgk.L.mgk_residual_restrict_range_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), db, du, dbc, 2, 2, None) != 0
    for p in (du, db, dbc):
        mgk.free(p)


@pytest.mark.parametrize("n,variant", [(31, 0), (63, 2), (127, 6), (255, 12), (127, 34), (255, 34)])
def test_sweep_with_norm_over_plane_ranges(mgk, orc, n, variant):
    """mgk_jacobi_sumsq_range_f64 x3 + mgk_partials_finish: the sweep equals the plain sweep bit for bit, the sum is that of
    the whole launch to rounding (block partials of the three launches reduced in slot order)"""
    rng = np.random.default_rng(4900 + n)
    As = _stencil(orc, 3, n)
    dinv = 1.0 / As[3]
    u, b = _rand(rng, n ** 3), _rand(rng, n ** 3)
    g = mgk.geom(3, n)
    du, db, dout = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g)
    mgk.L.mgk_set_tuning(variant, -1)
    off, np_ = 0, C.c_int()
    for z0, z1 in ((1, n - 1), (0, 1), (n - 1, n)):
        mgk._chk(mgk.L.mgk_jacobi_sumsq_range_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, du, dout, z0, z1, off, C.byref(np_), None))
        off += np_.value
    ss = C.c_double()
    mgk._chk(mgk.L.mgk_partials_finish(mgk.ctx, off, C.byref(ss), None))
    mgk.L.mgk_set_tuning(-1, -1)
    assert np.array_equal(mgk.from_field(g, dout), orc.jacobi(3, n, As, 0.8, b, u))
    want = orc.sumsq(orc.residual(3, n, As, b, u))
    assert abs(ss.value - want) <= 1e-13 * want
    for p in (du, db, dout):
        mgk.free(p)


def test_pinned_async_copies_and_delay(mgk):
    """the pieces of the device-resident norm reduction: pinned landing area, stream-ordered copies, and the link-time
    stand-in of the phantom communicator (holds its stream for about the requested time)"""
    import time
    h = C.c_void_p()
    mgk._chk(mgk.L.mgk_host_alloc(mgk.ctx, C.byref(h), 64))
    src = (C.c_double * 8)(*[1.5 * q for q in range(8)])
    d = mgk.alloc(64)
    ms = mgk.L.mgk_stream_comm(mgk.ctx)
    C.memmove(h, src, 64)
    mgk._chk(mgk.L.mgk_h2d_async(mgk.ctx, d, h, 64, ms))
    C.memset(h, 0, 64)        # safe only after the copy: synchronise first
    mgk._chk(mgk.L.mgk_sync(mgk.ctx, ms))
    mgk._chk(mgk.L.mgk_h2d_async(mgk.ctx, d, C.cast(src, C.c_void_p), 64, ms))
    mgk._chk(mgk.L.mgk_d2h_async(mgk.ctx, h, d, 64, ms))
    mgk._chk(mgk.L.mgk_sync(mgk.ctx, ms))
    assert list((C.c_double * 8).from_address(h.value)) == list(src)
    mgk._chk(mgk.L.mgk_sync(mgk.ctx, None))
    t0 = time.perf_counter()
    mgk._chk(mgk.L.mgk_delay_us(mgk.ctx, 20000.0, ms))
    mgk._chk(mgk.L.mgk_sync(mgk.ctx, ms))
    dt = time.perf_counter() - t0
    assert 0.019 <= dt <= 0.2, dt
    assert mgk.L.mgk_delay_us(mgk.ctx, -1.0, ms) != 0
    mgk.free(d)
    mgk._chk(mgk.L.mgk_host_free(mgk.ctx, h))


@pytest.mark.parametrize("n,cut", [(31, 7), (15, 2), (63, 20), (63, 3)])
def test_fused_residual_restrict_on_slabs_single_exchange(mgk, orc, n, cut):
    """mgk_residual_restrict_slab_f64: the inner slab holds the upper slab's planes 0 (u's hi ghost) and 1 (hi ghost of the far
    field) and b's hi ghost plane, evaluates the residual of the plane above itself and completes its last coarse plane: the
    two slabs together give the whole-grid restriction bit for bit, with no partial plane and no finishing kernel"""
    rng = np.random.default_rng(177 + n)
    nc = (n - 1) // 2
    As = _stencil(orc, 3, n)
    u, b = _rand(rng, n ** 3), _rand(rng, n ** 3)
    want = orc.restrict(3, n, orc.residual(3, n, As, b, u)).reshape(nc, nc, nc)
    L, coef = mgk.L, mgk.coef(As)
    w = u.reshape(n, n, n)
    # inner slab: fine planes [0, 2 cut), coarse planes [0, cut)
    g0, gc0, gfar = mgk.geom(3, n, n, 2 * cut), mgk.geom(3, nc, nc, cut), mgk.geom(3, n, n, 2)
    u0, b0, bc0 = _slab_field(mgk, g0, u, n, 0), _slab_field(mgk, g0, b, n, 0), mgk.field(gc0)
    far = np.zeros(gfar.total)
    for i in range(n):                      # hi ghost plane of the far field <- plane 1 of the slab above (global plane 2 cut + 1)
        o = gfar.org + 2 * gfar.plane + i * gfar.pitch
        far[o:o + n] = w[2 * cut + 1, i]
    dfar = mgk.upload(far)
    L.mgk_set_tuning(31 if n % 2 and cut % 2 else 30, -1)       # both forms of the kernel across the parametrisation
    for k0, k1 in ((1, cut - 1), (0, 1), (cut - 1, cut)) if cut >= 3 else ((0, cut),):
        mgk._chk(L.mgk_residual_restrict_slab_f64(mgk.ctx, C.byref(g0), C.byref(gc0), C.byref(gfar), coef, b0, u0, dfar, 1, bc0, k0, k1, None))
    # last slab: fine planes [2 cut, n), no rank above
    g1, gc1 = mgk.geom(3, n, n, n - 2 * cut), mgk.geom(3, nc, nc, nc - cut)
    u1, b1, bc1 = _slab_field(mgk, g1, u, n, 2 * cut), _slab_field(mgk, g1, b, n, 2 * cut), mgk.field(gc1)
    mgk._chk(L.mgk_residual_restrict_slab_f64(mgk.ctx, C.byref(g1), C.byref(gc1), None, coef, b1, u1, None, 0, bc1, 0, nc - cut, None))
    L.mgk_set_tuning(-1, -1)
    got0 = mgk.from_field(gc0, bc0).reshape(cut, nc, nc)
    got1 = mgk.from_field(gc1, bc1).reshape(nc - cut, nc, nc)
    assert np.array_equal(got0, want[:cut]) and np.array_equal(got1, want[cut:])
    # a far field of the wrong shape is refused
    gbad = mgk.geom(3, n, n, 3)
    assert L.mgk_residual_restrict_slab_f64(mgk.ctx, C.byref(g0), C.byref(gc0), C.byref(gbad), coef, b0, u0, dfar, 1, bc0, 0, cut, None) != 0
    for p in (u0, b0, bc0, dfar, u1, b1, bc1):
        mgk.free(p)


@pytest.mark.parametrize("nf", [3, 7, 63, 127, 255, 1023, 2047])
def test_fused_residual_restrict_2d_bit_exact(mgk, orc, nf):
    """mgk_residual_restrict_2d_f64 == residual followed by full weighting (src/solver.c:1534-1535), and its optional second
    output == the zero-guess sweep of the coarse level, bit for bit; several y chunkings"""
    rng = np.random.default_rng(7100 + nf)
    nc = (nf - 1) // 2
    As = _stencil(orc, 2, nf)
    Ac = _stencil(orc, 2, nc)
    u, b = _rand(rng, nf ** 2), _rand(rng, nf ** 2)
    gf, gc = mgk.geom(2, nf), mgk.geom(2, nc)
    du, db, dbc, duc0 = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.field(gc), mgk.field(gc)
    want = orc.restrict(2, nf, orc.residual(2, nf, As, b, u))
    dinv_c = 1.0 / Ac[2]
    for var, zc in ((-1, -1), (-1, 5), (-1, 64), (55, -1), (56, -1)):          # 55 / 56: the marching / the short-chunk form forced (round 3)
        mgk.L.mgk_set_tuning(var, zc)
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dbc, 8 * gc.total, None))
        mgk._chk(mgk.L.mgk_residual_restrict_2d_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), db, du, dbc, duc0, dinv_c, 0.8, None))
        got = mgk.from_field(gc, dbc)
        assert np.array_equal(got, want), f"variant={var} zc={zc} max diff {np.abs(got - want).max()}"
        assert np.array_equal(mgk.from_field(gc, duc0), 0.8 * (want * dinv_c))
        raw = mgk.raw_field(gc, dbc)
        assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * max(np.abs(got).sum(), 1e-300)
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, dbc, duc0):
        mgk.free(p)


# ---- the fused kernels on row-table operators (2-D stretched meshes) against numpy in the canonical term order ----
def _rt_apply(ct, u):
    """A u for the row-table operator: per grid row i the five coefficients {(i-1), W, C, E, (i+1)}, terms summed in that order"""
    n = u.shape[0]
    p = np.zeros((n + 2, n + 2))
    p[1:-1, 1:-1] = u
    t = ct[:, 0:1] * p[:-2, 1:-1]
    t = t + ct[:, 1:2] * p[1:-1, :-2]
    t = t + ct[:, 2:3] * p[1:-1, 1:-1]
    t = t + ct[:, 3:4] * p[1:-1, 2:]
    t = t + ct[:, 4:5] * p[2:, 1:-1]
    return t


def _rt_jacobi(ct, b, u, scale):
    res = b - _rt_apply(ct, u)
    return u + scale * (res * (1.0 / ct[:, 2:3]))


def _rt_tables(rng, n):
    q = float((n + 1) ** 2)
    ct = np.empty((n, 5))
    ct[:, 0] = q * rng.uniform(0.5, 1.5, n)
    ct[:, 1] = q * rng.uniform(0.5, 1.5, n)
    ct[:, 3] = ct[:, 1]
    ct[:, 4] = q * rng.uniform(0.5, 1.5, n)
    ct[:, 2] = -(ct[:, 0] + ct[:, 1] + ct[:, 3] + ct[:, 4])
    return ct, 1.0 / ct[:, 2]


@pytest.mark.parametrize("n", [3, 7, 63, 255, 509, 1023, 2047])
def test_row_table_forms_of_the_fused_2d_kernels(mgk, orc, n):
    """mgk_jacobi2_2d_rowcoef / mgk_jacobi_sumsq_rowcoef / mgk_residual_sumsq_rowcoef / mgk_prolong_jacobi_rowcoef /
    mgk_residual_restrict_2d_rowcoef: each equals the kernel-per-operation sequence on the same row tables, bit for bit"""
    rng = np.random.default_rng(8800 + n)
    nc = (n - 1) // 2
    ct, dt = _rt_tables(rng, n)
    has_c = nc >= 1 and nc % 2 == 1                 # 509: a sweep-only shape (no vertex-centred coarse grid below it)
    ctc, dtc = _rt_tables(rng, max(nc, 1))
    u, b, uc = _rand(rng, n * n).reshape(n, n), _rand(rng, n * n).reshape(n, n), _rand(rng, nc * nc)
    g, gc = mgk.geom(2, n), mgk.geom(2, nc if has_c else 1)
    du, db, dout = mgk.to_field(g, u.ravel()), mgk.to_field(g, b.ravel()), mgk.field(g)
    dct, ddt, ddtc = mgk.upload(ct.ravel()), mgk.upload(dt), mgk.upload(dtc)
    ss = C.c_double(0.0)
    for var, zc in ((-1, -1), (-1, 7), (38, 5)):             # 38: the prolongation sweep as independent waves on every size
        mgk.L.mgk_set_tuning(var, zc)
        # two sweeps in one pass
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_jacobi2_2d_rowcoef_f64(mgk.ctx, C.byref(g), dct, ddt, 0.8, db, du, dout, None))
        want = _rt_jacobi(ct, b, _rt_jacobi(ct, b, u, 0.8), 0.8)
        got = mgk.from_field(g, dout).reshape(n, n)
        assert np.array_equal(got, want), f"jacobi2 zc={zc}: {np.abs(got - want).max()}"
        raw = mgk.raw_field(g, dout)
        assert abs(np.abs(raw).sum() - np.abs(got).sum()) <= 1e-9 * np.abs(got).sum()
        # sweep + norm of the residual it forms, and the norm alone
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_jacobi_sumsq_rowcoef_f64(mgk.ctx, C.byref(g), dct, ddt, 0.8, db, du, dout, C.byref(ss), None))
        res = b - _rt_apply(ct, u)
        assert np.array_equal(mgk.from_field(g, dout).reshape(n, n), _rt_jacobi(ct, b, u, 0.8))
        assert abs(ss.value - float((res * res).sum())) <= 1e-12 * float((res * res).sum())
        mgk._chk(mgk.L.mgk_residual_sumsq_rowcoef_f64(mgk.ctx, C.byref(g), dct, db, du, C.byref(ss), None))
        assert abs(ss.value - float((res * res).sum())) <= 1e-12 * float((res * res).sum())
        if has_c:
            # prolongation fused into the sweep
            duc = mgk.to_field(gc, uc)
            corrected = orc.prolong_add(2, n, uc, u.ravel()).reshape(n, n)
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
            mgk._chk(mgk.L.mgk_prolong_jacobi_rowcoef_f64(mgk.ctx, C.byref(g), C.byref(gc), dct, ddt, 0.8, db, duc, du, dout, None))
            got = mgk.from_field(g, dout).reshape(n, n)
            want = _rt_jacobi(ct, b, corrected, 0.8)
            assert np.array_equal(got, want), f"prolong_jacobi zc={zc}: {np.abs(got - want).max()}"
            # residual + restriction (+ the coarse level's zero-guess sweep with ITS 1/diag table)
            dbc, duc0 = mgk.field(gc), mgk.field(gc)
            mgk._chk(mgk.L.mgk_residual_restrict_2d_rowcoef_f64(mgk.ctx, C.byref(g), C.byref(gc), dct, db, du, dbc, duc0, ddtc, 0.8, None))
            wantc = orc.restrict(2, n, res.ravel())
            assert np.array_equal(mgk.from_field(gc, dbc), wantc)
            assert np.array_equal(mgk.from_field(gc, duc0).reshape(nc, nc), 0.8 * (wantc.reshape(nc, nc) * dtc[:, None]))
            for p in (duc, dbc, duc0):
                mgk.free(p)
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, dout, dct, ddt, ddtc):
        mgk.free(p)


@pytest.mark.parametrize("n0,nlev", [(63, 5), (31, 3), (15, 4), (7, 2)])
def test_lds_tail_kernel_on_row_tables(mgk, orc, n0, nlev):
    """mgk_tail_cycle_rowcoef_f64 against the same levels stepped with numpy in the canonical order (zero-guess sweep, sweeps,
    residual + full weighting, prolongation + sweeps)"""
    rng = np.random.default_rng(8900 + n0)
    ns = [n0]
    for _ in range(nlev - 1):
        ns.append((ns[-1] - 1) // 2)
    tabs = [_rt_tables(rng, n) for n in ns]
    b0 = _rand(rng, n0 * n0).reshape(n0, n0)
    v0, v1, scale = 3, 3, 0.8

    def smooth(l, b, u, sweeps, zero):
        ct, dt = tabs[l]
        for it in range(sweeps):
            if it == 0 and zero:
                u = scale * (b * dt[:, None])
            else:
                u = _rt_jacobi(ct, b, u, scale)
        return u
    B, U = [b0], []
    for l in range(nlev):
        U.append(smooth(l, B[l], np.zeros_like(B[l]), v1 if l == nlev - 1 else v0, True))
        if l < nlev - 1:
            r = B[l] - _rt_apply(tabs[l][0], U[l])
            B.append(orc.restrict(2, ns[l], r.ravel()).reshape(ns[l + 1], ns[l + 1]))
    for l in range(nlev - 2, -1, -1):
        U[l] = orc.prolong_add(2, ns[l], U[l + 1].ravel(), U[l].ravel()).reshape(ns[l], ns[l])
        U[l] = smooth(l, B[l], U[l], v0, False)
    g = mgk.geom(2, n0)
    db, du = mgk.to_field(g, b0.ravel()), mgk.field(g)
    dts = [(mgk.upload(ct.ravel()), mgk.upload(dt)) for ct, dt in tabs]
    cta = (C.c_void_p * nlev)(*[C.cast(a, C.c_void_p).value for a, _ in dts])
    dta = (C.c_void_p * nlev)(*[C.cast(d, C.c_void_p).value for _, d in dts])
    nn = (C.c_int * nlev)(*ns)
    mgk._chk(mgk.L.mgk_tail_cycle_rowcoef_f64(mgk.ctx, C.byref(g), nlev, nn, cta, dta, scale, v0, v1, db, du, None))
    got = mgk.from_field(g, du).reshape(n0, n0)
    assert np.array_equal(got, U[0]), f"max diff {np.abs(got - U[0]).max()}"
    for p in [db, du] + [x for t in dts for x in t]:
        mgk.free(p)


@pytest.mark.parametrize("nf", [127, 255])
def test_sweep_fused_with_residual_restrict_bit_exact(mgk, orc, nf):
    """mgk_sweep_residual_restrict_f64 == mgk_jacobi_f64 followed by mgk_residual_restrict_jz_f64 (sweep, residual, full weighting,
    the coarse level's zero-guess sweep), bit for bit; tiles of 2 and of 4 rows, several z chunkings"""
    rng = np.random.default_rng(9900 + nf)
    nc = (nf - 1) // 2
    As, Asc = _stencil(orc, 3, nf), _stencil(orc, 3, nc)
    dinv, dinvc = 1.0 / As[3], 1.0 / Asc[3]
    u, b = _rand(rng, nf ** 3), _rand(rng, nf ** 3)
    gf, gc = mgk.geom(3, nf), mgk.geom(3, nc)
    assert mgk.L.mgk_sweep_residual_restrict_ok_f64(C.byref(gf), C.byref(gc)) == 1
    du, db, dw, dbc, duc = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.field(gf), mgk.field(gc), mgk.field(gc)
    w_ref, bc_ref, uc_ref = mgk.field(gf), mgk.field(gc), mgk.field(gc)
    mgk._chk(mgk.L.mgk_jacobi_f64(mgk.ctx, C.byref(gf), mgk.coef(As), dinv, 0.8, db, du, w_ref, None))
    mgk._chk(mgk.L.mgk_residual_restrict_jz_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), db, w_ref, bc_ref, uc_ref, dinvc, 0.8, None))
    want = [mgk.raw_field(gf, w_ref), mgk.raw_field(gc, bc_ref), mgk.raw_field(gc, uc_ref)]
    assert np.array_equal(mgk.from_field(gf, w_ref), orc.jacobi(3, nf, As, 0.8, b, u))
    for var, zc in [(-1, -1), (-1, 5), (-1, 16), (40, -1), (40, 7), (41, -1), (41, 9)]:
        mgk.L.mgk_set_tuning(var, zc)
        for f, g in ((dw, gf), (dbc, gc), (duc, gc)):
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, f, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_sweep_residual_restrict_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, du, dw, dbc, duc,
                                                       dinvc, 0.8, None))
        got = [mgk.raw_field(gf, dw), mgk.raw_field(gc, dbc), mgk.raw_field(gc, duc)]
        for name, x, y in zip(("swept field", "coarse rhs", "coarse first sweep"), got, want):
            assert np.array_equal(x, y), f"variant={var} zc={zc} {name}: max diff {np.abs(x - y).max()}"
    mgk.L.mgk_set_tuning(-1, -1)
    assert np.array_equal(mgk.from_field(gf, du), u)
    for p in (du, db, dw, dbc, duc, w_ref, bc_ref, uc_ref):
        mgk.free(p)


@pytest.mark.parametrize("n", [127, 255])
def test_two_sweeps_with_the_norm_of_the_input_residual(mgk, orc, n):
    """mgk_jacobi2_sumsq_f64: the field of mgk_jacobi2_f64 (bit for bit) and || b - A u ||^2 of the INPUT field (1e-13)"""
    rng = np.random.default_rng(9950 + n)
    As = _stencil(orc, 3, n)
    dinv = 1.0 / As[3]
    u, b = _rand(rng, n ** 3), _rand(rng, n ** 3)
    g = mgk.geom(3, n)
    assert mgk.L.mgk_jacobi2_sumsq_ok_f64(C.byref(g)) == 1
    du, db, dout, dref = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g), mgk.field(g)
    mgk._chk(mgk.L.mgk_jacobi2_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, du, dref, None))
    want = mgk.raw_field(g, dref)
    r = orc.residual(3, n, As, b, u)
    ss = C.c_double(0.0)
    for zc in (-1, 8, 29):
        mgk.L.mgk_set_tuning(-1, zc)
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_jacobi2_sumsq_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, du, dout, C.byref(ss), None))
        assert np.array_equal(mgk.raw_field(g, dout), want), f"zc={zc}"
        assert abs(ss.value - float(np.dot(r, r))) <= 1e-13 * float(np.dot(r, r)), f"zc={zc}"
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, dout, dref):
        mgk.free(p)


@pytest.mark.parametrize("nx,ny,nz", [(1023, 7, 9), (1023, 11, 5), (511, 11, 7), (255, 15, 3)])
def test_sweep_fused_with_residual_restrict_wide_rows(mgk, nx, ny, nz):
    """the 4- and 8-wave instances of mgk_sweep_residual_restrict_f64 (rows of 511 / 1023) on thin grids, against the two kernels it
    replaces (themselves pinned to the oracle): bit for bit, both tile heights"""
    rng = np.random.default_rng(9970 + nx + ny)
    nxc, nyc, nzc = (nx - 1) // 2, (ny - 1) // 2, (nz - 1) // 2
    q = float((nx + 1) ** 2)
    As = [q, q, q, -6.0 * q, q, q, q]
    dinv = 1.0 / As[3]
    gf, gc = mgk.geom(3, nx, ny, nz), mgk.geom(3, nxc, nyc, nzc)
    assert mgk.L.mgk_sweep_residual_restrict_ok_f64(C.byref(gf), C.byref(gc)) == 1
    u, b = _rand(rng, nx * ny * nz), _rand(rng, nx * ny * nz)
    du, db, dw, dbc, duc = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.field(gf), mgk.field(gc), mgk.field(gc)
    w_ref, bc_ref, uc_ref = mgk.field(gf), mgk.field(gc), mgk.field(gc)
    mgk._chk(mgk.L.mgk_jacobi_f64(mgk.ctx, C.byref(gf), mgk.coef(As), dinv, 0.8, db, du, w_ref, None))
    mgk._chk(mgk.L.mgk_residual_restrict_jz_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), db, w_ref, bc_ref, uc_ref, 0.25 * dinv, 0.8, None))
    want = [mgk.raw_field(gf, w_ref), mgk.raw_field(gc, bc_ref), mgk.raw_field(gc, uc_ref)]
    assert np.abs(want[1]).max() > 0
    for var, zc in [(-1, -1), (-1, 1), (40, -1), (40, 1), (41, -1), (41, 1)]:
        mgk.L.mgk_set_tuning(var, zc)
        for f, g in ((dw, gf), (dbc, gc), (duc, gc)):
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, f, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_sweep_residual_restrict_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, du, dw, dbc, duc,
                                                       0.25 * dinv, 0.8, None))
        got = [mgk.raw_field(gf, dw), mgk.raw_field(gc, dbc), mgk.raw_field(gc, duc)]
        for name, x, y in zip(("swept field", "coarse rhs", "coarse first sweep"), got, want):
            assert np.array_equal(x, y), f"variant={var} zc={zc} {name}: max diff {np.abs(x - y).max()}"
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, dw, dbc, duc, w_ref, bc_ref, uc_ref):
        mgk.free(p)


def _far_field(mgk, gfar, n, lo_plane=None, hi_plane=None):
    """far-plane field of geometry (n, n, 2): lo / hi ghost plane filled with the given n x n planes (None: zeros)"""
    f = np.zeros(gfar.total)
    for k, pl in ((-1, lo_plane), (2, hi_plane)):
        if pl is None:
            continue
        for i in range(n):
            o = gfar.org + k * gfar.plane + i * gfar.pitch
            f[o:o + n] = pl[i]
    return mgk.upload(f)


@pytest.mark.parametrize("n,cuts", [(127, (0, 20, 44, 63)), (127, (0, 2, 61, 63)), (255, (0, 64, 127))])
def test_four_pass_kernels_on_slabs_bit_exact(mgk, orc, n, cuts):
    """mgk_sweep_residual_restrict_slab_f64 and mgk_jacobi2_sumsq_slab_f64 on z-slabs (coarse plane ranges `cuts`): every slab, given
    its neighbours' planes the way the halo exchange delivers them (ghost planes of u and b, far / far2 / bfar), reproduces its part
    of the whole-grid results bit for bit -- swept field, coarse right-hand side, two-sweep field -- and the slabs' norm partials sum
    to the whole-grid norm; plane-range launches (interior first, boundaries after) included"""
    rng = np.random.default_rng(4100 + n + len(cuts))
    nc = (n - 1) // 2
    As = _stencil(orc, 3, n)
    dinv = 1.0 / As[3]
    u, b = _rand(rng, n ** 3), _rand(rng, n ** 3)
    L, coef = mgk.L, mgk.coef(As)
    # whole-grid reference from the kernels the slab forms replace (themselves pinned to the oracle)
    g, gc = mgk.geom(3, n), mgk.geom(3, nc)
    du, db, dw, dbc, d2 = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g), mgk.field(gc), mgk.field(g)
    mgk._chk(L.mgk_jacobi_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, db, du, dw, None))
    mgk._chk(L.mgk_residual_restrict_f64(mgk.ctx, C.byref(g), C.byref(gc), coef, db, dw, dbc, None))
    mgk._chk(L.mgk_jacobi2_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, db, du, d2, None))
    w_ref = mgk.from_field(g, dw).reshape(n, n, n)
    bc_ref = mgk.from_field(gc, dbc).reshape(nc, nc, nc)
    u2_ref = mgk.from_field(g, d2).reshape(n, n, n)
    r = orc.residual(3, n, As, b, u)
    norm_ref = float(np.dot(r, r))
    for p in (du, db, dw, dbc, d2):
        mgk.free(p)
    U, B = u.reshape(n, n, n), b.reshape(n, n, n)
    total = 0.0
    for s in range(len(cuts) - 1):
        kc0, kc1 = cuts[s], cuts[s + 1]
        last = (s == len(cuts) - 2)
        z0, z1 = 2 * kc0, (n if last else 2 * kc1)
        nz, nzc = z1 - z0, kc1 - kc0
        has_lo, has_hi = int(s > 0), int(not last)
        gs, gcs, gfar = mgk.geom(3, n, n, nz), mgk.geom(3, nc, nc, nzc), mgk.geom(3, n, n, 2)
        assert L.mgk_sweep_residual_restrict_slab_ok_f64(C.byref(gs), C.byref(gcs)) == 1
        us, bs = _slab_field(mgk, gs, u, n, z0), _slab_field(mgk, gs, b, n, z0)
        far = _far_field(mgk, gfar, n, U[z0 - 2] if has_lo else None, U[z1 + 1] if has_hi else None)
        far2 = _far_field(mgk, gfar, n, None, U[z1 + 2] if has_hi and z1 + 2 < n else None)
        bfar = _far_field(mgk, gfar, n, None, B[z1 + 1] if has_hi else None)
        out, bcs = mgk.field(gs), mgk.field(gcs)
        ranges = ((1, nzc - 2), (0, 1), (nzc - 2, nzc)) if nzc >= 5 else ((0, nzc),)
        for k0, k1 in ranges:
            mgk._chk(L.mgk_sweep_residual_restrict_slab_f64(mgk.ctx, C.byref(gs), C.byref(gcs), C.byref(gfar), coef, dinv, 0.8, bs, us, out,
                                                            far, far2, bfar, has_lo, has_hi, bcs, k0, k1, None))
        assert np.array_equal(mgk.from_field(gs, out).reshape(nz, n, n), w_ref[z0:z1]), f"slab {s}: swept field"
        assert np.array_equal(mgk.from_field(gcs, bcs).reshape(nzc, nc, nc), bc_ref[kc0:kc1]), f"slab {s}: coarse right-hand side"
        # two sweeps + norm, plane ranges; partial slots appended
        mgk._chk(L.mgk_memset0(mgk.ctx, out, 8 * gs.total, None))
        n1, off = C.c_int(0), 0
        zr = ((2, nz - 2), (0, 2), (nz - 2, nz)) if nz >= 6 else ((0, nz),)
        for a0, a1 in zr:
            mgk._chk(L.mgk_jacobi2_sumsq_slab_f64(mgk.ctx, C.byref(gs), C.byref(gfar), coef, dinv, 0.8, bs, us, out, far, has_lo, has_hi,
                                                  a0, a1, off, C.byref(n1), None))
            off += n1.value
        ss = C.c_double(0.0)
        mgk._chk(L.mgk_partials_finish(mgk.ctx, off, C.byref(ss), None))
        total += ss.value
        assert np.array_equal(mgk.from_field(gs, out).reshape(nz, n, n), u2_ref[z0:z1]), f"slab {s}: two-sweep field"
        for p in (us, bs, far, far2, bfar, out, bcs):
            mgk.free(p)
    assert abs(total - norm_ref) <= 1e-13 * norm_ref


@pytest.mark.parametrize("nf", [3, 7, 63, 121 * 2 + 1, 127, 255, 1023, 2047])
def test_2d_forms_of_the_four_pass_kernels_bit_exact(mgk, orc, nf):
    """mgk_sweep_residual_restrict_2d_f64 == sweep, residual, full weighting (+ the coarse level's zero-guess sweep);
    mgk_jacobi2_2d_sumsq_f64 == two sweeps + the norm of the input's residual; several y chunkings, tile edges (61 pairs per wave)"""
    rng = np.random.default_rng(7300 + nf)
    nc = (nf - 1) // 2
    As, Ac = _stencil(orc, 2, nf), _stencil(orc, 2, nc)
    dinv, dinv_c = 1.0 / As[2], 1.0 / Ac[2]
    u, b = _rand(rng, nf ** 2), _rand(rng, nf ** 2)
    gf, gc = mgk.geom(2, nf), mgk.geom(2, nc)
    du, db, dw, dbc, duc0 = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.field(gf), mgk.field(gc), mgk.field(gc)
    w1 = orc.jacobi(2, nf, As, 0.8, b, u)
    bc = orc.restrict(2, nf, orc.residual(2, nf, As, b, w1))
    w2 = orc.jacobi(2, nf, As, 0.8, b, w1)
    r = orc.residual(2, nf, As, b, u)
    ss = C.c_double(0.0)
    for zc in (-1, 1, 5, 64):
        mgk.L.mgk_set_tuning(-1, zc)
        for f, g in ((dw, gf), (dbc, gc), (duc0, gc)):
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, f, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_sweep_residual_restrict_2d_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, du, dw, dbc, duc0,
                                                          dinv_c, 0.8, None))
        assert np.array_equal(mgk.from_field(gf, dw), w1), f"zc={zc}: swept field"
        assert np.array_equal(mgk.from_field(gc, dbc), bc), f"zc={zc}: coarse right-hand side"
        assert np.array_equal(mgk.from_field(gc, duc0), 0.8 * (bc * dinv_c))
        for f, g, x in ((dw, gf, w1), (dbc, gc, bc)):
            raw = mgk.raw_field(g, f)
            assert abs(np.abs(raw).sum() - np.abs(x).sum()) <= 1e-9 * max(np.abs(x).sum(), 1e-300)      # ghosts / padding stay zero
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dw, 8 * gf.total, None))
        mgk._chk(mgk.L.mgk_jacobi2_2d_sumsq_f64(mgk.ctx, C.byref(gf), mgk.coef(As), dinv, 0.8, db, du, dw, C.byref(ss), None))
        assert np.array_equal(mgk.from_field(gf, dw), w2), f"zc={zc}: two sweeps"
        assert abs(ss.value - float(np.dot(r, r))) <= 1e-13 * float(np.dot(r, r))
    mgk.L.mgk_set_tuning(-1, -1)
    assert np.array_equal(mgk.from_field(gf, du), u)
    for p in (du, db, dw, dbc, duc0):
        mgk.free(p)


@pytest.mark.parametrize("n,prec", [(127, "f64"), (255, "f64"), (255, "f32")])
def test_three_sweeps_from_the_zero_guess_in_one_pass(mgk, orc, n, prec):
    """mgk_jacobi2_zero_*: J(J(J0(b))) reading b alone == mgk_jacobi_zero_* followed by mgk_jacobi2_* (themselves pinned to the oracle)"""
    rng = np.random.default_rng(5600 + n)
    As = _stencil(orc, 3, n)
    dinv = 1.0 / As[3]
    b = _rand(rng, n ** 3)
    if prec == "f64":
        g = mgk.geom(3, n)
        assert mgk.L.mgk_jacobi2_zero_ok_f64(C.byref(g)) == 1
        db, d1, d2, dout = mgk.to_field(g, b), mgk.field(g), mgk.field(g), mgk.field(g)
        mgk._chk(mgk.L.mgk_jacobi_zero_f64(mgk.ctx, C.byref(g), dinv, 0.8, db, d1, None))
        mgk._chk(mgk.L.mgk_jacobi2_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, d1, d2, None))
        want = mgk.raw_field(g, d2)
        assert np.array_equal(mgk.from_field(g, d2), orc.jacobi(3, n, As, 0.8, b, orc.jacobi(3, n, As, 0.8, b, orc.jacobi(3, n, As, 0.8, b, np.zeros(n ** 3), zero_guess=True))))
        for zc in (-1, 8, 21):
            mgk.L.mgk_set_tuning(-1, zc)
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
            mgk._chk(mgk.L.mgk_jacobi2_zero_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, dout, None))
            assert np.array_equal(mgk.raw_field(g, dout), want), f"zc={zc}"
        esz = 8
    else:
        g = mgk.geom32(n)
        assert mgk.L.mgk_jacobi2_zero_ok_f32(C.byref(g)) == 1
        db, d1, d2, dout = mgk.to_field32(g, b), mgk.alloc(4 * g.total), mgk.alloc(4 * g.total), mgk.alloc(4 * g.total)
        for f in (d1, d2, dout):
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, f, 4 * g.total, None))
        mgk._chk(mgk.L.mgk_jacobi_zero_f32(mgk.ctx, C.byref(g), dinv, 0.8, db, d1, None))
        mgk._chk(mgk.L.mgk_jacobi2_f32(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, d1, d2, None))
        want = mgk.from_field32(g, d2)
        for zc in (-1, 8, 21):
            mgk.L.mgk_set_tuning(-1, zc)
            mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 4 * g.total, None))
            mgk._chk(mgk.L.mgk_jacobi2_zero_f32(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, dout, None))
            assert np.array_equal(mgk.from_field32(g, dout), want), f"zc={zc}"
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (db, d1, d2, dout):
        mgk.free(p)


@pytest.mark.parametrize("nx,ny,nz", [(1023, 7, 9), (1023, 11, 5), (1023, 15, 17), (1023, 3, 33), (511, 7, 9), (511, 15, 33), (511, 511, 5)])
def test_prolongation_fused_into_a_two_sweep_pass(mgk, nx, ny, nz):
    """mgk_prolong_jacobi2_f64: J(J(u + P uc)) == mgk_prolong_jacobi_f64 followed by mgk_jacobi_f64, bit for bit, on thin
    grids with rows of 1023 / 511 (the 8- and 4-wave instances), several z chunkings"""
    rng = np.random.default_rng(6600 + nx + ny + nz)
    nxc, nyc, nzc = (nx - 1) // 2, (ny - 1) // 2, (nz - 1) // 2
    q = float((nx + 1) ** 2)
    As = [q, q, q, -6.0 * q, q, q, q]
    dinv = 1.0 / As[3]
    gf, gc = mgk.geom(3, nx, ny, nz), mgk.geom(3, nxc, nyc, nzc)
    assert mgk.L.mgk_prolong_jacobi2_ok_f64(C.byref(gf), C.byref(gc)) == 1
    u, b, uc = _rand(rng, nx * ny * nz), _rand(rng, nx * ny * nz), _rand(rng, nxc * nyc * nzc)
    du, db, duc, d1, d2, dout = mgk.to_field(gf, u), mgk.to_field(gf, b), mgk.to_field(gc, uc), mgk.field(gf), mgk.field(gf), mgk.field(gf)
    mgk._chk(mgk.L.mgk_prolong_jacobi_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, duc, du, d1, None))
    mgk._chk(mgk.L.mgk_jacobi_f64(mgk.ctx, C.byref(gf), mgk.coef(As), dinv, 0.8, db, d1, d2, None))
    want = mgk.raw_field(gf, d2)
    assert np.abs(want).max() > 0
    for var, zc in ((-1, -1), (-1, 2), (-1, 4), (46, -1), (46, 4)):
        mgk.L.mgk_set_tuning(var, zc)
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * gf.total, None))
        mgk._chk(mgk.L.mgk_prolong_jacobi2_f64(mgk.ctx, C.byref(gf), C.byref(gc), mgk.coef(As), dinv, 0.8, db, duc, du, dout, None))
        got = mgk.raw_field(gf, dout)
        assert np.array_equal(got, want), f"variant={var} zc={zc}: max diff {np.abs(got - want).max()}"
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, duc, d1, d2, dout):
        mgk.free(p)


@pytest.mark.parametrize("n", [127, 255])
def test_two_sweeps_with_the_norm_of_the_mid_iterate(mgk, orc, n):
    """mgk_jacobi2_sumsq_mid_f64: the field of mgk_jacobi2_f64 (bit for bit) and || b - A J(u) ||^2, the residual of the first sweep's
    output (1e-13)"""
    rng = np.random.default_rng(9960 + n)
    As = _stencil(orc, 3, n)
    dinv = 1.0 / As[3]
    u, b = _rand(rng, n ** 3), _rand(rng, n ** 3)
    g = mgk.geom(3, n)
    du, db, dout, dref = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g), mgk.field(g)
    mgk._chk(mgk.L.mgk_jacobi2_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, du, dref, None))
    want = mgk.raw_field(g, dref)
    r = orc.residual(3, n, As, b, orc.jacobi(3, n, As, 0.8, b, u))
    ss = C.c_double(0.0)
    for zc in (-1, 8, 29):
        mgk.L.mgk_set_tuning(-1, zc)
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
        mgk._chk(mgk.L.mgk_jacobi2_sumsq_mid_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, db, du, dout, C.byref(ss), None))
        assert np.array_equal(mgk.raw_field(g, dout), want), f"zc={zc}"
        assert abs(ss.value - float(np.dot(r, r))) <= 1e-13 * float(np.dot(r, r)), f"zc={zc}"
    mgk.L.mgk_set_tuning(-1, -1)
    for p in (du, db, dout, dref):
        mgk.free(p)


@pytest.mark.parametrize("n", [3, 7, 63, 255, 509, 1023, 2047])
def test_norm_pass_that_stores_the_residual_and_makes_the_next_sweep(mgk, orc, n):
    """mgk_jacobi_sumsq_store_f64 (the drop-in's KSPBuildResidual + VecNorm + first sweep of the next KSPSolve, 2-D): r is the residual
    kernel's, unew the sweep kernel's, bit for bit; the sum is ||r||^2; ghosts and padding of both outputs stay zero; u and b untouched"""
    rng = np.random.default_rng(9300 + n)
    q = float((n + 1) ** 2)
    As = [q, q, -4.0 * q, q, q]
    dinv = 1.0 / As[2]
    u, b = _rand(rng, n ** 2), _rand(rng, n ** 2)
    g = mgk.geom(2, n)
    du, db, dout, dr = mgk.to_field(g, u), mgk.to_field(g, b), mgk.field(g), mgk.field(g)
    ss = C.c_double(0.0)
    mgk._chk(mgk.L.mgk_jacobi_sumsq_store_f64(mgk.ctx, C.byref(g), mgk.coef(As), dinv, 0.8, None, None, db, du, dout, dr, C.byref(ss), None))
    res = orc.residual(2, n, As, b, u)
    assert np.array_equal(mgk.from_field(g, dr), res)
    assert np.array_equal(mgk.from_field(g, dout), orc.jacobi(2, n, As, 0.8, b, u))
    assert abs(ss.value - float(res @ res)) <= 1e-12 * float(res @ res)
    for f, v in ((dr, mgk.from_field(g, dr)), (dout, mgk.from_field(g, dout))):
        raw = mgk.raw_field(g, f)
        assert abs(np.abs(raw).sum() - np.abs(v).sum()) <= 1e-9 * np.abs(v).sum()
    assert np.array_equal(mgk.from_field(g, du), u) and np.array_equal(mgk.from_field(g, db), b)
    # row tables (stretched meshes)
    ct, dt = _rt_tables(rng, n)
    dct, ddt = mgk.upload(ct.ravel()), mgk.upload(dt)
    mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dout, 8 * g.total, None))
    mgk._chk(mgk.L.mgk_memset0(mgk.ctx, dr, 8 * g.total, None))
    mgk._chk(mgk.L.mgk_jacobi_sumsq_store_f64(mgk.ctx, C.byref(g), None, 1.0, 0.8, dct, ddt, db, du, dout, dr, C.byref(ss), None))
    U, B = u.reshape(n, n), b.reshape(n, n)
    rr = B - _rt_apply(ct, U)
    assert np.array_equal(mgk.from_field(g, dr).reshape(n, n), rr)
    assert np.array_equal(mgk.from_field(g, dout).reshape(n, n), _rt_jacobi(ct, B, U, 0.8))
    assert abs(ss.value - float((rr * rr).sum())) <= 1e-12 * float((rr * rr).sum())
    for p in (du, db, dout, dr, dct, ddt):
        mgk.free(p)


def _thin_slab_field(mgk, g, W, z0):
    """padded fp64 field of the z-slab [z0, z0 + g.nz) of a whole (nzw, ny, nx) array, ghost planes taken from the neighbours"""
    nzw, ny, nx = W.shape
    pad = np.zeros(g.total)
    for k in range(-1, g.nz + 1):
        kz = z0 + k
        if 0 <= kz < nzw:
            for i in range(ny):
                o = g.org + k * g.plane + i * g.pitch
                pad[o:o + nx] = W[kz, i]
    return mgk.upload(pad)


@pytest.mark.parametrize("n,nzcw,cuts", [(511, 15, (0, 4, 9, 15)), (511, 11, (0, 2, 11)), (1023, 13, (0, 5, 7, 13)), (511, 9, (0, 9))])
def test_ninety_one_byte_passes_on_slabs_bit_exact(mgk, orc, n, nzcw, cuts):
    """mgk_prolong_jacobi2_slab_f64 and mgk_jacobi2_sumsq_mid_slab_f64 (round 3: the 91-byte fine level on z-slabs) on thin grids n x n x (2 nzcw + 1)
    cut at the coarse planes `cuts`: every slab, given its neighbours' planes the way the grouped exchanges deliver them (ghost planes of u, b and of the
    coarse u; far: the neighbours' second planes of u; cfar: the lower neighbour's second-last coarse plane), reproduces its part of the whole-grid
    results of mgk_prolong_jacobi2_f64 / mgk_jacobi2_sumsq_mid_f64 (themselves pinned to the oracle: tests/test_headline_width_gpu.py) bit for bit, in
    the plane ranges the solver launches (interior first, boundaries after); the slabs' norm partials sum to the whole-grid norm"""
    rng = np.random.default_rng(9100 + n + nzcw)
    nc, nzw = (n - 1) // 2, 2 * nzcw + 1
    As = _stencil(orc, 3, n)
    dinv = 1.0 / As[3]
    L, coef = mgk.L, mgk.coef(As)
    g, gc = mgk.geom(3, n, n, nzw), mgk.geom(3, nc, nc, nzcw)
    U, B, UC = rng.uniform(-1, 1, (nzw, n, n)), rng.uniform(-1, 1, (nzw, n, n)), rng.uniform(-1, 1, (nzcw, nc, nc))
    du, db, duc = _thin_slab_field(mgk, g, U, 0), _thin_slab_field(mgk, g, B, 0), _thin_slab_field(mgk, gc, UC, 0)
    dpj, dj2 = mgk.field(g), mgk.field(g)
    assert L.mgk_prolong_jacobi2_ok_f64(C.byref(g), C.byref(gc)) == 1
    mgk._chk(L.mgk_prolong_jacobi2_f64(mgk.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.8, db, duc, du, dpj, None))
    ss = C.c_double(0.0)
    mgk._chk(L.mgk_jacobi2_sumsq_mid_f64(mgk.ctx, C.byref(g), coef, dinv, 0.8, db, du, dj2, C.byref(ss), None))
    pj_ref = mgk.from_field(g, dpj).reshape(nzw, n, n)
    j2_ref = mgk.from_field(g, dj2).reshape(nzw, n, n)
    norm_ref = ss.value
    for p in (du, db, duc, dpj, dj2):
        mgk.free(p)
    total = 0.0
    for s in range(len(cuts) - 1):
        kc0, kc1 = cuts[s], cuts[s + 1]
        last = (s == len(cuts) - 2)
        z0, z1 = 2 * kc0, (nzw if last else 2 * kc1)
        nz, nzc = z1 - z0, kc1 - kc0
        has_lo, has_hi = int(s > 0), int(not last)
        gs, gcs, gfar, gcfar = mgk.geom(3, n, n, nz), mgk.geom(3, nc, nc, nzc), mgk.geom(3, n, n, 2), mgk.geom(3, nc, nc, 2)
        assert L.mgk_prolong_jacobi2_slab_ok_f64(C.byref(gs), C.byref(gcs), has_hi) == 1
        us, bs, ucs = _thin_slab_field(mgk, gs, U, z0), _thin_slab_field(mgk, gs, B, z0), _thin_slab_field(mgk, gcs, UC, kc0)
        far = _far_field(mgk, gfar, n, U[z0 - 2] if has_lo else None, U[z1 + 1] if has_hi else None)
        cfar = _far_field(mgk, gcfar, nc, UC[kc0 - 2] if has_lo and kc0 >= 2 else None, None)
        out = mgk.field(gs)
        zi0, zi1 = (4 if has_lo else 0), (nz - 2 if has_hi else nz)
        ranges = [(zi0, zi1)] + ([(0, 4)] if has_lo else []) + ([(nz - 2, nz)] if has_hi else []) if zi1 - zi0 >= 2 else [(0, nz)]
        for zc in (-1, 8):
            L.mgk_set_tuning(-1, zc)
            mgk._chk(L.mgk_memset0(mgk.ctx, out, 8 * gs.total, None))
            for a0, a1 in ranges:
                mgk._chk(L.mgk_prolong_jacobi2_slab_f64(mgk.ctx, C.byref(gs), C.byref(gcs), C.byref(gfar), C.byref(gcfar), coef, dinv, 0.8, bs, ucs, us, out,
                                                        far, cfar, has_lo, has_hi, a0, a1, None))
            got = mgk.from_field(gs, out).reshape(nz, n, n)
            assert np.array_equal(got, pj_ref[z0:z1]), f"slab {s} zc={zc}: prolongation + two sweeps, planes {np.unique(np.nonzero(got != pj_ref[z0:z1])[0])}"
        L.mgk_set_tuning(-1, -1)
        mgk._chk(L.mgk_memset0(mgk.ctx, out, 8 * gs.total, None))
        n1, off = C.c_int(0), 0
        zr = ((2, nz - 2), (0, 2), (nz - 2, nz)) if nz >= 6 else ((0, nz),)
        for a0, a1 in zr:
            mgk._chk(L.mgk_jacobi2_sumsq_mid_slab_f64(mgk.ctx, C.byref(gs), C.byref(gfar), coef, dinv, 0.8, bs, us, out, far, has_lo, has_hi,
                                                      a0, a1, off, C.byref(n1), None))
            off += n1.value
        sp = C.c_double(0.0)
        mgk._chk(L.mgk_partials_finish(mgk.ctx, off, C.byref(sp), None))
        total += sp.value
        assert np.array_equal(mgk.from_field(gs, out).reshape(nz, n, n), j2_ref[z0:z1]), f"slab {s}: two sweeps of the mid-norm pass"
        for p in (us, bs, ucs, far, cfar, out):
            mgk.free(p)
    assert abs(total - norm_ref) <= 1e-13 * norm_ref
