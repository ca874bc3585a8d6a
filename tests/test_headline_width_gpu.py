"""The kernel INSTANCES the headline runs, against the CPU oracle directly.

test_kernels_gpu.py pins every kernel to the oracle on cubes up to 255^3 -- rows of at most 256, the 1- and 2-wave instances.
The default cycle at 1023^3 / 511^3 (BASELINE configs 3-5) runs the 8- and 4-wave instances of the same templates (rows of 1024 /
512: k_jacobi2r<double,8,3,*>, k_pj2r3<8>, k_rrrow<double,8,..>, k_jrow<double,8,..>, k_pjrow<double,8,..>, k_srr4b<8>, and the
<..,4,..> forms on level 1).  Here every fine-level entry point of that cycle is called on thin grids n x n x nz with n = 1023 and
511 -- the oracle evaluates such a slab in milliseconds (mgo_st_* with nz) -- and compared with compositions of the oracle's
operators: fields np.array_equal (bit for bit), sums of squares to 1e-13.  No GPU result is compared with another GPU result.
Then whole cycles: a 511^3 solve against mgo_vcycle (bit-identical u, history <= 1e-12), and fixed cycles at 1023^3 against the
oracle's matrix-free leg where the host has the memory for it.

Which instance each case covers (DESIGN.md section 5 carries the same table):
  mgk_jacobi_f64              k_jrow<double,8|4,2,false>        plain sweep (finalize_iterate, bench probe)           :1531
  mgk_jacobi_sumsq_f64        k_jrow<double,8|4,2,true>         sweep + norm of the input's residual                  :1545-1546 + :1531
  mgk_residual_sumsq_f64      k_stencil<..,MODE_RESNORM>        norm of the first / last cycle                        :1518, :1546
  mgk_jacobi2_f64             k_jacobi2r<double,8,3,0> / k_jacobi2b<double,4>   two sweeps                            :1531, :1542
  mgk_jacobi2_sumsq_f64       k_jacobi2r<double,8|4,3,1>        two sweeps + norm of the input's residual             :1546 + :1531
  mgk_jacobi2_sumsq_mid_f64   k_jacobi2r<double,8|4,3,2>        two sweeps + norm of the mid iterate's residual       :1542 + :1546 + :1531
  mgk_jacobi2_zero_f64        k_jacobi2<double,8|4,3,true>      three sweeps from the zero guess (level 1)            :1536
  mgk_prolong_jacobi_f64      k_pjrow<double,8|4,..>            prolongation + correction + sweep                     :1540-1542
  mgk_prolong_jacobi2_f64     k_pj2r3<8|4> (and k_pj2r<8|4>)    prolongation + correction + two sweeps                :1540-1542
  mgk_residual_restrict_f64 / _jz_f64   k_rrrow<double,8|4,1,*>  residual + full weighting (+ coarse zero-guess sweep) :1534-1536
  mgk_sweep_residual_restrict_f64       k_srr4b<8|4>            sweep + residual + full weighting                     :1531 + :1534-1535
(src/solver.c lines of /root/reference.)"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import Oracle

pytestmark = pytest.mark.gpu
RED_RTOL = 1e-13
SCALE = 6.0 / 7.0

SHAPES = [(1023, 3), (1023, 5), (1023, 9), (511, 3), (511, 5), (511, 9)]


@pytest.fixture(scope="module")
def orc():
    return Oracle()


class Thin:
    """seeded fields on an n x n x nz grid and its coarse grid, on the device and on the host"""

    def __init__(self, mgk, orc, n, nz, seed):
        self.mgk, self.orc, self.n, self.nz = mgk, orc, n, nz
        self.nc, self.nzc = (n - 1) // 2, (nz - 1) // 2
        rng = np.random.default_rng(seed)
        self.As = orc.level_stencil(3, n + 2, 0)[0]                 # the level whose grid has n unknowns per side
        self.Asc = orc.level_stencil(3, self.nc + 2, 0)[0]
        self.dinv, self.dinvc = 1.0 / self.As[3], 1.0 / self.Asc[3]
        N, Nc = n * n * nz, self.nc * self.nc * self.nzc
        self.u, self.b, self.uc = rng.uniform(-1, 1, N), rng.uniform(-1, 1, N), rng.uniform(-1, 1, Nc)
        self.g, self.gc = mgk.geom(3, n, n, nz), mgk.geom(3, self.nc, self.nc, self.nzc)
        self.du, self.db, self.duc = mgk.to_field(self.g, self.u), mgk.to_field(self.g, self.b), mgk.to_field(self.gc, self.uc)
        self.coef = mgk.coef(self.As)
        self._own = [self.du, self.db, self.duc]

    # oracle operators on the thin grid
    def J(self, u, zero_guess=False):
        return self.orc.jacobi(3, self.n, self.As, SCALE, self.b, u, zero_guess=zero_guess, nz=self.nz)

    def res(self, u):
        return self.orc.residual(3, self.n, self.As, self.b, u, nz=self.nz)

    def R(self, r):
        return self.orc.restrict(3, self.n, r, nzf=self.nz, nzc=self.nzc)

    def P(self, uc, u):
        return self.orc.prolong_add(3, self.n, uc, u, nzf=self.nz, nzc=self.nzc)

    def out(self, coarse=False):
        g = self.gc if coarse else self.g
        f = self.mgk.field(g)
        self.mgk._chk(self.mgk.L.mgk_memset0(self.mgk.ctx, f, 8 * g.total, None))
        self._own.append(f)
        return f

    def get(self, f, coarse=False):
        return self.mgk.from_field(self.gc if coarse else self.g, f)

    def ghosts_clean(self, f, coarse=False):
        g = self.gc if coarse else self.g
        raw, inner = self.mgk.raw_field(g, f), self.mgk.from_field(g, f)
        return abs(np.abs(raw).sum() - np.abs(inner).sum()) <= 1e-9 * max(np.abs(inner).sum(), 1e-300)

    def close(self):
        for p in self._own:
            self.mgk.free(p)


def _close(a, b):
    return abs(a - b) <= RED_RTOL * abs(b)


@pytest.mark.parametrize("n,nz", SHAPES)
def test_sweeps_and_norms_against_the_oracle(mgk, orc, n, nz):
    t = Thin(mgk, orc, n, nz, 31000 + n + nz)
    L, g, ss = mgk.L, C.byref(t.g), C.c_double()
    j1 = t.J(t.u)
    j2 = t.J(j1)
    r0, r1 = t.res(t.u), t.res(j1)
    for zc in (-1, 4):                                              # default marching chunks, and short ones (chunk seams inside nz = 9)
        L.mgk_set_tuning(-1, zc)
        o = t.out()
        mgk._chk(L.mgk_jacobi_f64(mgk.ctx, g, t.coef, t.dinv, SCALE, t.db, t.du, o, None))
        assert np.array_equal(t.get(o), j1), f"mgk_jacobi_f64 zc={zc}"
        assert t.ghosts_clean(o)
        o = t.out()
        mgk._chk(L.mgk_jacobi_sumsq_f64(mgk.ctx, g, t.coef, t.dinv, SCALE, t.db, t.du, o, C.byref(ss), None))
        assert np.array_equal(t.get(o), j1), f"mgk_jacobi_sumsq_f64 zc={zc}"
        assert _close(ss.value, orc.sumsq(r0))
        mgk._chk(L.mgk_residual_sumsq_f64(mgk.ctx, g, t.coef, t.db, t.du, C.byref(ss), None))
        assert _close(ss.value, orc.sumsq(r0))
        o = t.out()
        mgk._chk(L.mgk_residual_f64(mgk.ctx, g, t.coef, t.db, t.du, o, None))
        assert np.array_equal(t.get(o), r0), f"mgk_residual_f64 zc={zc}"
        mgk._chk(L.mgk_sumsq_f64(mgk.ctx, g, t.db, C.byref(ss), None))
        assert _close(ss.value, orc.sumsq(t.b))
        o = t.out()
        mgk._chk(L.mgk_jacobi2_f64(mgk.ctx, g, t.coef, t.dinv, SCALE, t.db, t.du, o, None))
        assert np.array_equal(t.get(o), j2), f"mgk_jacobi2_f64 zc={zc}"
        assert t.ghosts_clean(o)
        assert L.mgk_jacobi2_sumsq_ok_f64(g) == 1
        o = t.out()
        mgk._chk(L.mgk_jacobi2_sumsq_f64(mgk.ctx, g, t.coef, t.dinv, SCALE, t.db, t.du, o, C.byref(ss), None))
        assert np.array_equal(t.get(o), j2), f"mgk_jacobi2_sumsq_f64 zc={zc}"
        assert _close(ss.value, orc.sumsq(r0))
        o = t.out()
        mgk._chk(L.mgk_jacobi2_sumsq_mid_f64(mgk.ctx, g, t.coef, t.dinv, SCALE, t.db, t.du, o, C.byref(ss), None))
        assert np.array_equal(t.get(o), j2), f"mgk_jacobi2_sumsq_mid_f64 zc={zc}"
        assert _close(ss.value, orc.sumsq(r1))
        assert t.ghosts_clean(o)
        if L.mgk_jacobi2_zero_ok_f64(g) == 1:
            o = t.out()
            mgk._chk(L.mgk_jacobi2_zero_f64(mgk.ctx, g, t.coef, t.dinv, SCALE, t.db, o, None))
            z3 = t.J(t.J(t.J(np.zeros_like(t.u), zero_guess=True)))
            assert np.array_equal(t.get(o), z3), f"mgk_jacobi2_zero_f64 zc={zc}"
        o = t.out()
        mgk._chk(L.mgk_jacobi_zero_f64(mgk.ctx, g, t.dinv, SCALE, t.db, o, None))
        assert np.array_equal(t.get(o), t.J(np.zeros_like(t.u), zero_guess=True))
    L.mgk_set_tuning(-1, -1)
    assert np.array_equal(t.get(t.du), t.u) and np.array_equal(t.get(t.db), t.b)      # inputs untouched
    t.close()


@pytest.mark.parametrize("n,nz", SHAPES)
def test_transfer_passes_against_the_oracle(mgk, orc, n, nz):
    t = Thin(mgk, orc, n, nz, 32000 + n + nz)
    L, g, gc = mgk.L, C.byref(t.g), C.byref(t.gc)
    pu = t.P(t.uc, t.u)                                             # u + P uc
    pj1 = t.J(pu)
    pj2 = t.J(pj1)
    bc = t.R(t.res(t.u))                                            # R (b - A u)
    jz = orc.jacobi(3, t.nc, t.Asc, SCALE, bc, np.zeros_like(bc), zero_guess=True, nz=t.nzc)
    j1 = t.J(t.u)
    bc1 = t.R(t.res(j1))                                            # R (b - A J(u))
    jz1 = orc.jacobi(3, t.nc, t.Asc, SCALE, bc1, np.zeros_like(bc1), zero_guess=True, nz=t.nzc)
    for zc in (-1, 4):
        L.mgk_set_tuning(-1, zc)
        o = t.out()
        mgk._chk(L.mgk_prolong_jacobi_f64(mgk.ctx, g, gc, t.coef, t.dinv, SCALE, t.db, t.duc, t.du, o, None))
        assert np.array_equal(t.get(o), pj1), f"mgk_prolong_jacobi_f64 zc={zc}"
        assert t.ghosts_clean(o)
        assert L.mgk_prolong_jacobi2_ok_f64(g, gc) == 1
        for var in (-1, 46):                                        # the unrolled form k_pj2r3<8|4> (default) and the copying form k_pj2r<8|4>
            L.mgk_set_tuning(var, zc)
            o = t.out()
            mgk._chk(L.mgk_prolong_jacobi2_f64(mgk.ctx, g, gc, t.coef, t.dinv, SCALE, t.db, t.duc, t.du, o, None))
            assert np.array_equal(t.get(o), pj2), f"mgk_prolong_jacobi2_f64 variant={var} zc={zc}"
            assert t.ghosts_clean(o)
        L.mgk_set_tuning(-1, zc)
        o = t.out()
        mgk._chk(L.mgk_prolong_add_f64(mgk.ctx, g, gc, t.duc, t.du, None))          # in place on u: restore afterwards
        assert np.array_equal(t.get(t.du), pu), "mgk_prolong_add_f64"
        mgk.free(t.du)
        t._own.remove(t.du)
        t.du = mgk.to_field(t.g, t.u)
        t._own.append(t.du)
        oc = t.out(coarse=True)
        mgk._chk(L.mgk_residual_restrict_f64(mgk.ctx, g, gc, t.coef, t.db, t.du, oc, None))
        assert np.array_equal(t.get(oc, True), bc), f"mgk_residual_restrict_f64 zc={zc}"
        assert t.ghosts_clean(oc, True)
        oc, ou = t.out(coarse=True), t.out(coarse=True)
        mgk._chk(L.mgk_residual_restrict_jz_f64(mgk.ctx, g, gc, t.coef, t.db, t.du, oc, ou, t.dinvc, SCALE, None))
        assert np.array_equal(t.get(oc, True), bc), f"mgk_residual_restrict_jz_f64 zc={zc}: coarse right-hand side"
        assert np.array_equal(t.get(ou, True), jz), f"mgk_residual_restrict_jz_f64 zc={zc}: coarse zero-guess sweep"
        if L.mgk_sweep_residual_restrict_ok_f64(g, gc) == 1:
            for var in (-1, 40, 41):                                # default, and both tile heights of the LDS form
                L.mgk_set_tuning(var, zc)
                o, oc, ou = t.out(), t.out(coarse=True), t.out(coarse=True)
                mgk._chk(L.mgk_sweep_residual_restrict_f64(mgk.ctx, g, gc, t.coef, t.dinv, SCALE, t.db, t.du, o, oc, ou, t.dinvc, SCALE, None))
                assert np.array_equal(t.get(o), j1), f"mgk_sweep_residual_restrict_f64 var={var} zc={zc}: swept field"
                assert np.array_equal(t.get(oc, True), bc1), f"mgk_sweep_residual_restrict_f64 var={var} zc={zc}: coarse right-hand side"
                assert np.array_equal(t.get(ou, True), jz1), f"mgk_sweep_residual_restrict_f64 var={var} zc={zc}: coarse zero-guess sweep"
                assert t.ghosts_clean(o) and t.ghosts_clean(oc, True)
            L.mgk_set_tuning(-1, zc)
        # the two transfer kernels alone (levels that do not fuse)
        rr = t.out()
        mgk._chk(L.mgk_residual_f64(mgk.ctx, g, t.coef, t.db, t.du, rr, None))
        oc = t.out(coarse=True)
        mgk._chk(L.mgk_restrict_fw_f64(mgk.ctx, g, gc, rr, oc, None))
        assert np.array_equal(t.get(oc, True), bc), "mgk_restrict_fw_f64"
    L.mgk_set_tuning(-1, -1)
    assert np.array_equal(t.get(t.du), t.u) and np.array_equal(t.get(t.db), t.b) and np.array_equal(t.get(t.duc, True), t.uc)
    t.close()


def test_config3_solve_equals_the_oracle_cycle(orc):
    """BASELINE config 3 end to end: the default (fully fused) 511^3 solve against mgo_vcycle's matrix-free leg -- same iteration count,
    bit-identical solution, residual history to 1e-12 (north_star's tolerance; the norms differ in summation order only)."""
    from multigrid_petsc_amd.solver import Solver
    s = Solver(3, 513, 9, v=(3, 3), maxiter=40, scale=SCALE)
    s.set_rhs_problem()
    it = s.solve()
    rn, u = s.rnorm, s.solution()
    s.close()
    ref = orc.vcycle(3, 513, 9, 3, 3, maxiter=40, scale=SCALE, use_csr=0)
    assert it == ref["iters"], (it, ref["iters"])
    assert np.abs(rn / ref["rnorm"] - 1.0).max() <= 1e-12
    assert np.array_equal(u, ref["u"])


def _host_mem_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


@pytest.mark.timeout(1500)
def test_headline_cycles_equal_the_oracle_cycles(orc):
    """The headline configuration itself (1023^3, 10 levels): two default cycles on the GPU against two cycles of the oracle's
    matrix-free leg.  Compared: the residual history (1e-12) and the solution field, bit for bit.  The oracle needs about 60 GiB of
    host memory at this size (7 vectors of 8.6 GB on the fine level); skipped, with that reason, where the host has less."""
    need = 100.0
    have = _host_mem_gib()
    if have < need:
        pytest.skip(f"the oracle's 1023^3 cycle needs ~{need:.0f} GiB of host memory, {have:.0f} GiB available")
    from multigrid_petsc_amd.solver import Solver
    cycles = 2
    s = Solver(3, 1025, 10, v=(3, 3), maxiter=cycles + 1, scale=SCALE)
    s.set_rhs_problem()
    s.cycles(cycles)
    s.sync()
    rn, e, u = s.rnorm, s.error_norms(), s.solution()
    s.close()
    ref = orc.vcycle(3, 1025, 10, 3, 3, maxiter=cycles, scale=SCALE, use_csr=0, fixed_cycles=cycles)
    assert ref["iters"] == cycles and len(rn) == cycles + 1
    assert np.abs(rn / ref["rnorm"] - 1.0).max() <= 1e-12
    same = np.array_equal(u, ref["u"])
    del u
    eo = orc.error_norms(3, 1025, ref["u"])
    assert same, "solution after two cycles differs from the oracle's"
    # (the three GetError sums over 1.07e9 points: the max is exact, the sums differ in summation order only -- 9e-12 observed)
    assert e[0] == eo[0] and abs(e[1] - eo[1]) <= 1e-10 * eo[1] and abs(e[2] - eo[2]) <= 1e-10 * eo[2]
