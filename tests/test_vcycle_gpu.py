"""GPU parity of the whole V-cycle driver (C host + HIP kernels) against the CPU oracle.

Bar (BASELINE.json north_star): iteration count identical, fp64 residual norms within 1e-12 relative
per entry, solution field identical (the kernels are bit-faithful; only the norm's summation order
differs), plus the closed-form known answer  err_max = d*pi^2 / ((4d/h^2) sin^2(pi h/2)) - 1."""
import math

import numpy as np
import pytest

from oracle import Oracle

pytestmark = pytest.mark.gpu
RTOL = 1e-12


@pytest.fixture(scope="module")
def orc():
    return Oracle()


CASES = [
    # dim npts levels scale       (poisson.in default: npts 17, 2 levels, v 3,3)
    (2, 17, 2, 1.0), (2, 17, 2, 0.8), (2, 17, 4, 0.8), (2, 33, 5, 0.8), (2, 129, 7, 0.8), (2, 129, 6, 0.8),
    (2, 129, 2, 0.8), (2, 513, 9, 0.8), (2, 9, 1, 0.8),
    (3, 9, 3, 0.8), (3, 17, 4, 6.0 / 7.0), (3, 17, 2, 1.0), (3, 33, 5, 6.0 / 7.0), (3, 65, 6, 6.0 / 7.0),
    (3, 129, 7, 6.0 / 7.0),
]


@pytest.mark.parametrize("dim,npts,levels,scale", CASES)
def test_solve_matches_oracle(orc, dim, npts, levels, scale):
    from multigrid_petsc_amd.solver import Solver
    maxiter = 200
    s = Solver(dim, npts, levels, v=(3, 3), maxiter=maxiter, scale=scale)
    s.set_rhs_problem()
    it = s.solve()
    ref = orc.vcycle(dim, npts, levels, 3, 3, maxiter=maxiter, scale=scale, use_csr=0)
    assert it == ref["iters"]
    assert abs(s.bnorm - ref["bnorm"]) <= RTOL * ref["bnorm"]
    rn = s.rnorm
    assert rn.shape == ref["rnorm"].shape
    rel = np.abs(rn - ref["rnorm"]) / ref["rnorm"]
    assert rel.max() <= RTOL, f"residual history differs: max rel {rel.max()}"
    u = s.solution()
    assert np.array_equal(u, ref["u"]), f"solution not bit-identical, max diff {np.abs(u - ref['u']).max()}"
    e = s.error_norms()
    eref = orc.error_norms(dim, npts, ref["u"])
    assert e[0] == eref[0]
    assert abs(e[1] - eref[1]) <= RTOL * eref[1] and abs(e[2] - eref[2]) <= RTOL * eref[2]
    if it < maxiter:
        h = 1.0 / (npts - 1)
        kat = dim * math.pi ** 2 / ((4 * dim / h ** 2) * math.sin(math.pi * h / 2) ** 2) - 1.0
        assert abs(e[0] - kat) <= 5e-7     # iteration stops at ||r|| <= 1e-7 ||b||, |u| = O(1)
    s.close()


@pytest.mark.parametrize("dim,npts,levels", [(2, 65, 5), (3, 33, 4)])
def test_chebyshev_smoother_matches_oracle(orc, dim, npts, levels):
    from multigrid_petsc_amd.solver import Solver
    # Jacobi-preconditioned operator has spectrum in (0, 2); smooth the upper part
    emin, emax = 0.2, 2.0
    s = Solver(dim, npts, levels, v=(3, 3), maxiter=60, ksp_type="chebyshev", eigenvalues=(emin, emax))
    s.set_rhs_problem()
    it = s.solve()
    ref = orc.vcycle(dim, npts, levels, 3, 3, maxiter=60, ksp_type=1, emin=emin, emax=emax, use_csr=0)
    assert it == ref["iters"]
    rel = np.abs(s.rnorm - ref["rnorm"]) / ref["rnorm"]
    assert rel.max() <= RTOL
    assert np.array_equal(s.solution(), ref["u"])
    s.close()


def test_unfused_final_residual_same_history(orc):
    from multigrid_petsc_amd.solver import Solver
    a = Solver(3, 33, 4, scale=6.0 / 7.0, maxiter=50, fuse=7)
    b = Solver(3, 33, 4, scale=6.0 / 7.0, maxiter=50, fuse=0)
    for s in (a, b):
        s.set_rhs_problem()
        s.solve()
    assert a.iterations == b.iterations
    assert np.allclose(a.rnorm, b.rnorm, rtol=1e-13, atol=0)
    assert np.array_equal(a.solution(), b.solution())
    a.close()
    b.close()


@pytest.mark.parametrize("dim,npts,levels,v", [(3, 129, 7, (3, 3)), (3, 257, 8, (3, 2)), (2, 2049, 11, (3, 3)), (2, 129, 7, (2, 1))])
def test_hip_graph_replay_of_coarse_levels_is_bit_identical(orc, dim, npts, levels, v):
    from multigrid_petsc_amd.solver import Solver
    scale = 6.0 / 7.0 if dim == 3 else 0.8
    res = []
    for graph in (1, 0):
        s = Solver(dim, npts, levels, v=v, scale=scale, maxiter=40, graph=graph)
        s.set_rhs_problem()
        it = s.solve()
        first = (it, s.rnorm.copy(), s.solution())
        s.set_rhs_problem()                 # second solve on the same object: the recorded graph is replayed again
        it2 = s.solve()
        assert it2 == it and np.array_equal(s.rnorm, first[1]) and np.array_equal(s.solution(), first[2])
        res.append(first)
        s.close()
    assert res[0][0] == res[1][0]
    # (norms: the pass that forms them may differ between the two -- two sweeps per pass when level 0 does not feed the graph --
    # and with it the order of the sum; the fields are the same bits)
    assert np.allclose(res[0][1], res[1][1], rtol=1e-13, atol=0) and np.array_equal(res[0][2], res[1][2])
    if npts <= 129:
        ref = orc.vcycle(dim, npts, levels, v[0], v[1], maxiter=40, scale=scale)
        assert res[0][0] == ref["iters"] and np.array_equal(res[0][2], ref["u"])


def test_fixed_cycles_and_random_rhs(orc):
    """bench path: mg_solver_cycles from an arbitrary RHS"""
    from multigrid_petsc_amd.solver import Solver
    rng = np.random.default_rng(11)
    s = Solver(3, 17, 3, scale=0.8, maxiter=10)
    b = rng.uniform(-1, 1, s.local_unknowns)
    s.set_rhs(b)
    s.cycles(4)
    assert s.iterations == 4
    r = s.rnorm
    assert np.all(np.diff(r) < 0)          # contracts every cycle
    s.close()


@pytest.mark.parametrize("dim,npts,levels,v,scale", [
    (2, 17, 1, (3, 3), 0.8),        # one level: the cycle is just v0 sweeps per iteration (loops :1533-1544 are empty)
    (3, 9, 1, (2, 5), 0.8),
    (2, 5, 2, (3, 3), 0.8),         # 3x3 fine grid, one coarse unknown
    (3, 5, 2, (1, 1), 0.8),
    (2, 33, 3, (1, 2), 0.8),        # asymmetric -v
    (3, 17, 3, (2, 4), 6.0 / 7.0),
    (3, 17, 4, (4, 1), 6.0 / 7.0),
    (2, 65, 6, (2, 0), 0.8),        # no coarsest sweeps: KSPSolve with max_it 0 zero-fills the coarsest correction
    (3, 33, 5, (1, 0), 6.0 / 7.0),
])
def test_edge_configurations_match_oracle(orc, dim, npts, levels, v, scale):
    from multigrid_petsc_amd.solver import Solver
    maxiter = 40
    s = Solver(dim, npts, levels, v=v, maxiter=maxiter, scale=scale)
    s.set_rhs_problem()
    it = s.solve()
    ref = orc.vcycle(dim, npts, levels, v[0], v[1], maxiter=maxiter, scale=scale, use_csr=0)
    assert it == ref["iters"]
    rel = np.abs(s.rnorm - ref["rnorm"]) / ref["rnorm"]
    assert rel.max() <= RTOL
    assert np.array_equal(s.solution(), ref["u"])
    s.close()


def test_bad_arguments_fail_loudly():
    from multigrid_petsc_amd.solver import Solver, MgError
    with pytest.raises(MgError):
        Solver(3, 1000, 3)              # npts-1 not divisible by 2^(levels-1)
    with pytest.raises(MgError):
        Solver(4, 17, 2)                # dimension
    with pytest.raises(MgError):
        Solver(2, 17, 5)                # too many levels: the coarsest grid would be empty
    with pytest.raises(MgError):
        Solver(2, 17, 2, ksp_type="chebyshev")      # no eigenvalue bounds
    with pytest.raises(MgError):
        Solver(3, 17, 2, rank=0, nranks=2)          # ranks without a communicator


def test_own_c_driver_prints_reference_style_summary(orc, tmp_path):
    """multigrid_petsc_amd/mgpoisson: the product's C main (counterpart of src/poisson.c) with the reference's option spelling"""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multigrid_petsc_amd", "mgpoisson")
    (tmp_path / "poisson.in").write_text("# same syntax as the reference's options file\n-npts 65\n-levels 6\n-iter 100\n-v 3,3\n")
    p = subprocess.run([exe, "-dim", "3", "-ksp_richardson_scale", repr(6.0 / 7.0)], cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    ref = orc.vcycle(3, 65, 6, 3, 3, maxiter=100, scale=6.0 / 7.0)
    assert int(re.search(r"Number of iterations:\s+(\d+)", p.stdout).group(1)) == ref["iters"]
    rel = float(re.search(r"Relative residual = (\S+)", p.stdout).group(1))
    assert abs(rel - ref["rnorm"][-1] / ref["rnorm"][0]) <= 1e-12 * rel
    e0 = float(re.search(r"error\[0\] = (\S+)", p.stdout).group(1))
    assert e0 == orc.error_norms(3, 65, ref["u"])[0]
    rdat = np.array((tmp_path / "rData.dat").read_text().split(), dtype=np.float64)
    assert np.max(np.abs(rdat - ref["rnorm"] / ref["rnorm"][0]) / rdat) <= 1e-12


@pytest.mark.parametrize("dim,npts,levels", [(3, 65, 6), (2, 257, 8), (3, 33, 1)])
def test_speculative_first_sweep_changes_nothing(dim, npts, levels):
    """fuse bit 3: the kernel that evaluates ||r|| at the end of cycle k also makes the first sweep of cycle k+1.
    Same fields, same history; the solution left behind when the iteration stops is u, not the speculative sweep;
    fixed-count cycling (bench.py's path) agrees with the convergence-driven loop."""
    from multigrid_petsc_amd.solver import Solver
    scale = 6.0 / 7.0 if dim == 3 else 0.8
    res = {}
    for fuse in (7, 15):
        s = Solver(dim, npts, levels, v=(3, 3), maxiter=60, scale=scale, fuse=fuse)
        s.set_rhs_problem()
        it = s.solve()
        res[fuse] = (it, s.rnorm.copy(), s.solution())
        s.close()
    assert res[7][0] == res[15][0]
    assert np.abs(res[7][1] / res[15][1] - 1).max() <= 1e-13
    assert np.array_equal(res[7][2], res[15][2])
    s = Solver(dim, npts, levels, v=(3, 3), maxiter=60, scale=scale, fuse=15)
    s.set_rhs_problem()
    s.cycles(2)
    s.cycles(res[15][0] - 2)
    s.sync()
    assert np.array_equal(s.solution(), res[15][2]) and np.abs(s.rnorm / res[15][1] - 1).max() <= 1e-13
    s.close()


@pytest.mark.parametrize("dim,npts,levels", [(2, 257, 8), (2, 129, 3), (3, 65, 6), (3, 33, 2)])
def test_two_sweep_passes_change_nothing(dim, npts, levels):
    """fuse bit 5 with a low pair_min_n: pre- and post-smoothing sweeps 2 and 3 run as one pass on every level the
    coarse-level graph allows; fields and history equal the one-sweep-per-launch cycle"""
    from multigrid_petsc_amd.solver import Solver
    scale = 6.0 / 7.0 if dim == 3 else 0.8
    res = {}
    for fuse, pmin in ((31, 0), (63, 7)):
        s = Solver(dim, npts, levels, v=(3, 3), maxiter=60, scale=scale, fuse=fuse, pair_min_n=pmin)
        s.set_rhs_problem()
        it = s.solve()
        res[fuse] = (it, s.rnorm.copy(), s.solution())
        s.close()
    assert res[31][0] == res[63][0]
    assert np.abs(res[31][1] / res[63][1] - 1).max() <= 1e-13
    assert np.array_equal(res[31][2], res[63][2])


@pytest.mark.parametrize("npts,levels,v,kw", [
    (129, 6, (3, 3), {"pair_min_n": 63}), (257, 7, (3, 3), {"pair_min_n": 127}), (257, 7, (2, 2), {"pair_min_n": 127}),
    (257, 7, (3, 3), {"pair_min_n": 127, "graph": 0}), (257, 7, (2, 2), {"pair_min_n": 127, "graph": 0}),
    (257, 4, (4, 3), {"pair_min_n": 127, "graph": 0}), (257, 7, (5, 1), {"pair_min_n": 127, "graph": 0}), (129, 6, (1, 2), {}),
    (257, 8, (3, 3), {}), (129, 3, (3, 3), {"pair_min_n": 31}), (257, 8, (3, 3), {"graph": 0}),
])
def test_two_sweep_norm_pass_and_sweep_inside_the_restriction(orc, npts, levels, v, kw):
    """fuse bit 10 on shapes its kernels are built for (n = 127, 255): the pass that closes a cycle makes the norm and the first TWO
    sweeps of the next cycle (mgk_jacobi2_sumsq_f64), the last pre-smoothing sweep runs in the restriction's pass
    (mgk_sweep_residual_restrict_f64), for odd and even sweep counts, with and without the coarse-level graph: bit-identical to the
    oracle, and to the same cycle without bit 10"""
    from multigrid_petsc_amd.solver import Solver
    s = Solver(3, npts, levels, v=v, maxiter=80, scale=6.0 / 7.0, **kw)
    s.set_rhs_problem()
    it = s.solve()
    ref = orc.vcycle(3, npts, levels, v[0], v[1], maxiter=80, scale=6.0 / 7.0)
    assert it == ref["iters"]
    assert np.abs(s.rnorm / ref["rnorm"] - 1).max() <= RTOL
    u = s.solution()
    assert np.array_equal(u, ref["u"])
    s.close()
    s = Solver(3, npts, levels, v=v, maxiter=80, scale=6.0 / 7.0, fuse=63 | 256 | 512, **kw)
    s.set_rhs_problem()
    assert s.solve() == it and np.array_equal(s.solution(), u)
    s.close()


@pytest.mark.parametrize("npts,levels,v,kw", [
    (513, 9, (3, 3), {"pair_min_n": 127, "graph": 0}), (513, 9, (3, 3), {"pair_min_n": 127}), (1025, 10, (2, 2), {"pair_min_n": 255, "graph": 0}),
    (513, 4, (4, 3), {"pair_min_n": 63, "graph": 0}), (257, 8, (3, 3), {}), (4097, 12, (3, 3), {}),
])
def test_two_sweep_norm_pass_and_sweep_inside_the_restriction_2d(orc, npts, levels, v, kw, monkeypatch):
    """the 2-D forms (mgk_jacobi2_2d_sumsq_f64, mgk_sweep_residual_restrict_2d_f64) in the cycle: bit-identical to the oracle and to the
    same cycle without fuse bit 10 (the sweep inside the restriction on every level from 127^2 on, not only from 2047^2)"""
    from multigrid_petsc_amd.solver import Solver
    monkeypatch.setenv("MG_SRR2D_MIN_N", "127")
    s = Solver(2, npts, levels, v=v, maxiter=80, scale=0.8, **kw)
    s.set_rhs_problem()
    it = s.solve()
    u, rn = s.solution(), s.rnorm.copy()
    s.close()
    if npts <= 1025:
        ref = orc.vcycle(2, npts, levels, v[0], v[1], maxiter=80, scale=0.8)
        assert it == ref["iters"] and np.abs(rn / ref["rnorm"] - 1).max() <= RTOL and np.array_equal(u, ref["u"])
    s2 = Solver(2, npts, levels, v=v, maxiter=80, scale=0.8, fuse=63 | 256 | 512, **kw)
    s2.set_rhs_problem()
    assert s2.solve() == it and np.array_equal(s2.solution(), u) and np.allclose(s2.rnorm, rn, rtol=1e-13, atol=0)
    s2.close()


@pytest.mark.parametrize("dim,npts,levels,v", [(3, 33, 4, (1, 1)), (3, 33, 4, (2, 3)), (3, 65, 5, (4, 2)), (3, 65, 6, (5, 5)),
                                                 (2, 129, 6, (2, 1)), (2, 257, 7, (4, 4))])
def test_other_sweep_counts_with_all_fusions(orc, dim, npts, levels, v):
    """-v v0,v1 other than 3,3 with every fusion on (speculative first sweep, two-sweep passes from a low pair_min_n,
    fused transfers, graph replay): the pass/swap bookkeeping must hold for odd and even counts alike"""
    from multigrid_petsc_amd.solver import Solver
    scale = 6.0 / 7.0 if dim == 3 else 0.8
    s = Solver(dim, npts, levels, v=v, maxiter=80, scale=scale, pair_min_n=7)
    s.set_rhs_problem()
    it = s.solve()
    ref = orc.vcycle(dim, npts, levels, v[0], v[1], maxiter=80, scale=scale)
    assert it == ref["iters"]
    assert np.abs(s.rnorm / ref["rnorm"] - 1).max() <= RTOL
    assert np.array_equal(s.solution(), ref["u"])
    s.close()


@pytest.mark.parametrize("npts,levels,precision", [(33, 4, "fp64"), (65, 6, "fp64"), (65, 5, "mixed")])
def test_fused_residual_restriction_on_small_levels(orc, npts, levels, precision):
    """fuse bit 7 keeps the fused residual+restriction kernel on levels below 255^3 (by default two short kernels run
    there): same cycle bit for bit"""
    from multigrid_petsc_amd.solver import Solver
    s = Solver(3, npts, levels, maxiter=80, scale=6.0 / 7.0, fuse=63 | 128, precision=precision)
    s.set_rhs_problem()
    it = s.solve()
    ref = orc.vcycle(3, npts, levels, 3, 3, maxiter=80, scale=6.0 / 7.0) if precision == "fp64" else orc.vcycle_mixed(npts, levels, maxiter=80, scale=6.0 / 7.0)
    assert it == ref["iters"] and np.array_equal(s.solution(), ref["u"])
    assert np.abs(s.rnorm / ref["rnorm"] - 1).max() <= RTOL
    s.close()


@pytest.mark.parametrize("mesh,npts,levels,kw", [
    (1, 33, 4, {}), (2, 33, 4, {}), (1, 129, 6, {}), (2, 257, 7, {}), (1, 513, 8, {}), (2, 513, 9, {}),
    (1, 513, 8, {"pair_min_n": 63}), (2, 1025, 8, {"pair_min_n": 255}), (1, 257, 7, {"fuse": 0}), (2, 513, 9, {"graph": 0}),
    (1, 2049, 11, {}), (2, 2049, 11, {"graph": 0}), (1, 513, 8, {"pair_min_n": 63, "graph": 0}), (2, 513, 7, {"pair_min_n": 63, "graph": 0}),
])
def test_own_driver_on_stretched_meshes(orc, mesh, npts, levels, kw, monkeypatch):
    monkeypatch.setenv("MG_SRR2D_MIN_N", "63")       # the sweep inside the restriction on the small test levels too
    """SURVEY 8(f) N1 in the own driver: -mesh 1/2 (src/mesh.c:45-107,165-169) -- per-row coefficient tables on every level and
    the SAME fused cycle as on the uniform mesh on the row-table forms of its kernels (fused prolongation+sweep, residual+
    restriction+zero-guess sweep, norm+speculative sweep, two-sweep passes, the LDS tail, the coarse-level graph); fuse=0 is the
    kernel-per-operation cycle.  Bit-identical to the oracle's assembled stretched-mesh leg."""
    from multigrid_petsc_amd.solver import Solver
    s = Solver(2, npts, levels, v=(3, 3), maxiter=1000, scale=0.8, mesh=mesh, **kw)
    s.set_rhs_problem()
    it = s.solve()
    ref = orc.vcycle(2, npts, levels, 3, 3, maxiter=1000, scale=0.8, use_csr=1, mesh=mesh)
    assert it == ref["iters"] < 1000
    assert np.abs(s.rnorm / ref["rnorm"] - 1).max() <= RTOL
    assert np.array_equal(s.solution(), ref["u"])
    e, eref = s.error_norms(), orc.error_norms_mesh(npts, mesh, ref["u"])
    assert e[0] == eref[0] and abs(e[1] - eref[1]) <= RTOL * eref[1] and abs(e[2] - eref[2]) <= RTOL * eref[2]
    s.close()


@pytest.mark.parametrize("mesh,npts,levels", [(1, 65, 5), (2, 129, 6), (1, 513, 8), (2, 1025, 4)])
def test_own_driver_chebyshev_on_stretched_meshes(orc, mesh, npts, levels):
    """-mesh 1/2 with the Chebyshev smoother: the recurrence of the uniform mesh on the level's row tables (mgk_cheby_rowcoef_f64);
    bit-identical to the oracle's assembled stretched-mesh leg with its Chebyshev smoother"""
    from multigrid_petsc_amd.solver import Solver
    s = Solver(2, npts, levels, v=(3, 3), maxiter=200, ksp_type="chebyshev", eigenvalues=(0.2, 2.0), mesh=mesh)
    s.set_rhs_problem()
    it = s.solve()
    ref = orc.vcycle(2, npts, levels, 3, 3, maxiter=200, ksp_type=1, emin=0.2, emax=2.0, use_csr=1, mesh=mesh)
    assert it == ref["iters"]
    assert np.abs(s.rnorm / ref["rnorm"] - 1).max() <= RTOL
    assert np.array_equal(s.solution(), ref["u"])
    s.close()


@pytest.mark.parametrize("dim,npts,levels,precision", [
    (3, 33, 5, "fp64"), (3, 65, 6, "fp64"), (3, 65, 4, "fp64"), (3, 17, 4, "fp64"), (3, 129, 7, "fp64"),
    (2, 129, 7, "fp64"), (2, 65, 6, "fp64"), (2, 513, 9, "fp64"), (2, 129, 3, "fp64"), (3, 65, 6, "mixed"), (3, 129, 7, "mixed"),
])
def test_lds_tail_kernel_equals_the_per_operation_levels(dim, npts, levels, precision):
    """fuse bit 9: the levels with n <= 15 (3-D) / <= 63 (2-D) as ONE kernel with their fields in LDS (mgk_tail_cycle_*) against
    the same cycle with one launch per operation on those levels: iteration count, residual history and solution identical"""
    from multigrid_petsc_amd.solver import Solver
    scale = 6.0 / 7.0 if dim == 3 else 0.8
    res = []
    for fuse in (63 | 256, 63 | 256 | 512):
        s = Solver(dim, npts, levels, scale=scale, maxiter=60, precision=precision, fuse=fuse)
        s.set_rhs_problem()
        it = s.solve()
        res.append((it, s.rnorm.copy(), s.solution()))
        s.close()
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2])


def test_lds_tail_kernel_refuses_what_does_not_fit(mgk):
    import ctypes as C
    g = mgk.geom(3, 31)
    n = (C.c_int * 2)(31, 15)
    k7 = (C.c_double * 14)(*([1.0] * 14))
    di = (C.c_double * 2)(1.0, 1.0)
    f = mgk.field(g)
    assert mgk.L.mgk_tail_max_n(3) == 15 and mgk.L.mgk_tail_max_n(2) == 63
    assert mgk.L.mgk_tail_cycle_f64(mgk.ctx, C.byref(g), 2, n, k7, di, 1.0, 3, 3, f, f, None) != 0        # 31^3 does not fit in LDS
    g2 = mgk.geom(3, 15)
    n2 = (C.c_int * 2)(15, 6)
    assert mgk.L.mgk_tail_cycle_f64(mgk.ctx, C.byref(g2), 2, n2, k7, di, 1.0, 3, 3, f, f, None) != 0      # not a 2n+1 hierarchy
    mgk.free(f)
