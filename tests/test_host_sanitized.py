"""The product's host C -- csrc/petsc_shim.c, mg_solver.c, mg_comm.c, driver/mgpoisson.c: option parsing, MatSetValue -> CSR,
stencil recognition, PCMG set-up / tear-down, slab and ghost-plane bookkeeping, graph capture, deferred norms -- compiled with
-fsanitize=address,undefined over tests/mock_mgk.cpp (host-memory stand-ins for the kernel ABI, canonical arithmetic) and run
on the CPU.  Any sanitizer report fails the test; where the mock's arithmetic makes a comparison meaningful the results are
also checked against the oracle.  Test infrastructure only: nothing here is loaded by the product."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multigrid_petsc_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "_san")
REF = "/root/reference"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _cc(args):
    p = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, " ".join(args) + "\n" + p.stdout[-4000:]


@pytest.fixture(scope="module")
def san():
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no host compiler")
    os.makedirs(OUT, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include")]
    objs = []
    _cc(["g++", "-std=c++17", "-ffp-contract=off"] + SAN + inc + ["-c", os.path.join(ROOT, "tests", "mock_mgk.cpp"), "-o", os.path.join(OUT, "mock_mgk.o")])
    objs.append(os.path.join(OUT, "mock_mgk.o"))
    for f in ("petsc_shim", "mg_solver", "mg_comm"):
        o = os.path.join(OUT, f + ".o")
        _cc(["gcc", "-std=c99", "-ffp-contract=off", "-D_POSIX_C_SOURCE=200809L", "-Wall"] + SAN + inc + ["-c", os.path.join(CSRC, f + ".c"), "-o", o])
        objs.append(o)
    link = SAN + ["-lstdc++", "-lm", "-ldl", "-lpthread"]
    exes = {}
    exes["mgpoisson"] = os.path.join(OUT, "san_mgpoisson")
    _cc(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L"] + SAN + inc + [os.path.join(CSRC, "driver", "mgpoisson.c")] + objs + ["-o", exes["mgpoisson"]] + link)
    exes["slab"] = os.path.join(OUT, "san_slab")
    _cc(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L"] + SAN + inc + [os.path.join(ROOT, "tests", "san_slab.c")] + objs + ["-o", exes["slab"]] + link)
    if os.path.isdir(os.path.join(REF, "src")):
        robjs = []
        for f in ("array", "matbuild", "mesh", "problem", "solver", "poisson"):
            o = os.path.join(OUT, "ref_" + f + ".o")      # the reference's own sources, compiled where they lie, NOT sanitized (their
            _cc(["gcc", "-std=c99", "-O1", "-w", "-I" + os.path.join(REF, "include")] + inc + ["-c", os.path.join(REF, "src", f + ".c"), "-o", o])   # leaks are theirs)
            robjs.append(o)
        exes["refdriver"] = os.path.join(OUT, "san_refdriver")
        _cc(["gcc"] + robjs + objs + ["-o", exes["refdriver"]] + link)
    return exes


def _run(exe, args, cwd, env=None, ok_codes=(0,)):
    e = dict(ENV)
    e.update(env or {})
    p = subprocess.run([exe] + args, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert "ERROR: AddressSanitizer" not in p.stdout and "runtime error:" not in p.stdout, p.stdout[-6000:]
    assert p.returncode in ok_codes, (p.returncode, p.stdout[-4000:])
    return p.stdout


@pytest.mark.parametrize("args,dim,npts,levels,scale", [
    (["-dim", "2", "-npts", "33", "-levels", "4", "-ksp_richardson_scale", "0.8"], 2, 33, 4, 0.8),
    (["-dim", "2", "-npts", "129", "-levels", "7", "-ksp_richardson_scale", "0.8", "-map", "0"], 2, 129, 7, 0.8),
    (["-dim", "3", "-npts", "33", "-levels", "5", "-ksp_richardson_scale", "0.857142857142857095"], 3, 33, 5, 6.0 / 7.0),
    (["-dim", "3", "-npts", "17", "-levels", "2", "-ksp_richardson_scale", "0.857142857142857095"], 3, 17, 2, 6.0 / 7.0),
    # pairs of sweeps on every level: the norm pass makes two sweeps, the last pre-smoothing sweep runs in the restriction's pass
    (["-dim", "3", "-npts", "33", "-levels", "4", "-ksp_richardson_scale", "0.857142857142857095", "-mg_pair_min_n", "7"], 3, 33, 4, 6.0 / 7.0),
    (["-dim", "3", "-npts", "33", "-levels", "3", "-ksp_richardson_scale", "0.857142857142857095", "-mg_pair_min_n", "7", "-mg_graph", "0"], 3, 33, 3, 6.0 / 7.0),
    (["-dim", "3", "-npts", "33", "-levels", "5", "-ksp_richardson_scale", "0.857142857142857095", "-mg_fuse", "1087"], 3, 33, 5, 6.0 / 7.0),
])
def test_own_driver_under_sanitizers_matches_the_oracle(san, tmp_path, args, dim, npts, levels, scale):
    out = _run(san["mgpoisson"], args + ["-pc_type", "jacobi", "-write_fields", "1"], tmp_path, env={"MOCK_MGK_STATS": "1"})
    if "-mg_graph" in args and dim == 3:        # level 0 does not feed a graph: prolongation + two sweeps, norm of the mid iterate, final sweep
        m = re.search(r"pj2=(\d+) mid=(\d+)", out)
        assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0, out[-300:]
    ref = Oracle().vcycle(dim, npts, levels, 3, 3, maxiter=100000, scale=scale, use_csr=0)
    it = int(re.search(r"Number of iterations:\s+(\d+)", out).group(1))
    assert it == ref["iters"]
    rdat = np.array((tmp_path / "rData.dat").read_text().split(), dtype=np.float64)
    want = ref["rnorm"] / ref["rnorm"][0]
    assert np.max(np.abs(rdat - want) / want) <= 1e-12
    u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
    assert np.array_equal(u, ref["u"])


@pytest.mark.parametrize("prec,v0,fuse,pair,levels", [("mixed", 2, 7533, 31, 4), ("mixed", 4, 32, 31, 4), ("fp64", 2, 32, 7, 4), ("fp64", 2, 32 | 1024, 15, 5),
                                                      ("fp64", 4, 32 | 4 | 256, 7, 3)])
def test_sweep_groupings_that_swap_the_graph_feeding_level_an_odd_number_of_times(san, tmp_path, prec, v0, fuse, pair, levels):
    """Pairs of sweeps WITHOUT the fused prolongation and an even v0: the level that feeds the coarse-level recording swaps u / tmp three times per
    cycle, so the recorded restriction would read the stale buffer in every second cycle (found by tools/stress_solver.py, seed 11: max|du| 3.4e-9
    with equal iteration counts).  The solver now compares that level's pointers with the recorded ones and records again when they differ."""
    args = ["-dim", "3", "-npts", "33", "-levels", str(levels), "-v", f"{v0},3", "-ksp_richardson_scale", "0.857142857142857095", "-mg_fuse", str(fuse),
            "-mg_pair_min_n", str(pair), "-pc_type", "jacobi", "-write_fields", "1"] + (["-precision", "mixed"] if prec == "mixed" else [])
    out = _run(san["mgpoisson"], args, tmp_path)
    orc = Oracle()
    ref = (orc.vcycle_mixed(33, levels, v0, 3, maxiter=100000, scale=6.0 / 7.0) if prec == "mixed"
           else orc.vcycle(3, 33, levels, v0, 3, maxiter=100000, scale=6.0 / 7.0, use_csr=0))
    assert int(re.search(r"Number of iterations:\s+(\d+)", out).group(1)) == ref["iters"]
    u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
    assert np.array_equal(u, ref["u"].ravel())


@pytest.mark.parametrize("mesh,npts,levels,extra", [(1, 65, 5, []), (2, 33, 4, []), (1, 257, 8, []), (2, 257, 3, []),
                                                    (1, 129, 6, ["-mg_pair_min_n", "15", "-mg_graph", "0"]), (2, 129, 7, ["-mg_pair_min_n", "15"])])
def test_own_driver_stretched_mesh_under_sanitizers_matches_the_oracle(san, tmp_path, mesh, npts, levels, extra):
    """-mesh 1/2: the host logic of the FUSED stretched-mesh cycle (row-table forms of PJ / JNORM / fused restriction / tail)
    against the oracle's assembled stretched-mesh leg -- iteration count, history, field"""
    out = _run(san["mgpoisson"], ["-dim", "2", "-npts", str(npts), "-levels", str(levels), "-mesh", str(mesh), "-ksp_richardson_scale", "0.8",
                                  "-pc_type", "jacobi", "-write_fields", "1"] + extra, tmp_path, env={"MG_SRR2D_MIN_N": "15"})
    ref = Oracle().vcycle(2, npts, levels, 3, 3, maxiter=100000, scale=0.8, use_csr=1, mesh=mesh)
    assert int(re.search(r"Number of iterations:\s+(\d+)", out).group(1)) == ref["iters"]
    rdat = np.array((tmp_path / "rData.dat").read_text().split(), dtype=np.float64)
    want = ref["rnorm"] / ref["rnorm"][0]
    assert np.max(np.abs(rdat - want) / want) <= 1e-12
    u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
    assert np.array_equal(u, ref["u"])


@pytest.mark.parametrize("mesh,npts,levels", [(1, 65, 5), (2, 33, 4), (1, 129, 3)])
def test_own_driver_chebyshev_on_stretched_meshes_under_sanitizers_matches_the_oracle(san, tmp_path, mesh, npts, levels):
    """-mesh 1/2 with -ksp_type chebyshev: the three-term recurrence on the level's row tables (mgk_jacobi_zero_rowcoef, mgk_rowcoef mode 0,
    mgk_cheby_rowcoef) against the oracle's assembled stretched-mesh leg with its Chebyshev smoother"""
    out = _run(san["mgpoisson"], ["-dim", "2", "-npts", str(npts), "-levels", str(levels), "-mesh", str(mesh), "-ksp_type", "chebyshev",
                                  "-ksp_chebyshev_eigenvalues", "0.2,2.0", "-pc_type", "jacobi", "-write_fields", "1"], tmp_path)
    ref = Oracle().vcycle(2, npts, levels, 3, 3, maxiter=100000, ksp_type=1, emin=0.2, emax=2.0, use_csr=1, mesh=mesh)
    assert int(re.search(r"Number of iterations:\s+(\d+)", out).group(1)) == ref["iters"]
    rdat = np.array((tmp_path / "rData.dat").read_text().split(), dtype=np.float64)
    want = ref["rnorm"] / ref["rnorm"][0]
    assert np.max(np.abs(rdat - want) / want) <= 1e-12
    u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
    assert np.array_equal(u, ref["u"])


@pytest.mark.parametrize("args", [
    ["-dim", "3", "-npts", "33", "-levels", "4", "-precision", "mixed", "-ksp_richardson_scale", "0.857142857142857095"],
    ["-dim", "2", "-npts", "65", "-levels", "5", "-ksp_type", "chebyshev", "-ksp_chebyshev_eigenvalues", "0.2,2.0"],
    ["-dim", "3", "-npts", "33", "-levels", "4", "-ksp_type", "chebyshev", "-ksp_chebyshev_eigenvalues", "0.2,2.0"],
])
def test_own_driver_other_paths_under_sanitizers(san, tmp_path, args):
    out = _run(san["mgpoisson"], args + ["-pc_type", "jacobi"], tmp_path)
    assert "Number of iterations" in out and float(re.search(r"Relative residual = (\S+)", out).group(1)) < 1e-6


def test_own_driver_bad_options_under_sanitizers(san, tmp_path):
    for args in (["-levels", "2", "-grids", "3"], ["-map", "7"], ["-npts", "18"], ["-dim", "4"], ["-levels", "40"]):
        _run(san["mgpoisson"], args, tmp_path, ok_codes=(1, 2))
    (tmp_path / "poisson.in").write_text("# comment only\n-npts 17 # trailing\n-levels 2\n-v 3,3\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" + "-x y " * 300 + "\n")
    _run(san["mgpoisson"], ["-dim", "2"], tmp_path)            # more tokens than the option store holds: must not overflow


@pytest.mark.parametrize("P,npts,levels,dmin,extra", [(2, 33, 4, 15, []), (3, 33, 4, 15, []), (4, 65, 5, 15, []), (2, 33, 4, 15, ["mixed"]), (8, 65, 4, 31, [])])
def test_slab_ranks_under_sanitizers(san, tmp_path, P, npts, levels, dmin, extra):
    out = _run(san["slab"], [str(P), str(npts), str(levels), str(dmin)] + extra, tmp_path, env={"MOCK_MGK_STATS": "1"})
    assert f"SAN_SLAB_OK P={P}" in out
    if not extra:           # fp64: the slab forms of the norm + two sweeps pass and of the sweep inside the restriction were taken
        # (since the round's second session the default is the 91-byte scheme -- prolongation + two sweeps, mid-iterate norm -- wherever every slab has
        # >= 8 planes; thinner slabs (P = 8 at 63 planes) keep the 99-byte one)
        m = re.search(r"MOCK_MGK_STATS j2n=(\d+) srr=(\d+) j2n_slab=(\d+) srr_slab=(\d+) pj2=(\d+) mid=(\d+) pj2_slab=(\d+) mid_slab=(\d+)", out)
        assert m and ((int(m.group(3)) > 0 and int(m.group(4)) > 0) or (int(m.group(7)) > 0 and int(m.group(8)) > 0)), out[-400:]
        if P == 8:
            assert int(m.group(3)) > 0 and int(m.group(4)) > 0 and int(m.group(7)) == 0, out[-400:]
        else:
            assert int(m.group(7)) > 0 and int(m.group(8)) > 0, out[-400:]


@pytest.mark.parametrize("P,npts,levels,dmin", [(2, 33, 4, 15), (3, 33, 4, 15), (8, 65, 4, 31)])
def test_slab_ranks_over_the_peer_transport_under_sanitizers(san, tmp_path, P, npts, levels, dmin):
    """the peer transport's host logic (csrc/mg_comm.c: mailbox slots, flag words and their sequence numbers, gather box, all-reduce slots)
    with ranks as threads over the mock's host-memory primitives: transport self-test on every rank, then the slab solve equal to the
    single rank bit for bit, under ASan / UBSan"""
    out = _run(san["slab"], [str(P), str(npts), str(levels), str(dmin), "peer"], tmp_path)
    assert f"SAN_SLAB_OK P={P}" in out and "transport=peer" in out


def test_peer_transport_times_out_instead_of_hanging(san, tmp_path):
    """a neighbour that maps the memory and then never exchanges: with MG_PEER_TIMEOUT_S = 1 the flag waits give up, the calls return, and the
    next host-synchronising hook (all-reduce) and the `check` hook report MGK_ECOMM"""
    out = _run(san["slab"], ["peer_timeout"], tmp_path)
    assert "SAN_PEER_TIMEOUT_OK" in out


@pytest.mark.parametrize("P,bad", [(2, 1), (3, 1), (8, 5)])
def test_selftest_gate_returns_on_every_rank_when_one_plane_is_wrong(san, tmp_path, P, bad):
    """mg_comm_selftest must be collective-safe on FAILURE.  The loopback transport is told to hand rank `bad` a wrong
    lo ghost plane; every rank-thread must return from the gate (a rank that left the fixed sequence early would leave the others in
    a barrier: this test would time out), the bad rank with MGK_ECOMM and the first mismatch named, the others with 0."""
    out = _run(san["slab"], ["fault", str(P), str(bad)], tmp_path)
    assert f"SAN_FAULT_OK P={P} bad_rank={bad}" in out


# objects the REFERENCE's own code never releases (rv[0] is duplicated twice, src/solver.c:1460 and :1515; the PCMG and
# I-cycle drivers keep their work vectors): leaks of the caller, not of the drop-in -- everything else still counts
REF_LEAKS = "leak:MultigridVcycle\nleak:MultigridPetscPCMG\nleak:MultigridIcycle\nleak:SetUpSolver\nleak:SetUpPostProcess\n"


def _refdrv(san, tmp_path, opts, env=None):
    if "refdriver" not in san:
        pytest.skip("the reference tree is not on this machine")
    (tmp_path / "poisson.in").write_text(opts)
    (tmp_path / "lsan.supp").write_text(REF_LEAKS)
    e = dict(env or {})
    e["LSAN_OPTIONS"] = "suppressions=" + str(tmp_path / "lsan.supp") + ":print_suppressions=0"
    return _run(san["refdriver"], [], tmp_path, env=e)


@pytest.mark.parametrize("npts,levels,extra,env", [
    (17, 2, "", None), (65, 5, "", None), (33, 4, "", {"MGPETSC_NO_RECOGNITION": "1"}), (33, 4, "-mesh 1\n", None),
    (65, 5, "", {"MGPETSC_PAIR_MIN_N": "7"}),
])
def test_reference_driver_over_the_shim_under_sanitizers(san, tmp_path, npts, levels, extra, env):
    """the reference's unmodified main/Assemble/MultigridVcycle/Postprocessing drive the sanitized shim: MatSetValue staging,
    CSR build, recognition at MatAssemblyEnd, KSP set-up from the options, VecGetArray mirrors, Destroy* in the reference's order"""
    mesh = "-mesh 0\n" if "-mesh" not in extra else ""
    opts = (f"-npts {npts}\n{mesh}-iter 1000\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n"
            f"-pc_type jacobi\n-ksp_richardson_scale 0.8\n{extra}")
    out = _refdrv(san, tmp_path, opts, env)
    it = int(re.search(r"Number of iterations:\s+(\d+)", out).group(1))
    if not extra:
        ref = Oracle().vcycle(2, npts, levels, 3, 3, maxiter=1000, scale=0.8, use_csr=0)
        assert it == ref["iters"]
        u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
        assert np.array_equal(u, ref["u"])


@pytest.mark.parametrize("mesh,npts,levels,v0,v1", [(0, 65, 5, 3, 3), (1, 65, 5, 3, 3), (2, 33, 3, 3, 3),
                                                     # max_it = 0: the recurrence still takes the step that precedes its loop (oracle/mgo.c; round 3)
                                                     (0, 33, 3, 0, 2), (0, 33, 4, 2, 0), (1, 33, 3, 0, 1), (0, 17, 1, 0, 0)])
def test_reference_driver_chebyshev_under_sanitizers(san, tmp_path, mesh, npts, levels, v0, v1):
    """-ksp_type chebyshev through the reference's own -cycle 0 driver (uniform and stretched meshes): KSPSolve restarts the recurrence
    from the zero guess on every coarse level of every cycle, so the guess must really be zero-filled (a stale x was read as p_{k-1})"""
    out = _refdrv(san, tmp_path, f"-npts {npts}\n-mesh {mesh}\n-iter 200\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 0\n-v {v0},{v1}\n-moreNorm 0\n"
                  "-pc_type jacobi\n-ksp_type chebyshev\n-ksp_chebyshev_eigenvalues 0.2,2.0\n")
    ref = Oracle().vcycle(2, npts, levels, v0, v1, maxiter=200, ksp_type=1, emin=0.2, emax=2.0, use_csr=1, mesh=mesh)
    assert int(re.search(r"Number of iterations:\s+(\d+)", out).group(1)) == ref["iters"]
    u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
    assert np.array_equal(u, ref["u"])


def _pcmg_exact_coarse(orc, _dense, npts, levels, sweeps, scale, maxiter, rtol=1e-7):
    """numpy restatement of outer Richardson + PCMG V-cycle (the recursion in petsc_shim.c: mg_cycle) with Richardson + Jacobi
    level smoothers and an EXACT coarse solve (PETSc's default preonly + LU): residual history and solution"""
    A = [_dense(orc, "A", npts, l) for l in range(levels)]
    R = [_dense(orc, "R", npts, l) for l in range(levels - 1)]
    P = [_dense(orc, "P", npts, l) for l in range(levels - 1)]

    def smooth(l, b, x):
        d = 1.0 / np.diag(A[l])
        for _ in range(sweeps):
            x = x + scale * (d * (b - A[l] @ x))
        return x

    def cycle(l, b):
        if l == levels - 1:
            return np.linalg.solve(A[l], b)
        x = smooth(l, b, np.zeros_like(b))
        xc = cycle(l + 1, R[l] @ (b - A[l] @ x))
        return smooth(l, b, x + P[l] @ xc)

    b = orc.rhs(2, npts)
    x = np.zeros_like(b)
    hist = [np.linalg.norm(b)]
    while len(hist) - 1 < maxiter and hist[-1] > rtol * hist[0]:
        x = x + cycle(0, b - A[0] @ x)
        hist.append(np.linalg.norm(b - A[0] @ x))
    return np.array(hist), x


def _dense_from_oracle(orc, which, npts, l):
    m = orc.build(which, 2, npts, l)
    d = np.zeros((orc.L.mgo_csr_nrows(m), orc.L.mgo_csr_ncols(m)))
    for r, (cols, vals) in enumerate(orc.csr_rows(m)):
        d[r, list(cols)] = vals
    return d


@pytest.mark.parametrize("npts,levels,coarse", [(33, 3, "default"), (17, 2, "lu"), (33, 4, "lu"),
                                                (33, 5, "default"), (65, 6, "lu")])    # full depth: the coarsest grid is 1 x 1, where the exact solve is one undamped
                                                                                       # sweep and PCMG's levels from 63^2 down run as ONE tail launch (round 3)
def test_reference_driver_pcmg_exact_coarse_solve_under_sanitizers(san, tmp_path, npts, levels, coarse):
    """-cycle 8 with the exact coarse solve (PETSc's default preonly + LU; host-side inversion, dense mat-vec through the kernel ABI)
    against a dense numpy restatement with numpy.linalg.solve on the coarsest grid"""
    lv = "-mg_levels_ksp_type richardson\n-mg_levels_pc_type jacobi\n-mg_levels_ksp_max_it 3\n-mg_levels_ksp_richardson_scale 0.8\n"
    if coarse == "lu":
        lv += "-mg_coarse_ksp_type preonly\n-mg_coarse_pc_type lu\n"
    out = _refdrv(san, tmp_path, f"-npts {npts}\n-mesh 0\n-iter 100\n-grids {levels}\n-levels {levels}\n-cycle 8\n-map 2\n-v 3,3\n-moreNorm 0\n" + lv)
    orc = Oracle()
    hist, x = _pcmg_exact_coarse(orc, _dense_from_oracle, npts, levels, 3, 0.8, 100)
    assert int(re.search(r"Number of iterations:\s+(\d+)", out).group(1)) == len(hist) - 1
    u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
    assert np.max(np.abs(u - x)) <= 1e-10 * np.abs(x).max()
    assert "type: lu" in out
    if 2 ** levels == npts - 1:                              # full depth: one tail launch per application of the preconditioner, nothing computed twice
        d0 = tmp_path / "notail"
        d0.mkdir()
        out0 = _refdrv(san, d0, f"-npts {npts}\n-mesh 0\n-iter 100\n-grids {levels}\n-levels {levels}\n-cycle 8\n-map 2\n-v 3,3\n-moreNorm 0\n" + lv, {"MGPETSC_TAIL": "0"})
        assert (d0 / "uData.dat").read_text() == (tmp_path / "uData.dat").read_text()      # bit for bit what the level-by-level cycle gives


@pytest.mark.parametrize("npts,levels,v0,v1,mesh", [
    (33, 4, 1, 1, 0), (33, 4, 2, 2, 0), (33, 4, 4, 4, 0), (33, 4, 3, 1, 0), (33, 4, 1, 3, 0), (33, 4, 5, 2, 0), (33, 3, 3, 3, 0), (33, 2, 3, 3, 0), (33, 1, 3, 3, 0),
    (65, 5, 2, 4, 0), (65, 3, 4, 3, 0), (129, 6, 3, 3, 0), (129, 7, 4, 1, 0), (17, 3, 3, 3, 1), (33, 4, 2, 2, 2), (65, 5, 4, 4, 1), (65, 4, 3, 5, 2),
])
def test_reference_driver_sweep_counts_and_depths(san, tmp_path, npts, levels, v0, v1, mesh):
    """the reference's -v v0,v1 and -levels in combinations it is not usually run with: the drop-in's fast paths have preconditions on the sweep
    counts (three sweeps per pass from max_it >= 3, the norm pass without a stored r at max_it = 3, the tail recorder with one v0 above the
    coarsest level) -- whichever of them apply, the iteration count and the solution are the oracle's"""
    opts = (f"-npts {npts}\n-mesh {mesh}\n-iter 400\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 2\n-v {v0},{v1}\n-moreNorm 0\n"
            "-pc_type jacobi\n-ksp_richardson_scale 0.8\n")
    out = _refdrv(san, tmp_path, opts, {"MGPETSC_LAZY_STATS": "1"})
    it = int(re.search(r"Number of iterations:\s+(\d+)", out).group(1))
    ref = Oracle().vcycle(2, npts, levels, v0, v1, maxiter=400, scale=0.8, use_csr=1 if mesh else 0, mesh=mesh)
    assert it == ref["iters"], (it, ref["iters"])
    u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
    assert np.array_equal(u, ref["u"])
    tl = re.search(r"(\d+) coarse sub-cycles run as ONE tail launch, (\d+) recordings replayed", out)
    if levels >= 3 and (npts - 1) // 2 - 1 <= 63 and v0 >= 1:
        assert int(tl.group(1)) >= it - 1, out[-600:]           # the recorder did take the coarse levels (every cycle, or every cycle but the first)


@pytest.mark.parametrize("mesh,cycle", [(0, 0), (1, 0), (0, 8)])
def test_lazy_temporaries_of_the_reference_loop(san, tmp_path, mesh, cycle):
    """KSPBuildResidual -> MatMult(res) and MatMult(pro) -> VecAXPY -> KSPSolve of the reference's loop (src/solver.c:1531-1546) run as the
    fused residual + restriction and prolongation + sweep kernels; r and rv are never computed (dropped unread), results identical with
    MGPETSC_LAZY=0 (every call executed at once) and equal to the oracle"""
    levels, npts = 4, 65
    lv = ("-mg_levels_ksp_type richardson\n-mg_levels_pc_type jacobi\n-mg_levels_ksp_max_it 3\n-mg_levels_ksp_richardson_scale 0.8\n"
          "-mg_coarse_ksp_type richardson\n-mg_coarse_pc_type jacobi\n-mg_coarse_ksp_max_it 3\n-mg_coarse_ksp_richardson_scale 0.8\n") if cycle == 8 else ""
    opts = (f"-npts {npts}\n-mesh {mesh}\n-iter 1000\n-grids {levels}\n-levels {levels}\n-cycle {cycle}\n-map 0\n-v 3,3\n-moreNorm 0\n"
            + (lv if cycle == 8 else "-pc_type jacobi\n-ksp_richardson_scale 0.8\n"))
    res = {}
    # "1": the lazy temporaries alone (MGPETSC_TAIL=0: every level by its own launches), "0": every call executed at once,
    # "tail": the default -- the levels from 31^2 down also recorded and run as ONE tail launch per cycle (round 3)
    for lazy in ("1", "0", "tail"):
        d = tmp_path / lazy
        d.mkdir()
        out = _refdrv(san, d, opts, {"MGPETSC_LAZY": "1" if lazy == "tail" else lazy, "MGPETSC_TAIL": "1" if lazy == "tail" else "0", "MGPETSC_LAZY_STATS": "1"})
        it = int(re.search(r"Number of iterations:\s+(\d+)", out).group(1))
        m = re.search(r"lazy temporaries: (\d+) residual\+restriction passes, (\d+) prolongation sweeps fused; computed after all: (\d+) residuals, "
                      r"(\d+) prolongations, (\d+) corrections; (\d+) dropped unread; (\d+) zero-guess sweeps out of the restriction's pass; (\d+) norm passes that store r and make the next sweep, (\d+) of those sweeps adopted; "
                      r"(\d+) norm passes that left r deferred, (\d+) of those", out)
        assert m, out[-800:]
        st = [int(x) for x in m.groups()]
        tl = re.search(r"(\d+) coarse sub-cycles run as ONE tail launch, (\d+) recordings replayed call by call, (\d+) times their unread", out)
        tl = [int(x) for x in tl.groups()]
        if lazy == "tail":
            if cycle == 0:
                assert tl[0] == it and tl[1] == 0 and tl[2] <= 1, tl          # one launch per cycle, nothing replayed; the last cycle's unread intermediates
                                                                              # are computed once, when the reference destroys the solvers before the vectors
            else:
                assert tl[0] == it and tl[1] == 0 and tl[2] <= 1, tl          # PCMG's own cycle: its levels from 63^2 down (here: all of them) as one launch per application
        elif lazy == "1":
            assert tl == [0, 0, 0], tl
            assert st[0] == it * (levels - 1) and st[1] == it * (levels - 1), st      # every restriction and every first post-sweep fused
            assert st[3] == 0 and st[4] == 0, st                                      # rv never computed, no correction left over
            if cycle == 0:
                assert st[6] == it * (levels - 1), st                                 # every coarse pre-smoothing starts inside the restriction's pass
                assert st[7] + st[9] == it and st[2] == 0, st                         # every closing norm makes the next cycle's first sweeps (r stored, or -- max_it = 3 -- left deferred)
                assert st[8] == it - 1, st                                            # ... which every cycle but the first adopts
                assert st[9] == it and st[10] == it - 1, st                           # round 3: r never stored; it follows the old iterate into the work vector and is dropped unread
            else:
                assert st[2] <= it + 1, st                                            # PCMG: only the outer residual whose norm is monitored
        else:
            assert st == [0] * 11
        res[lazy] = (it, (d / "rData.dat").read_text(), (d / "uData.dat").read_text())
    assert res["1"][0] == res["0"][0] and res["1"][2] == res["0"][2]                  # cycle count, solution file
    assert res["tail"][0] == res["0"][0] and res["tail"][2] == res["0"][2] and res["tail"][1] == res["1"][1]
    r1, r0 = (np.array(res[q][1].split(), dtype=np.float64) for q in ("1", "0"))      # (the fused norm pass sums r^2 in another order)
    assert np.max(np.abs(r1 / r0 - 1)) <= 1e-12
    if cycle == 0:
        ref = Oracle().vcycle(2, npts, levels, 3, 3, maxiter=1000, scale=0.8, use_csr=1 if mesh else 0, mesh=mesh)
        assert res["1"][0] == ref["iters"]
        assert np.array_equal(np.array(res["1"][2].split(), dtype=np.float64), ref["u"])


def test_reference_driver_pcmg_and_icycle_under_sanitizers(san, tmp_path):
    lv = ("-mg_levels_ksp_type richardson\n-mg_levels_pc_type jacobi\n-mg_levels_ksp_max_it 3\n-mg_levels_ksp_richardson_scale 0.8\n"
          "-mg_coarse_ksp_type richardson\n-mg_coarse_pc_type jacobi\n-mg_coarse_ksp_max_it 3\n-mg_coarse_ksp_richardson_scale 0.8\n")
    out = _refdrv(san, tmp_path, "-npts 33\n-mesh 0\n-iter 200\n-grids 4\n-levels 4\n-cycle 8\n-map 2\n-v 3,3\n-moreNorm 0\n" + lv)
    assert "Number of iterations" in out
    d2 = tmp_path / "c1"
    d2.mkdir()
    out = _refdrv(san, d2, "-npts 17\n-mesh 0\n-iter 50\n-grids 1\n-levels 1\n-cycle 1\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n")
    assert "Number of iterations" in out
    # several grids in one level: the coupled operator (A_g on the diagonal, R^k A_g below, A_g P^k cut to P's window above;
    # src/solver.c:255-487) is recognised at MatAssemblyEnd and applied by the stencil / transfer kernels on composite vectors; with
    # recognition off it stays an assembled AIJ matrix on the generic CSR kernel -- same residual history and solution up to summation order
    res = {}
    cases = ((17, 2, 5), (9, 2, 3), (5, 2, 2), (17, 3, 4), (33, 4, 3), (33, 3, 3))
    for tag, env in (("c1b", None), ("c1c", {"MGPETSC_NO_RECOGNITION": "1"})):
        d3 = tmp_path / tag
        d3.mkdir()
        for npts, grids, it in cases:
            out = _refdrv(san, d3, f"-npts {npts}\n-mesh 0\n-iter {it}\n-grids {grids}\n-levels 1\n-cycle 1\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.5\n", env)
            assert ("matrix-free level operator of several grids" in out) == (env is None), out[-600:]
            assert ("assembled AIJ (generic CSR kernel)" in out) == (env is not None)
            res[tag, npts, grids] = (np.array((d3 / "rData.dat").read_text().split(), dtype=np.float64), np.array((d3 / "uData.dat").read_text().split(), dtype=np.float64))
    for npts, grids, it in cases:
        (r1, u1), (r2, u2) = res["c1b", npts, grids], res["c1c", npts, grids]
        assert r1.shape == r2.shape and np.max(np.abs(r1 - r2)) <= 1e-12 * np.abs(r2).max()
        assert np.max(np.abs(u1 - u2)) <= 1e-12 * np.abs(u2).max()
