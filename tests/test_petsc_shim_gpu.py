"""GPU parity of the PETSc-surface drop-in (include/petscksp.h, libmgpetsc.so).

1. ctypes leg: the call sequence of Assemble + MultigridVcycle (src/solver.c:1156-1209,1414-1575) issued through
   the shim's own entry points (MatCreateAIJ/MatSetValue/.../KSPSolve/KSPBuildResidual/MatMult/VecAXPY/VecNorm),
   with stencil recognition on (matrix-free kernels) and off (generic AIJ kernel): both must reproduce the oracle.
2. reference-driver leg: the reference's UNMODIFIED src/*.c, linked against the drop-in by __graft_entry__.build()
   (build/refdriver/poisson), run on the GPU with an options file of ours; its outputs (rData.dat, uData.dat,
   eData.dat, iteration count) are compared with the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import Oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDRV = os.path.join(ROOT, "build", "refdriver", "poisson")
ADD, INSERT, FINAL, NORM_2 = 2, 1, 0, 1


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def _shim():
    from multigrid_petsc_amd._lib import load_mgpetsc
    L = load_mgpetsc()
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    L.PetscInitialize.argtypes = [vp, vp, C.c_char_p, C.c_char_p]
    L.PetscOptionsSetValue.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.MatCreateAIJ.argtypes = [i, i, i, i, i, i, vp, i, vp, C.POINTER(vp)]
    L.MatSetValue.argtypes = [vp, i, i, d, i]
    L.MatAssemblyBegin.argtypes = [vp, i]
    L.MatAssemblyEnd.argtypes = [vp, i]
    L.MatCreateVecs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.MatMult.argtypes = [vp, vp, vp]
    L.MatDestroy.argtypes = [C.POINTER(vp)]
    L.VecSetValue.argtypes = [vp, i, d, i]
    L.VecAssemblyBegin.argtypes = [vp]
    L.VecAssemblyEnd.argtypes = [vp]
    L.VecDuplicate.argtypes = [vp, C.POINTER(vp)]
    L.VecSet.argtypes = [vp, d]
    L.VecAXPY.argtypes = [vp, d, vp]
    L.VecNorm.argtypes = [vp, i, C.POINTER(d)]
    L.VecGetArray.argtypes = [vp, C.POINTER(C.POINTER(d))]
    L.VecRestoreArray.argtypes = [vp, C.POINTER(C.POINTER(d))]
    L.VecDestroy.argtypes = [C.POINTER(vp)]
    L.KSPCreate.argtypes = [i, C.POINTER(vp)]
    L.KSPSetType.argtypes = [vp, C.c_char_p]
    L.KSPSetOperators.argtypes = [vp, vp, vp]
    L.KSPSetNormType.argtypes = [vp, i]
    L.KSPSetTolerances.argtypes = [vp, d, d, d, i]
    L.KSPSetFromOptions.argtypes = [vp]
    L.KSPSetInitialGuessNonzero.argtypes = [vp, i]
    L.KSPSolve.argtypes = [vp, vp, vp]
    L.KSPBuildResidual.argtypes = [vp, vp, vp, C.POINTER(vp)]
    L.KSPDestroy.argtypes = [C.POINTER(vp)]
    L.MatView.argtypes = [vp, vp]
    return L


def _vcycle_through_shim(L, orc, npts, levels, scale, maxiter):
    """Assemble (2-D) and cycle exactly as the reference's -cycle 0 path does, through the drop-in's API."""
    vp = C.c_void_p
    n = [(npts - 1) // 2 ** l - 1 for l in range(levels)]
    A, u, b, R, P = [], [], [], [], []
    for l in range(levels):
        As, _ = orc.level_stencil(2, npts, l)
        m = vp()
        N = n[l] * n[l]
        L.MatCreateAIJ(1, N, N, -1, -1, 6, None, 6, None, C.byref(m))
        for row in range(N):
            i0, j0 = divmod(row, n[l])
            if i0 - 1 >= 0:
                L.MatSetValue(m, row, row - n[l], As[0], ADD)
            if j0 - 1 >= 0:
                L.MatSetValue(m, row, row - 1, As[1], ADD)
            L.MatSetValue(m, row, row, As[2], ADD)
            if j0 + 1 < n[l]:
                L.MatSetValue(m, row, row + 1, As[3], ADD)
            if i0 + 1 < n[l]:
                L.MatSetValue(m, row, row + n[l], As[4], ADD)
        L.MatAssemblyBegin(m, FINAL)
        L.MatAssemblyEnd(m, FINAL)
        uu, bb = vp(), vp()
        L.MatCreateVecs(m, C.byref(uu), C.byref(bb))
        A.append(m); u.append(uu); b.append(bb)
    rhs = orc.rhs(2, npts)
    for row, val in enumerate(rhs):
        L.VecSetValue(b[0], row, val, INSERT)
    L.VecAssemblyBegin(b[0]); L.VecAssemblyEnd(b[0])
    wr, wp = np.zeros(9), np.zeros(9)
    orc.L.mgo_restriction_stencil(wr.ctypes.data)
    orc.L.mgo_prolongation_stencil(wp.ctypes.data)
    for l in range(levels - 1):
        nf, nc = n[l], n[l + 1]
        r, p = vp(), vp()
        L.MatCreateAIJ(1, nc * nc, nf * nf, -1, -1, 9, None, 9, None, C.byref(r))
        L.MatCreateAIJ(1, nf * nf, nc * nc, -1, -1, 4, None, 4, None, C.byref(p))
        for c in range(nc * nc):
            i1, j1 = divmod(c, nc)
            for di in range(3):
                for dj in range(3):
                    f = (2 * i1 + di) * nf + 2 * j1 + dj
                    L.MatSetValue(r, c, f, wr[di * 3 + dj], ADD)
                    L.MatSetValue(p, f, c, wp[di * 3 + dj], ADD)
        for m in (r, p):
            L.MatAssemblyBegin(m, FINAL); L.MatAssemblyEnd(m, FINAL)
        R.append(r); P.append(p)
    # MultigridVcycle
    rv = []
    for l in range(levels):
        t = vp(); L.VecDuplicate(b[l], C.byref(t)); rv.append(t)
    ksp = []
    for l in range(levels):
        k = vp(); L.KSPCreate(1, C.byref(k))
        L.KSPSetType(k, b"richardson"); L.KSPSetOperators(k, A[l], A[l]); L.KSPSetNormType(k, 0)
        L.KSPSetTolerances(k, 1e-7, -2.0, -2.0, 3)
        L.KSPSetFromOptions(k)
        ksp.append(k)
    val = C.c_double()
    L.VecNorm(b[0], NORM_2, C.byref(val)); bnorm = val.value
    L.VecSet(u[0], 0.0)
    L.MatMult(A[0], u[0], rv[0]); L.VecAXPY(rv[0], -1.0, b[0])
    L.VecNorm(rv[0], NORM_2, C.byref(val)); rn = [val.value]
    it = 0
    V = vp()
    while it < maxiter and 1e8 * bnorm > rn[-1] and rn[-1] > 1e-7 * bnorm:
        L.KSPSolve(ksp[0], b[0], u[0])
        if it == 0:
            L.KSPSetInitialGuessNonzero(ksp[0], 1)
        for l in range(1, levels):
            L.KSPBuildResidual(ksp[l - 1], None, rv[l - 1], C.byref(V))
            L.MatMult(R[l - 1], V, b[l])
            L.KSPSolve(ksp[l], b[l], u[l])
            if l != levels - 1:
                L.KSPSetInitialGuessNonzero(ksp[l], 1)
        for l in range(levels - 2, -1, -1):
            L.MatMult(P[l], u[l + 1], rv[l]); L.VecAXPY(u[l], 1.0, rv[l])
            L.KSPSolve(ksp[l], b[l], u[l])
            if l != 0:
                L.KSPSetInitialGuessNonzero(ksp[l], 0)
        L.KSPBuildResidual(ksp[0], None, rv[0], C.byref(V))
        L.VecNorm(V, NORM_2, C.byref(val)); rn.append(val.value)
        it += 1
    px = C.POINTER(C.c_double)()
    L.VecGetArray(u[0], C.byref(px))
    sol = np.array([px[q] for q in range(n[0] * n[0])])
    L.VecRestoreArray(u[0], C.byref(px))
    for lst in (ksp,):
        for h in lst:
            L.KSPDestroy(C.byref(h))
    for h in rv + u + b:
        L.VecDestroy(C.byref(h))
    for h in A + R + P:
        L.MatDestroy(C.byref(h))
    return it, np.array(rn), sol, bnorm


@pytest.mark.parametrize("recognise", [True, False])
@pytest.mark.parametrize("npts,levels,scale", [(17, 2, 0.8), (17, 4, 0.8), (33, 3, 1.0)])
def test_shim_call_sequence_matches_oracle(orc, recognise, npts, levels, scale):
    L = _shim()
    L.PetscInitialize(None, None, None, None)
    L.PetscOptionsSetValue(None, b"-pc_type", b"jacobi")
    L.PetscOptionsSetValue(None, b"-ksp_richardson_scale", repr(scale).encode())
    if recognise:
        os.environ.pop("MGPETSC_NO_RECOGNITION", None)
    else:
        os.environ["MGPETSC_NO_RECOGNITION"] = "1"
    try:
        it, rn, sol, bnorm = _vcycle_through_shim(L, orc, npts, levels, scale, 150)
    finally:
        os.environ.pop("MGPETSC_NO_RECOGNITION", None)
    ref = orc.vcycle(2, npts, levels, 3, 3, maxiter=150, scale=scale, use_csr=1)
    assert it == ref["iters"]
    assert abs(bnorm - ref["bnorm"]) <= 1e-12 * ref["bnorm"]
    assert np.max(np.abs(rn - ref["rnorm"]) / ref["rnorm"]) <= 1e-12
    assert np.array_equal(sol, ref["u"])


def _run_reference_driver(tmp_path, opts, extra_env=None):
    (tmp_path / "poisson.in").write_text("# options of the reference driver (same keys as its poisson.in)\n" + opts)
    env = dict(os.environ)
    env.update(extra_env or {})
    p = subprocess.run([REFDRV], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    out = p.stdout
    it = int(re.search(r"Number of iterations:\s+(\d+)", out).group(1))
    rdat = np.array((tmp_path / "rData.dat").read_text().split(), dtype=np.float64)
    u = np.array((tmp_path / "uData.dat").read_text().split(), dtype=np.float64)
    e = np.array((tmp_path / "eData.dat").read_text().split(), dtype=np.float64)
    return it, rdat, u, e, out


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent: __graft_entry__.build() links it only "
                    "where /root/reference exists (the build container); the binary then travels with the snapshot")
@pytest.mark.parametrize("npts,levels,scale,env", [
    (17, 2, 0.8, None), (17, 2, 1.0, None), (129, 7, 0.8, None), (129, 6, 0.8, None), (513, 9, 0.8, None),
    (33, 4, 0.8, {"MGPETSC_NO_RECOGNITION": "1"}),
    (257, 8, 0.8, {"MGPETSC_PAIR_MIN_N": "7"}),          # KSPSolve runs its sweeps two per pass
])
def test_unmodified_reference_driver_on_the_gpu(orc, tmp_path, npts, levels, scale, env):
    opts = (f"-npts {npts}\n-mesh 0\n-iter 1000\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n"
            f"-pc_type jacobi\n-ksp_richardson_scale {scale!r}\n")
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts, env)
    ref = orc.vcycle(2, npts, levels, 3, 3, maxiter=1000, scale=scale, use_csr=0)
    assert it == ref["iters"]
    want = ref["rnorm"] / ref["rnorm"][0]                # solver.c:1554-1557 normalises by rnorm[0]
    assert rdat.size == it + 1
    assert np.max(np.abs(rdat - want) / want) <= 1e-12
    assert np.array_equal(u, ref["u"])                   # %.16e round-trips a double
    eref = orc.error_norms(2, npts, ref["u"])
    assert np.allclose(e, eref, rtol=1e-12, atol=0)
    assert "matrix-free 5-point stencil" in out or env  # KSPView reports the recognised operator


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
@pytest.mark.parametrize("npts,levels,v0,v1,mesh", [
    (33, 4, 1, 1, 0), (65, 5, 2, 4, 0), (129, 7, 4, 1, 0), (257, 8, 4, 4, 0), (513, 9, 5, 2, 0), (257, 5, 3, 3, 0), (129, 6, 2, 2, 2), (257, 8, 4, 3, 1),
])
def test_reference_driver_sweep_counts_and_depths_on_the_gpu(orc, tmp_path, npts, levels, v0, v1, mesh):
    """-v and -levels in combinations the reference is not usually run with (the drop-in's fast paths have preconditions on the sweep counts;
    the CPU tier runs a larger set of these over the mock): iteration count and solution are the oracle's"""
    opts = (f"-npts {npts}\n-mesh {mesh}\n-iter 400\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 2\n-v {v0},{v1}\n-moreNorm 0\n"
            "-pc_type jacobi\n-ksp_richardson_scale 0.8\n")
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts)
    ref = orc.vcycle(2, npts, levels, v0, v1, maxiter=400, scale=0.8, use_csr=1 if mesh else 0, mesh=mesh)
    assert it == ref["iters"]
    assert np.array_equal(u, ref["u"])


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
def test_reference_driver_default_options_file(orc, tmp_path):
    """the reference's default run (poisson.in: npts 17, 2 grids, 2 levels, V(3,3), no -pc_type): PETSc would use ILU(0);
    the drop-in says so on stderr and smooths with Jacobi, scale 1 -> 99 cycles."""
    opts = "-npts 17\n-mesh 0\n-iter 100000\n-grids 2\n-levels 2\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n"
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts)
    ref = orc.vcycle(2, 17, 2, 3, 3, maxiter=100000, scale=1.0, use_csr=0)
    assert it == ref["iters"] == 99
    assert "default preconditioner (ILU(0)) is not provided" in out                    # stderr note
    assert "Jacobi substituted for PETSc's default ILU(0)" in out                      # and in the stdout report
    assert np.array_equal(u, ref["u"])
    # MGPETSC_DEFAULT_PC=jacobi opts in silently; =refuse stops the run instead of printing PETSc-looking numbers
    it2, _, u2, _, out2 = _run_reference_driver(tmp_path, opts, {"MGPETSC_DEFAULT_PC": "jacobi"})
    assert it2 == 99 and np.array_equal(u2, u) and "ILU(0)" not in out2.split("KSP Object")[0]
    env = dict(os.environ, MGPETSC_DEFAULT_PC="refuse")
    p = subprocess.run([REFDRV], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 2 and "MGPETSC_DEFAULT_PC=refuse" in p.stdout


def test_own_driver_names_the_mapping_style_it_was_given(tmp_path):
    """src/poisson.c:190-192: the PrintInfo block names the -map style (with one grid per level all three give the same map)"""
    for m, name in ((0, "Grid after grid"), (1, "Through the grids"), (2, "Local grid after grid")):
        p = subprocess.run([MGPOISSON, "-dim", "2", "-npts", "17", "-levels", "2", "-grids", "2", "-pc_type", "jacobi",
                            "-ksp_richardson_scale", "0.8", "-map", str(m)], cwd=tmp_path, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=300)
        assert p.returncode == 0, p.stdout
        line = [l for l in p.stdout.splitlines() if l.startswith("Mapping style")][0]
        assert line.split(":")[1].strip() == name


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
@pytest.mark.parametrize("mesh,npts,levels,env", [(1, 33, 4, None), (2, 33, 4, None), (1, 129, 6, None), (2, 65, 5, None), (1, 513, 8, None),
                                                  (1, 65, 5, {"MGPETSC_NO_RECOGNITION": "1"})])
def test_reference_driver_stretched_meshes(orc, tmp_path, mesh, npts, levels, env):
    """-mesh 1/2 make the 5 coefficients depend on y (src/mesh.c:45-107, src/problem.c:17-21).
    MatAssemblyEnd recognises "5-point rows, coefficients constant along a grid row" and runs the same marching kernel
    with a per-row coefficient table; with recognition switched off the assembled AIJ kernels run instead.  Same parity
    bar as -mesh 0 either way."""
    scale = 0.8
    opts = (f"-npts {npts}\n-mesh {mesh}\n-iter 1000\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 0\n-v 3,3\n-moreNorm 0\n"
            f"-pc_type jacobi\n-ksp_richardson_scale {scale!r}\n")
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts, env)
    ref = orc.vcycle(2, npts, levels, 3, 3, maxiter=1000, scale=scale, use_csr=1, mesh=mesh)
    assert it == ref["iters"]
    want = ref["rnorm"] / ref["rnorm"][0]
    assert np.max(np.abs(rdat - want) / want) <= 1e-12
    assert np.array_equal(u, ref["u"])
    assert np.allclose(e, orc.error_norms_mesh(npts, mesh, ref["u"]), rtol=1e-12, atol=0)
    assert ("assembled AIJ (generic CSR kernel)" if env else "row-dependent coefficients") in out


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
@pytest.mark.parametrize("mesh,npts,levels", [(0, 65, 5), (0, 257, 7), (1, 65, 5), (2, 129, 6), (1, 513, 8)])
def test_reference_driver_chebyshev_on_stretched_meshes(orc, tmp_path, mesh, npts, levels):
    """KSPCHEBYSHEV + PCJACOBI through the unmodified reference driver over the drop-in: on the constant stencil (-mesh 0) and on
    the row-table operator of -mesh 1/2.  (Every coarse-level KSPSolve restarts the recurrence from a zero guess that must really
    be zero-filled: a stale x was once read as p_{k-1}, which only this test sees -- PCMG zero-fills x itself.)"""
    opts = (f"-npts {npts}\n-mesh {mesh}\n-iter 200\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 0\n-v 3,3\n-moreNorm 0\n"
            "-pc_type jacobi\n-ksp_type chebyshev\n-ksp_chebyshev_eigenvalues 0.2,2.0\n")
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts)
    ref = orc.vcycle(2, npts, levels, 3, 3, maxiter=200, ksp_type=1, emin=0.2, emax=2.0, use_csr=1, mesh=mesh)
    assert it == ref["iters"]
    want = ref["rnorm"] / ref["rnorm"][0]
    assert np.max(np.abs(rdat - want) / want) <= 1e-12
    assert np.array_equal(u, ref["u"])
    assert ("row-dependent coefficients" if mesh else "matrix-free 5-point stencil") in out


MGPOISSON = os.path.join(ROOT, "multigrid_petsc_amd", "mgpoisson")


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
@pytest.mark.parametrize("npts,levels,mesh", [(17, 2, 0), (65, 5, 0), (129, 7, 0), (65, 5, 1), (129, 6, 2)])
def test_own_driver_writes_the_reference_output_files(tmp_path, npts, levels, mesh):
    """the product's own driver against the reference's own Postprocessing code (unmodified
    src/solver.c:1317-1380 running over the shim): uData.dat, XgridData.dat, YgridData.dat byte for byte;
    eData.dat / rData.dat number for number (their sums are reduced in another order: 1e-12); PrintInfo lines."""
    opts = (f"-npts {npts}\n-mesh {mesh}\n-iter 1000\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n"
            f"-pc_type jacobi\n-ksp_richardson_scale 0.8\n")
    a, b = tmp_path / "ref", tmp_path / "own"
    a.mkdir()
    b.mkdir()
    it, rdat, u, e, out_ref = _run_reference_driver(a, opts)
    (b / "poisson.in").write_text(opts)
    p = subprocess.run([MGPOISSON, "-dim", "2"], cwd=b, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:]
    for name in ("uData.dat", "XgridData.dat", "YgridData.dat"):
        assert (a / name).read_bytes() == (b / name).read_bytes(), name
    for name in ("eData.dat", "rData.dat"):
        x = np.array((a / name).read_text().split(), dtype=np.float64)
        y = np.array((b / name).read_text().split(), dtype=np.float64)
        assert x.shape == y.shape and np.max(np.abs(x - y) / np.abs(x)) <= 1e-12, name
    assert (a / "eData.dat").read_text().count("\n") == (b / "eData.dat").read_text().count("\n") == 3

    def info(txt):
        blk = txt[txt.index("====="):]
        return [ln for ln in blk.splitlines() if ln.split(":")[0].strip() in
                ("Size", "Mesh Type", "Number of grids", "Number of levels", "Number of grids per level",
                 "Number of unknowns per level", "Mapping style", "Cycle", "Number of smoothing steps",
                 "Number of processes", "Number of iterations")]
    assert info(out_ref) == info(p.stdout)


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
@pytest.mark.parametrize("npts,levels,scale,ksp", [(17, 2, 1.0, "richardson"), (33, 5, 0.8, "richardson"), (129, 7, 0.8, "richardson"),
                                                   (513, 9, 0.8, "richardson"), (65, 5, None, "chebyshev")])
def test_reference_driver_cycle8_pcmg(orc, tmp_path, npts, levels, scale, ksp):
    """-cycle 8 (MultigridPetscPCMG, src/solver.c:1884-1989) -- the reference's own code drives
    PCMG through the drop-in: outer Richardson + multiplicative V-cycle with -mg_levels_* / -mg_coarse_* solvers.
    Checked against the oracle's restatement of that recursion (bit-identical fields) -- parity with PETSc itself
    is unpinned (PCMG internals are version dependent; see petsc_shim.c)."""
    lv = (f"-mg_levels_ksp_type {ksp}\n-mg_levels_pc_type jacobi\n-mg_levels_ksp_max_it 3\n"
          f"-mg_coarse_ksp_type {ksp}\n-mg_coarse_pc_type jacobi\n-mg_coarse_ksp_max_it 3\n")
    if ksp == "richardson":
        lv += f"-mg_levels_ksp_richardson_scale {scale!r}\n-mg_coarse_ksp_richardson_scale {scale!r}\n"
        kw = dict(scale=scale)
    else:
        lv += "-mg_levels_ksp_chebyshev_eigenvalues 0.2,2.0\n-mg_coarse_ksp_chebyshev_eigenvalues 0.2,2.0\n"
        kw = dict(ksp_type=1, emin=0.2, emax=2.0)
    opts = f"-npts {npts}\n-mesh 0\n-iter 400\n-grids {levels}\n-levels {levels}\n-cycle 8\n-map 2\n-v 3,3\n-moreNorm 0\n" + lv
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts)
    ref = orc.pcmg(2, npts, levels, 3, 3, maxiter=400, **kw)
    assert it == ref["iters"] < 400
    want = ref["rnorm"] / ref["rnorm"][0]
    assert rdat.size == it + 1
    assert np.max(np.abs(rdat - want) / want) <= 1e-12
    assert np.array_equal(u, ref["u"])
    assert "Petsc-V-Cycle" in out and "type: mg" in out
    if ksp == "richardson":          # and it is the -cycle 0 iteration in correction form
        vc = orc.vcycle(2, npts, levels, 3, 3, maxiter=400, scale=scale)
        assert vc["iters"] == it


def _dense(orc, which, npts, l):
    m = orc.build(which, 2, npts, l)
    rows = orc.csr_rows(m)
    nr, nc = orc.L.mgo_csr_nrows(m), orc.L.mgo_csr_ncols(m)
    d = np.zeros((nr, nc))
    for r, (cols, vals) in enumerate(rows):
        d[r, list(cols)] = vals
    return d


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


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
@pytest.mark.parametrize("npts,levels,coarse", [(33, 3, "default"), (33, 2, "default"), (65, 3, "lu"), (65, 5, "default"), (129, 3, "lu")])
def test_reference_driver_cycle8_exact_coarse_solve(orc, tmp_path, npts, levels, coarse):
    """-cycle 8 with PETSc's default coarse solver (preonly + LU, src/solver.c:1931-1932 takes it as it comes): the drop-in inverts
    the coarsest operator once on the host and applies the dense inverse on the GPU (mgk_dense_mult_f64) -- an exact solve, given
    explicitly (-mg_coarse_ksp_type preonly -mg_coarse_pc_type lu) or by default.  Against a dense numpy restatement of the cycle
    with numpy.linalg.solve on the coarsest grid: same cycle count, history and solution to 1e-10 (parity with PETSc: unpinned)."""
    lv = "-mg_levels_ksp_type richardson\n-mg_levels_pc_type jacobi\n-mg_levels_ksp_max_it 3\n-mg_levels_ksp_richardson_scale 0.8\n"
    if coarse == "lu":
        lv += "-mg_coarse_ksp_type preonly\n-mg_coarse_pc_type lu\n"
    opts = f"-npts {npts}\n-mesh 0\n-iter 100\n-grids {levels}\n-levels {levels}\n-cycle 8\n-map 2\n-v 3,3\n-moreNorm 0\n" + lv
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts)
    hist, x = _pcmg_exact_coarse(orc, _dense, npts, levels, 3, 0.8, 100)
    assert it == len(hist) - 1 < 100
    # (the two evaluate b - A x in different summation orders: their norms differ by rounding of the size of ||b||, not of ||r||)
    assert np.max(np.abs(rdat - hist / hist[0])) <= 1e-13
    assert np.max(np.abs(u - x)) <= 1e-10 * np.abs(x).max()
    assert "type: lu" in out and "type: preonly" in out


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
@pytest.mark.parametrize("npts,scale,env", [(9, 0.8, None), (17, 1.0, None), (33, 0.8, None), (17, 0.8, {"MGPETSC_NO_RECOGNITION": "1"})])
def test_reference_driver_cycle1_single_grid(orc, tmp_path, npts, scale, env):
    """-cycle 1 (MultigridIcycle, src/solver.c:1991-2060) with one grid: a monitored Richardson +
    Jacobi iteration on the fine operator.  Bit-identical to the oracle's restatement (mgo_icycle)."""
    opts = (f"-npts {npts}\n-mesh 0\n-iter 20000\n-grids 1\n-levels 1\n-cycle 1\n-map 2\n-v 3,3\n-moreNorm 0\n"
            f"-pc_type jacobi\n-ksp_richardson_scale {scale!r}\n")
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts, env)
    ref = orc.icycle(2, npts, maxiter=20000, scale=scale)
    assert it == ref["iters"] < 20000
    assert np.max(np.abs(rdat - ref["rnorm"] / ref["rnorm"][0]) / rdat) <= 1e-12
    assert np.array_equal(u, ref["u"])
    assert "I-Cycle" in out


def _level_operator(orc, npts, grids):
    """dense restatement of levelMatrixA (src/solver.c:489-510) from the oracle's A, R, P: A_g on the diagonal, R^k A_g1 below,
    (A_g1 P^k) masked to P^k's window above; the right-hand side [f, R f, R R f, ...] (levelvecb, :558-620).  One value per window
    offset in the upper blocks, as the reference evaluates one expression per offset (:395-466), which a BLAS product does not promise."""
    A = [_dense(orc, "A", npts, g) for g in range(grids)]
    R = [_dense(orc, "R", npts, g) for g in range(grids - 1)]
    P = [_dense(orc, "P", npts, g) for g in range(grids - 1)]
    n = [(npts - 1) // 2 ** g - 1 for g in range(grids)]
    blocks = [[None] * grids for _ in range(grids)]
    for g0 in range(grids):
        for g1 in range(grids):
            if g0 == g1:
                blocks[g0][g1] = A[g0]
            elif g1 < g0:
                Rk = np.eye(n[g1] ** 2)
                for g in range(g1, g0):
                    Rk = R[g] @ Rk
                blocks[g0][g1] = Rk @ A[g1]
            else:                                   # row block g0 (finer), column block g1 (coarser)
                Pk = np.eye(n[g1] ** 2)
                for g in range(g1 - 1, g0 - 1, -1):
                    Pk = P[g] @ Pk
                W = (A[g0] @ Pk) * (Pk != 0)
                S, first = 2 ** (g1 - g0), {}
                for f, c in np.argwhere(W != 0):
                    d = (f // n[g0] - S * (c // n[g1]), f % n[g0] - S * (c % n[g1]))
                    W[f, c] = first.setdefault(d, W[f, c])
                blocks[g0][g1] = W
    f = orc.rhs(2, npts)
    b = [f]
    for g in range(grids - 1):
        b.append(R[g] @ b[-1])
    return np.block(blocks), np.concatenate(b), n


@pytest.mark.skipif(not os.path.exists(REFDRV), reason="build/refdriver/poisson absent")
@pytest.mark.parametrize("npts,grids,iters,env", [(9, 2, 25, None), (17, 2, 25, None), (65, 2, 25, None), (17, 2, 25, {"MGPETSC_NO_RECOGNITION": "1"}),
                                                  (65, 2, 25, {"MGPETSC_NO_RECOGNITION": "1"}), (17, 3, 25, None), (33, 3, 20, None), (33, 4, 20, None),
                                                  (65, 3, 20, None), (33, 4, 20, {"MGPETSC_NO_RECOGNITION": "1"})])
def test_reference_driver_cycle1_several_grids_in_one_level(orc, tmp_path, npts, grids, iters, env):
    """-cycle 1 with 2, 3 or 4 grids in ONE level: the reference assembles its coupled level operator (src/solver.c:255-487)
    with its own unmodified code; the drop-in recognises the block structure at MatAssemblyEnd (entry by entry) and applies it with
    the stencil and transfer kernels on composite vectors (mgk_apply, mgk_restrict_fw, mgk_apply_add, mgk_window_add); with recognition
    off it stays an assembled AIJ matrix on the generic CSR kernels.
    Restated in dense numpy from the oracle's A, R, P (_level_operator); x += s D^-1 (b - M x).  Point-Jacobi Richardson does not
    converge on this operator (the author's runs rely on PETSc PCs the drop-in does not provide), so a fixed number of iterations
    is compared: residual history to 1e-9, fine-grid part of x to 1e-9.  Parity with PETSc itself: unpinned."""
    s = 0.3
    opts = (f"-npts {npts}\n-mesh 0\n-iter {iters}\n-grids {grids}\n-levels 1\n-cycle 1\n-map 2\n-v 3,3\n-moreNorm 0\n"
            f"-pc_type jacobi\n-ksp_richardson_scale {s!r}\n")
    it, rdat, u, e, out = _run_reference_driver(tmp_path, opts, env)
    assert it == iters and rdat.size == iters + 1
    assert ("matrix-free level operator of several grids" in out) == (env is None)
    assert ("assembled AIJ (generic CSR kernel)" in out) == (env is not None)
    M, b, n = _level_operator(orc, npts, grids)
    nf = n[0] ** 2
    dinv = 1.0 / np.diag(M)
    x = np.zeros(M.shape[0])
    hist = [np.linalg.norm(b)]
    for _ in range(iters):
        x = x + s * (dinv * (b - M @ x))
        hist.append(np.linalg.norm(b - M @ x))
    hist = np.array(hist) / hist[0]
    # KSPSetResidualHistory(ksp, rnorm, numIter, ...) (src/solver.c:2017): a buffer of numIter entries holds the norms of
    # iterations 0..numIter-1 (PETSc stops logging when the buffer is full); entry numIter is never written when the
    # solve runs out of iterations
    assert np.max(np.abs(rdat[:-1] / hist[:-1] - 1)) <= 1e-9
    assert np.max(np.abs(u - x[:nf])) <= 1e-9 * np.abs(x[:nf]).max()


def _assemble(L, M):
    m = C.c_void_p()
    L.MatCreateAIJ(1, M.shape[0], M.shape[1], -1, -1, 30, None, 0, None, C.byref(m))
    for r, c in zip(*np.nonzero(M)):
        L.MatSetValue(m, int(r), int(c), float(M[r, c]), ADD)
    L.MatAssemblyBegin(m, FINAL)
    L.MatAssemblyEnd(m, FINAL)
    return m


def _set(L, v, a):
    for q, val in enumerate(a):
        L.VecSetValue(v, q, float(val), INSERT)
    L.VecAssemblyBegin(v)
    L.VecAssemblyEnd(v)


def _get(L, v, n):
    p = C.POINTER(C.c_double)()
    L.VecGetArray(v, C.byref(p))
    a = np.ctypeslib.as_array(p, shape=(n,)).copy()
    L.VecRestoreArray(v, C.byref(p))
    return a


@pytest.mark.parametrize("npts,grids", [(5, 2), (9, 2), (33, 2), (17, 3), (33, 4)])
def test_level_operator_of_several_grids_runs_on_the_stencil_and_transfer_kernels(orc, capfd, npts, grids):
    """The coupled operator of the I-cycle (src/solver.c:255-487), entered value by value through MatSetValue: recognised at
    MatAssemblyEnd, MatMult / MatMultAdd / MatResidual on composite vectors equal the dense product to rounding;
    a matrix that differs from the pattern in ONE entry stays on the generic CSR kernel."""
    L = _shim()
    L.PetscInitialize(None, None, None, None)
    L.MatMultAdd.argtypes = [C.c_void_p] * 4
    L.MatResidual.argtypes = [C.c_void_p] * 4
    M = _level_operator(orc, npts, grids)[0]
    N = M.shape[0]
    rng = np.random.default_rng(npts)
    for variant in ("exact", "perturbed W", "perturbed R A_h"):
        Mv = M.copy()
        nf = (npts - 2) ** 2
        if variant == "perturbed W":
            r, c = np.argwhere(Mv[:nf, nf:] != 0)[3]
            Mv[r, nf + c] *= 1.0 + 1e-6
        if variant == "perturbed R A_h":
            r, c = np.argwhere(Mv[nf:, :nf] != 0)[2]
            Mv[nf + r, c] *= 1.0 + 1e-6
        m = _assemble(L, Mv)
        L.MatView(m, None)
        out = capfd.readouterr().out
        # (one coarse point: every window offset occurs once, so ANY nine weights are a member of the family the kernel applies)
        expect = variant == "exact" or (variant == "perturbed W" and npts == 5)
        assert ("matrix-free level operator of several grids" in out) == expect, (variant, out)
        x, y, z, w = (C.c_void_p() for _ in range(4))
        L.MatCreateVecs(m, C.byref(x), C.byref(y))
        L.VecDuplicate(y, C.byref(z))
        L.VecDuplicate(y, C.byref(w))
        xv, bv = rng.standard_normal(N), rng.standard_normal(N)
        _set(L, x, xv)
        _set(L, z, bv)
        L.MatMult(m, x, y)
        ref = Mv @ xv
        tol = 1e-13 * np.abs(Mv).max() * np.abs(xv).max() * 30
        assert np.max(np.abs(_get(L, y, N) - ref)) <= tol
        L.MatResidual(m, z, x, w)
        assert np.max(np.abs(_get(L, w, N) - (bv - ref))) <= tol
        L.MatMultAdd(m, x, z, w)
        assert np.max(np.abs(_get(L, w, N) - (bv + ref))) <= tol
        L.MatMultAdd(m, x, z, z)
        assert np.max(np.abs(_get(L, z, N) - (bv + ref))) <= tol
        for v in (x, y, z, w):
            L.VecDestroy(C.byref(v))
        L.MatDestroy(C.byref(m))


def test_lazy_temporaries_keep_petsc_semantics(orc):
    """see tests/shim_semantics.py (shared with the CPU tier, which runs it over the host-memory mock of the kernel ABI)"""
    from shim_semantics import lazy_temporaries_keep_petsc_semantics, type_shim
    lazy_temporaries_keep_petsc_semantics(type_shim(_shim()), orc)


def test_speculative_sweep_of_the_norm_pass_is_adopted_only_when_nothing_changed(orc):
    from shim_semantics import speculative_sweep_is_adopted_only_when_nothing_changed, type_shim
    speculative_sweep_is_adopted_only_when_nothing_changed(type_shim(_shim()), orc)


def test_residual_left_deferred_by_the_norm_pass(orc):
    """max_it = 3: the norm pass makes all three sweeps of the next solve and does NOT store r; see tests/shim_semantics.py"""
    from shim_semantics import residual_left_deferred_by_the_norm_pass, type_shim
    residual_left_deferred_by_the_norm_pass(type_shim(_shim()), orc)


def test_recorded_coarse_subcycle_keeps_petsc_semantics(orc):
    """the levels from 63^2 down are recorded and run as ONE tail launch; see tests/shim_semantics.py"""
    from shim_semantics import recorded_coarse_subcycle_keeps_petsc_semantics, type_shim
    recorded_coarse_subcycle_keeps_petsc_semantics(type_shim(_shim()), orc)


def test_pcmg_level_vectors_after_the_tail_launch():
    """a process of its own (it sets -mg_levels_* in the options database): see tests/shim_semantics.py"""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "multigrid_petsc_amd", "libmgpetsc.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shim_semantics.py"), lib, "pcmgtail"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "SEMANTICS_OK pcmgtail" in p.stdout, (p.returncode, p.stdout[-3000:])


def test_richardson_with_lu_is_damped_not_exact():
    """a process of its own (it changes -pc_type in the options database): see tests/shim_semantics.py"""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "multigrid_petsc_amd", "libmgpetsc.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shim_semantics.py"), lib, "lu"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "SEMANTICS_OK lu" in p.stdout, (p.returncode, p.stdout[-3000:])


def test_random_programs_of_petsc_calls_keep_petsc_semantics():
    """programs of PETSc calls drawn at random over a two-level set-up against a call-by-call numpy model -- the fragments of the reference's loop
    started and interrupted wherever the draw says (a process of its own; see tests/shim_semantics.py: random_programs_keep_petsc_semantics)"""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "multigrid_petsc_amd", "libmgpetsc.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shim_semantics.py"), lib, "random", "1", "25"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "SEMANTICS_OK random" in p.stdout, (p.returncode, p.stdout[-3000:])
