"""CPU tests of the product's C ABI: the libraries load without a GPU, export every symbol the headers
declare, refuse to run without a device (no CPU fallback), and the integer host logic is right."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[A-Za-z0-9_]+)\s*\(" % prefix, txt)))


def test_libmgk_exports_every_declared_symbol():
    from multigrid_petsc_amd._lib import load_mgk
    L = load_mgk()
    names = _declared("mgk.h", "mgk_")
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"libmgk.so does not export {n}"


def test_libmgpetsc_exports_every_declared_symbol():
    from multigrid_petsc_amd._lib import load_mgpetsc
    L = load_mgpetsc()
    names = _declared("mgsolve.h", "mg_") + _declared("mg_comm.h", "mg_comm_")
    assert len(names) >= 25
    for n in set(names):
        assert hasattr(L, n), f"libmgpetsc.so does not export {n}"


def test_geometry_layout():
    from multigrid_petsc_amd.mgk import Geom, _sigs
    from multigrid_petsc_amd._lib import load_mgk
    L = load_mgk()
    _sigs(L)
    g = Geom()
    assert L.mgk_geom_init(C.byref(g), 3, 1023, 1023, 1023) == 0
    assert g.pitch == 1040 and g.pitch % 16 == 0            # 16 + 1024 doubles: whole 128-B lines
    assert g.plane == 1040 * 1025 and g.org == g.plane + g.pitch + 16
    assert g.total == g.plane * 1025 + g.pitch
    assert L.mgk_geom_init(C.byref(g), 2, 4095, 4095, 1) == 0
    assert g.pitch == 4112 and g.nz == 1 and g.org == g.pitch + 16
    assert L.mgk_geom_init(C.byref(g), 3, 4, 4, 4) != 0     # even nx is rejected
    assert L.mgk_geom_init(C.byref(g), 4, 3, 3, 3) != 0


def test_no_cpu_fallback_without_a_device():
    from multigrid_petsc_amd.mgk import Mgk, MgkError
    from multigrid_petsc_amd._lib import load_mgk
    if load_mgk().mgk_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(MgkError, match="no HIP device"):
        Mgk(0)
    from multigrid_petsc_amd.solver import Solver, MgError
    with pytest.raises(MgError):
        Solver(2, 17, 2)


def test_ranges_and_maps_match_reference_semantics():
    from multigrid_petsc_amd import solver as S
    from oracle import Oracle
    orc = Oracle()
    L = S._lib()
    for tot, procs in ((225, 1), (16129, 8), (9, 8), (49, 3), (1070599167, 8)):
        want = np.zeros(procs + 1, dtype=np.int32)
        orc.L.mgo_get_ranges(tot, procs, want.ctypes.data)
        assert np.array_equal(S.get_ranges(tot, procs), want)
    for npts in (17, 129, 1025):
        for g in range(4):
            assert L.mg_grid_n(npts, g) == orc.L.mgo_grid_n(npts, g)
    k, i, j = C.c_int(), C.c_int(), C.c_int()
    for dim, n in ((2, 15), (3, 7)):
        for idx in (0, 1, n, n * n - 1, n ** dim - 1):
            L.mg_global_to_grid(dim, n, idx, C.byref(k), C.byref(i), C.byref(j))
            assert L.mg_grid_to_global(dim, n, k.value, i.value, j.value) == idx


@pytest.mark.parametrize("npts,ldist,nranks", [(1025, 4, 8), (1025, 4, 2), (1025, 3, 4), (513, 3, 8), (129, 1, 3), (65, 2, 5)])
def test_slab_split_is_nested_and_covers(npts, ldist, nranks):
    """multi-GPU decomposition: slabs tile every distributed level, starts are even, and the slab of level l
    is exactly twice the slab of level l+1 (+ the one extra plane on the last rank)."""
    from multigrid_petsc_amd.solver import slab_range, _lib
    L = _lib()
    for l in range(ldist + 1):
        n = L.mg_grid_n(npts, l)
        prev_end = 0
        for r in range(nranks):
            a, b = slab_range(npts, ldist, l, r, nranks)
            assert a == prev_end and b > a
            if l < ldist:
                assert a % 2 == 0
                ca, cb = slab_range(npts, ldist, l + 1, r, nranks)
                assert a == 2 * ca and b == (n if r == nranks - 1 else 2 * cb)
            prev_end = b
        assert prev_end == n


@pytest.mark.parametrize("npts,levels", [(9, 2), (17, 3), (33, 4)])
@pytest.mark.parametrize("style", [0, 1, 2])
def test_implicit_maps_equal_the_reference_maps_bit_exactly(npts, levels, style):
    """a9: the product keeps no index arrays; its formula maps must reproduce, entry for entry, the grid->global and
    global->(i,j,g) arrays the reference builds (src/matbuild.c:146-323, restated in the oracle) for every -map style
    when each level holds one grid, and its ranges for any number of ranks."""
    import ctypes as C
    from multigrid_petsc_amd import solver as S
    from oracle import Oracle
    orc = Oracle()
    L = S._lib()
    for l in range(levels):
        n = L.mg_grid_n(npts, l)
        tot = n * n
        for procs in (1, 3, 8):
            glob = np.zeros(3 * tot, dtype=np.int32)
            grid = np.zeros(tot, dtype=np.int32)
            ranges = np.zeros(procs + 1, dtype=np.int32)
            assert orc.L.mgo_mapping_2d(npts, levels, levels, style, procs, l, glob.ctypes.data, grid.ctypes.data, ranges.ctypes.data) == 0
            mine = np.array([L.mg_grid_to_global(2, n, 0, i, j) for i in range(n) for j in range(n)], dtype=np.int64)
            assert np.array_equal(mine, grid)
            k, i, j = C.c_int(), C.c_int(), C.c_int()
            for idx in range(tot):
                L.mg_global_to_grid(2, n, idx, C.byref(k), C.byref(i), C.byref(j))
                assert (i.value, j.value, l) == tuple(glob[3 * idx:3 * idx + 3])
            assert np.array_equal(S.get_ranges(tot, procs), ranges)


def _petsc_surface():
    """every function include/petscksp.h declares (prototypes only: `type name(args);` at the start of a line)"""
    txt = open(os.path.join(ROOT, "include", "petscksp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith("#"))
    names = re.findall(r"^\s*(?:PetscErrorCode|PetscViewer|int|double)\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(", txt, flags=re.M)
    return sorted(set(names))


def test_libmgpetsc_exports_the_whole_petsc_surface():
    """the drop-in must export every PETSc / MPI entry point its header promises the reference driver"""
    from multigrid_petsc_amd._lib import load_mgpetsc
    L = load_mgpetsc()
    names = _petsc_surface()
    assert len(names) >= 80 and "KSPSolve" in names and "MPI_Wtime" in names and "PCMGSetX" in names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"libmgpetsc.so does not export {missing}"


def test_reference_objects_need_nothing_the_drop_in_lacks(tmp_path):
    """In the build container (the reference is present): compile the reference's six UNMODIFIED sources against
    include/petscksp.h and require every undefined symbol of the objects to be exported by libmgpetsc.so / libmgk.so or to
    come from libc / libm -- the link of build/refdriver/poisson cannot fail on a missing PETSc entry point."""
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "src")):
        pytest.skip("the reference tree is not on this machine")
    from multigrid_petsc_amd._lib import load_mgpetsc, load_mgk
    Lp, Lk = load_mgpetsc(), load_mgk()
    libc = C.CDLL(None)
    libm = C.CDLL("libm.so.6")
    undefined = set()
    for f in ("array", "matbuild", "mesh", "problem", "solver", "poisson"):
        o = str(tmp_path / (f + ".o"))
        subprocess.run(["gcc", "-std=c99", "-O1", "-w", "-I" + os.path.join(ref, "include"), "-I" + os.path.join(ROOT, "include"),
                        "-c", os.path.join(ref, "src", f + ".c"), "-o", o], check=True)
        out = subprocess.run(["nm", "-u", o], check=True, stdout=subprocess.PIPE, text=True).stdout
        undefined |= {l.split()[-1] for l in out.splitlines() if l.strip()}
    defined = set()
    for f in ("array", "matbuild", "mesh", "problem", "solver", "poisson"):
        out = subprocess.run(["nm", "--defined-only", str(tmp_path / (f + ".o"))], check=True, stdout=subprocess.PIPE, text=True).stdout
        defined |= {l.split()[-1] for l in out.splitlines() if l.strip()}
    need = sorted(undefined - defined - {"_GLOBAL_OFFSET_TABLE_"})
    surface = set(_petsc_surface())
    missing = [n for n in need if not (hasattr(Lp, n) or hasattr(Lk, n) or hasattr(libc, n) or hasattr(libm, n))]
    assert not missing, f"undefined in the reference objects and provided by nobody: {missing}"
    petsc_like = [n for n in need if re.match(r"(Petsc|Vec|Mat|KSP|PC|IS|MPI_|PETSC_)", n)]
    assert len(petsc_like) >= 60
    not_declared = [n for n in petsc_like if n not in surface]
    assert not not_declared, f"used by the reference but not declared in include/petscksp.h: {not_declared}"


def test_own_driver_refuses_option_combinations_it_does_not_build(tmp_path):
    """mgpoisson checks its options before it touches the GPU: -grids != -levels (several grids per level), -map outside 0..2,
    another -pc_type or -cycle stop with exit code 2 and a message (the reference guards its own combinations the same way,
    src/poisson.c:61-71)"""
    import subprocess
    exe = os.path.join(ROOT, "multigrid_petsc_amd", "mgpoisson")
    if not os.path.exists(exe):
        pytest.skip("mgpoisson is not built")
    for args, msg in ((["-levels", "2", "-grids", "3"], "-grids 3 differs from -levels 2"), (["-map", "5"], "-map must be"),
                      (["-pc_type", "ilu"], "only -pc_type jacobi"), (["-cycle", "3"], "only -cycle 0")):
        p = subprocess.run([exe] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
        assert p.returncode == 2 and msg in p.stdout, (args, p.returncode, p.stdout)
