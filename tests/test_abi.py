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
