"""CPU tests (run everywhere): the oracle against the pins we have.

The reference ships no tests or golden vectors and its floating-point work lives in PETSc, which is
not installed.  Pins available: (1) the reference-run outputs recorded in SURVEY.md
8(c2) for the integer half (tests/golden/survey_c2.json), (2) the closed-form discrete-eigenvector
known answer, (3) bit-equality of the two independent restatements inside the oracle (assembled AIJ
path following solver.c's MatSetValue loops vs matrix-free stencils)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import Oracle

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_c2.json")))


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def _maps(orc, npts, grids, levels, style, procs, l):
    tot = orc.L.mgo_level_total_2d(npts, grids, levels, l)
    glob = np.zeros(3 * tot, dtype=np.int32)
    grid = np.zeros(tot, dtype=np.int32)
    ranges = np.zeros(procs + 1, dtype=np.int32)
    rc = orc.L.mgo_mapping_2d(npts, grids, levels, style, procs, l, glob.ctypes.data, grid.ctypes.data, ranges.ctypes.data)
    assert rc == 0
    return glob.reshape(tot, 3), grid, ranges


def test_integer_half_against_survey_observations(orc):
    c = GOLD["case_npts17"]
    for l in range(2):
        n = orc.L.mgo_grid_n(c["npts"], l)
        assert n == c["level_n"][l]
        glob, grid, ranges = _maps(orc, c["npts"], c["grids"], c["levels"], c["map"], c["procs"], l)
        assert list(ranges) == c["ranges"][l]
        assert np.array_equal(grid, np.arange(n * n))                      # lexicographic
        assert np.array_equal(glob[:, 0], np.repeat(np.arange(n), n))      # i = row
        assert np.array_equal(glob[:, 1], np.tile(np.arange(n), n))        # j = column
        assert orc.level_stencil(2, c["npts"], l)[1] == c["h"][l]
    w = np.zeros(9)
    orc.L.mgo_restriction_stencil(w.ctypes.data)
    assert list(w) == c["res0"] and w.sum() == 1.0
    orc.L.mgo_prolongation_stencil(w.ctypes.data)
    assert list(w) == c["pro0"] and w.sum() == 4.0
    assert orc.L.mgo_mesh_h(2, 17) == c["mesh_h"]

    c = GOLD["case_npts129"]
    for l, size in enumerate(c["level_sizes"]):
        assert orc.L.mgo_level_total_2d(c["npts"], c["grids"], c["levels"], l) == size
    r = np.zeros(9, dtype=np.int32)
    orc.L.mgo_get_ranges(16129, 8, r.ctypes.data)
    assert list(r[:3]) == c["ranges_level0_head"] and r[-1] == c["ranges_level0_last"]
    orc.L.mgo_get_ranges(9, 8, r.ctypes.data)
    assert list(r) == c["ranges_level5"]


@pytest.mark.parametrize("npts,levels", [(9, 2), (17, 3), (33, 5)])
@pytest.mark.parametrize("procs", [1, 2, 4, 8])
def test_one_grid_per_level_all_styles_coincide(orc, npts, levels, procs):
    """F7: with -grids == -levels the three -map styles give the same lexicographic map and ranges."""
    for l in range(levels):
        ref = _maps(orc, npts, levels, levels, 0, procs, l)
        n = orc.L.mgo_grid_n(npts, l)
        assert np.array_equal(ref[1], np.arange(n * n))
        for style in (1, 2):
            got = _maps(orc, npts, levels, levels, style, procs, l)
            for a, b in zip(ref, got):
                assert np.array_equal(a, b)
        assert ref[2][-1] == n * n and np.all(np.diff(ref[2]) >= 0)
        sizes = np.diff(ref[2])
        assert sizes.max() - sizes.min() <= 1


@pytest.mark.parametrize("style", [0, 1, 2])
@pytest.mark.parametrize("procs", [1, 2, 3, 8])
def test_multi_grid_level_maps_are_consistent(orc, style, procs):
    """levels < grids: the last level holds several grids (src/matbuild.c:27-47).  Every style must give
    a bijection between (grid, i, j) and the global index, with ranges covering all unknowns."""
    npts, grids, levels = 17, 3, 2
    l = 1
    tot = orc.L.mgo_level_total_2d(npts, grids, levels, l)
    assert tot == 7 * 7 + 3 * 3
    glob, grid, ranges = _maps(orc, npts, grids, levels, style, procs, l)
    assert sorted(grid.tolist()) == list(range(tot))
    sizes = [7, 3]
    off = 0
    for lg, n in enumerate(sizes):
        for i in range(n):
            for j in range(n):
                idx = grid[off + i * n + j]
                assert tuple(glob[idx]) == (i, j, 1 + lg)
        off += n * n
    assert ranges[0] == 0 and ranges[-1] == tot and np.all(np.diff(ranges) >= 0)


def test_assembled_rows_follow_fillJacobians(orc):
    """5-point rows, ascending columns, Dirichlet by dropping neighbours (src/solver.c:239-251)."""
    A = orc.build("A", 2, 9, 0)
    rows = orc.csr_rows(A)
    n, c = 7, 64.0
    assert len(rows) == 49
    cols, vals = rows[0]
    assert list(cols) == [0, 1, 7] and list(vals) == [-4 * c, c, c]
    cols, vals = rows[3 * n + 3]
    assert list(cols) == [17, 23, 24, 25, 31] and list(vals) == [c, c, -4 * c, c, c]
    R = orc.csr_rows(orc.build("R", 2, 9, 0))
    assert len(R) == 9 and all(len(cv[0]) == 9 for cv in R)
    assert list(R[0][0]) == [0, 1, 2, 7, 8, 9, 14, 15, 16]
    P = orc.csr_rows(orc.build("P", 2, 9, 0))
    assert max(len(cv[0]) for cv in P) == 4
    assert list(P[8][0]) == [0] and list(P[8][1]) == [1.0]           # fine (1,1) coincides with coarse (0,0)
    assert list(P[0][1]) == [0.25]                                   # corner fine point: one parent, 1/4


@pytest.mark.parametrize("dim,npts,levels,scale", [(2, 17, 2, 1.0), (2, 17, 2, 0.8), (2, 33, 4, 0.8),
                                                   (2, 129, 7, 0.8), (3, 9, 2, 0.8), (3, 17, 4, 6.0 / 7.0)])
def test_two_restatements_agree_bitwise_and_hit_the_known_answer(orc, dim, npts, levels, scale):
    a = orc.vcycle(dim, npts, levels, 3, 3, maxiter=400, scale=scale, use_csr=1)
    b = orc.vcycle(dim, npts, levels, 3, 3, maxiter=400, scale=scale, use_csr=0)
    assert a["iters"] == b["iters"] < 400
    assert np.array_equal(a["rnorm"], b["rnorm"]) and np.array_equal(a["u"], b["u"])
    h = 1.0 / (npts - 1)
    kat = dim * math.pi ** 2 / ((4 * dim / h ** 2) * math.sin(math.pi * h / 2) ** 2) - 1.0
    err = orc.error_norms(dim, npts, a["u"])
    assert abs(err[0] - kat) <= 5e-7
    # stopping rule of src/solver.c:1530
    assert a["rnorm"][-1] <= 1e-7 * a["bnorm"] < a["rnorm"][-2]


def test_known_answer_and_survey_cycle_counts(orc):
    k = GOLD["kat_error_max"]
    for npts, key in ((17, "npts17"), (129, "npts129")):
        h = 1.0 / (npts - 1)
        kat = 2 * math.pi ** 2 / ((8 / h ** 2) * math.sin(math.pi * h / 2) ** 2) - 1.0
        assert abs(kat - k[key]) <= 1e-8 * k[key] + 1e-12
    g = GOLD["survey_indicative_cycles"]
    r = orc.vcycle(2, 129, 7, 3, 3, maxiter=100, scale=0.8)
    assert r["iters"] == g["npts129_levels7"]["cycles"]
    assert abs(r["rnorm"][1] / r["rnorm"][0] - g["npts129_levels7"]["first_reduction"]) < 1e-4
    r = orc.vcycle(2, 129, 6, 3, 3, maxiter=100, scale=0.8)
    assert r["iters"] == g["npts129_levels6"]["cycles"]
    assert abs(r["rnorm"][1] / r["rnorm"][0] - g["npts129_levels6"]["first_reduction"]) < 1e-4
    assert orc.vcycle(2, 17, 2, 3, 3, maxiter=200, scale=1.0)["iters"] == g["npts17_levels2_scale1"]["cycles"]
    assert orc.vcycle(2, 17, 2, 3, 3, maxiter=200, scale=0.8)["iters"] == g["npts17_levels2_scale08"]["cycles"]


def test_chebyshev_restatements_agree(orc):
    a = orc.vcycle(2, 33, 4, 3, 3, maxiter=100, ksp_type=1, emin=0.2, emax=2.0, use_csr=1)
    b = orc.vcycle(2, 33, 4, 3, 3, maxiter=100, ksp_type=1, emin=0.2, emax=2.0, use_csr=0)
    assert a["iters"] == b["iters"] < 100
    assert np.array_equal(a["rnorm"], b["rnorm"]) and np.array_equal(a["u"], b["u"])


def test_rhs_and_coords_follow_repeated_addition(orc):
    c = orc.coords(17)
    d = 1.0 / 16
    acc = [0.0]
    for _ in range(15):
        acc.append(acc[-1] + d)
    acc.append(1.0)
    assert list(c) == acc
    b = orc.rhs(2, 17).reshape(15, 15)
    PI = 3.14159265358979323846
    assert b[2, 5] == -2 * PI * PI * math.sin(PI * c[6]) * math.sin(PI * c[3])


def test_slab_operators_match_whole_grid(orc):
    """P-way z-slab evaluation with ghost planes == whole-grid evaluation (basis of the multi-GPU path)."""
    rng = np.random.default_rng(0)
    n = 15
    As, _ = orc.level_stencil(3, n + 2, 0)
    u, b = rng.uniform(-1, 1, n ** 3), rng.uniform(-1, 1, n ** 3)
    whole = orc.jacobi(3, n, As, 0.8, b, u)
    U, B = u.reshape(n, n, n), b.reshape(n, n, n)
    cuts = [0, 4, 8, 12, 15]
    parts = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        zlo = np.ascontiguousarray(U[a - 1]) if a > 0 else None
        zhi = np.ascontiguousarray(U[e]) if e < n else None
        parts.append(orc.jacobi(3, n, As, 0.8, np.ascontiguousarray(B[a:e]).ravel(),
                                np.ascontiguousarray(U[a:e]).ravel(), nz=e - a, zlo=zlo, zhi=zhi))
    assert np.array_equal(np.concatenate(parts), whole)


# ---- third restatement: committed scipy.sparse vectors (tests/golden/make_golden.py) ----
from golden_cases import GOLD as NPZ, CYCLE_KEYS, MESH_KEYS, cycle_case, mesh_case

GOLD_RTOL = 1e-12      # bar of BASELINE.json north_star; observed: bit-identical fields


@pytest.mark.parametrize("key", CYCLE_KEYS)
@pytest.mark.parametrize("use_csr", [0, 1])
def test_oracle_cycle_matches_committed_scipy_vectors(orc, key, use_csr):
    g = cycle_case(key)
    r = orc.vcycle(g["dim"], g["npts"], g["levels"], g["v0"], g["v1"], maxiter=g["maxiter"], scale=g["scale"], use_csr=use_csr)
    assert r["iters"] == g["iters"]
    assert abs(r["bnorm"] - g["bnorm"]) <= GOLD_RTOL * g["bnorm"]
    assert np.abs(r["rnorm"] - g["rnorm"]).max() <= GOLD_RTOL * g["rnorm"][0]
    assert np.abs(r["rnorm"] / g["rnorm"] - 1).max() <= 1e-9          # per entry, down to 1e-8 of rnorm[0]
    assert np.abs(r["u"] - g["u"]).max() <= GOLD_RTOL * np.abs(g["u"]).max()
    err = orc.error_norms(g["dim"], g["npts"], r["u"])
    assert np.abs(np.asarray(err) / g["err"] - 1).max() <= 1e-11


@pytest.mark.parametrize("dim,npts", [(2, 17), (2, 33), (2, 129), (3, 9), (3, 17), (3, 33)])
def test_oracle_rhs_matches_committed_vectors(orc, dim, npts):
    assert np.abs(orc.rhs(dim, npts) - NPZ["b0_d%d_n%d" % (dim, npts)]).max() <= 1e-13 * dim * math.pi ** 2


@pytest.mark.parametrize("dim,nf", [(2, 31), (3, 15)])
def test_oracle_operators_match_committed_vectors(orc, dim, nf):
    x, xc, y = NPZ["xfer_d%d_fine" % dim], NPZ["xfer_d%d_coarse" % dim], NPZ["xfer_d%d_base" % dim]
    As = orc.level_stencil(dim, nf + 2, 0)[0]
    assert np.array_equal(orc.restrict(dim, nf, x), NPZ["xfer_d%d_restricted" % dim])
    assert np.array_equal(orc.prolong_add(dim, nf, xc, y.copy()), NPZ["xfer_d%d_prolonged" % dim])
    assert np.array_equal(orc.apply(dim, nf, As, x), NPZ["xfer_d%d_applied" % dim])
    assert np.array_equal(orc.residual(dim, nf, As, y, x), NPZ["xfer_d%d_residual" % dim])


@pytest.mark.parametrize("npts,levels", [(9, 3), (17, 4), (33, 5)])
@pytest.mark.parametrize("procs", [1, 2, 4, 8])
def test_ranges_match_committed_vectors(orc, npts, levels, procs):
    want = NPZ["ranges_n%d_p%d" % (npts, procs)]
    for l in range(levels):
        for style in (0, 1, 2):
            _, _, ranges = _maps(orc, npts, levels, levels, style, procs, l)
            assert list(ranges) == list(want[l])


# ---- -cycle 8 (PCMG) restatement: SURVEY 8(f) N4 ----
@pytest.mark.parametrize("dim,npts,levels,scale", [(2, 33, 5, 0.8), (2, 129, 7, 0.8), (2, 17, 2, 1.0), (3, 17, 4, 6.0 / 7.0)])
def test_pcmg_restatement_is_the_vcycle_in_correction_form(orc, dim, npts, levels, scale):
    """Outer Richardson(1) + one PCMG V-cycle per application is algebraically the -cycle 0 iteration written in
    correction form (x += M r): same iterates up to rounding, hence same cycle count and residual history.
    This cross-pins the PCMG restatement (parity unpinned by the reference: PETSc-internal) against the V-cycle."""
    a = orc.vcycle(dim, npts, levels, 3, 3, maxiter=400, scale=scale)
    b = orc.pcmg(dim, npts, levels, 3, 3, maxiter=400, scale=scale)
    c = orc.pcmg(dim, npts, levels, 3, 3, maxiter=400, scale=scale, use_csr=1)
    assert b["iters"] == c["iters"] and np.array_equal(b["rnorm"], c["rnorm"]) and np.array_equal(b["u"], c["u"])
    assert a["iters"] == b["iters"]
    assert np.abs(b["rnorm"] / a["rnorm"] - 1).max() <= 1e-6      # rounding differs between the two forms
    assert np.abs(a["u"] - b["u"]).max() <= 1e-9 * np.abs(a["u"]).max()


def test_icycle_restatement(orc):
    """-cycle 1, one grid: both legs agree bit for bit; 9x9 grid converges to the discrete solution (KAT)."""
    a = orc.icycle(2, 9, maxiter=2000, scale=0.8)
    b = orc.icycle(2, 9, maxiter=2000, scale=0.8, use_csr=1)
    assert a["iters"] == b["iters"] < 2000 and np.array_equal(a["rnorm"], b["rnorm"]) and np.array_equal(a["u"], b["u"])
    h = 1.0 / 8
    kat = 2 * math.pi ** 2 / ((8 / h ** 2) * math.sin(math.pi * h / 2) ** 2) - 1.0
    assert abs(orc.error_norms(2, 9, a["u"])[0] - kat) <= 5e-7
    assert np.all(np.diff(a["rnorm"]) < 0)


@pytest.mark.parametrize("key", MESH_KEYS)
def test_oracle_stretched_mesh_cycle_matches_committed_scipy_vectors(orc, key):
    """-mesh 1/2: the oracle's assembled leg against the scipy.sparse restatement (coordinates and metrics through Python's
    math module: agreement to 1e-12, not bit for bit)"""
    g = mesh_case(key)
    r = orc.vcycle(2, g["npts"], g["levels"], g["v0"], g["v1"], maxiter=g["maxiter"], scale=g["scale"], use_csr=1, mesh=g["mesh"])
    assert r["iters"] == g["iters"]
    assert abs(r["bnorm"] - g["bnorm"]) <= GOLD_RTOL * g["bnorm"]
    assert np.abs(r["rnorm"] - g["rnorm"]).max() <= 1e-11 * g["rnorm"][0]
    assert np.abs(r["u"] - g["u"]).max() <= 1e-11 * np.abs(g["u"]).max()
    err = orc.error_norms_mesh(g["npts"], g["mesh"], r["u"])
    assert np.abs(np.asarray(err) / g["err"] - 1).max() <= 1e-9
