"""PETSc semantics of the drop-in's lazy temporaries in call orders the reference's loop does not use.  Shared by the GPU
tier (the real libmgpetsc.so) and the CPU tier (petsc_shim.c over tests/mock_mgk.cpp, built by tests/test_shim_semantics_cpu.py): the
functions take the ctypes library `L` with its argument types set (`type_shim`)."""
import ctypes as C

import numpy as np

ADD, INSERT, FINAL, NORM_2 = 2, 1, 0, 1


def type_shim(L):
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    L.PetscInitialize.argtypes = [vp, vp, C.c_char_p, C.c_char_p]
    L.PetscOptionsSetValue.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.MatCreateAIJ.argtypes = [i, i, i, i, i, i, vp, i, vp, C.POINTER(vp)]
    L.MatSetValue.argtypes = [vp, i, i, d, i]
    L.MatAssemblyBegin.argtypes = [vp, i]
    L.MatAssemblyEnd.argtypes = [vp, i]
    L.MatCreateVecs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.MatMult.argtypes = [vp, vp, vp]
    L.MatResidual.argtypes = [vp, vp, vp, vp]
    L.MatDestroy.argtypes = [C.POINTER(vp)]
    L.VecSetValue.argtypes = [vp, i, d, i]
    L.VecAssemblyBegin.argtypes = [vp]
    L.VecAssemblyEnd.argtypes = [vp]
    L.VecDuplicate.argtypes = [vp, C.POINTER(vp)]
    L.VecSet.argtypes = [vp, d]
    L.VecCopy.argtypes = [vp, vp]
    L.VecScale.argtypes = [vp, d]
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
    L.KSPRichardsonSetScale.argtypes = [vp, d]
    L.KSPGetPC.argtypes = [vp, C.POINTER(vp)]
    L.PCSetType.argtypes = [vp, C.c_char_p]
    L.PCMGSetLevels.argtypes = [vp, i, vp]
    L.PCMGGetCoarseSolve.argtypes = [vp, C.POINTER(vp)]
    L.PCMGGetSmoother.argtypes = [vp, i, C.POINTER(vp)]
    for f in (L.PCMGSetInterpolation, L.PCMGSetRestriction, L.PCMGSetR, L.PCMGSetRhs, L.PCMGSetX):
        f.argtypes = [vp, i, vp]
    return L


def _dense(orc, which, npts, l):
    m = orc.build(which, 2, npts, l)
    rows = orc.csr_rows(m)
    nr, nc = orc.L.mgo_csr_nrows(m), orc.L.mgo_csr_ncols(m)
    d = np.zeros((nr, nc))
    for r, (cols, vals) in enumerate(rows):
        d[r, list(cols)] = vals
    return d



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




def lazy_temporaries_keep_petsc_semantics(L, orc):
    """the deferred residual / prolongation / correction vectors of the drop-in in call orders the reference's loop does
    NOT use: a deferred vector read later, an operand changed or destroyed before the deferred vector is read, a correction that no sweep
    follows -- every value must be what call-by-call execution gives"""
    L.PetscInitialize(None, None, None, None)
    npts = 33
    A, P, R = _dense(orc, "A", npts, 0), _dense(orc, "P", npts, 0), _dense(orc, "R", npts, 0)
    mA, mP, mR = _assemble(L, A), _assemble(L, P), _assemble(L, R)
    nf, nc = A.shape[0], P.shape[1]
    rng = np.random.default_rng(3)
    xv, bv, ucv = rng.standard_normal(nf), rng.standard_normal(nf), rng.standard_normal(nc)
    x, b, r, rv, u = (C.c_void_p() for _ in range(5))
    uc, bc = C.c_void_p(), C.c_void_p()
    L.MatCreateVecs(mA, C.byref(x), C.byref(b))
    for v in (r, rv, u):
        L.VecDuplicate(x, C.byref(v))
    L.MatCreateVecs(mR, None, C.byref(bc))
    L.MatCreateVecs(mP, C.byref(uc), None)
    tol = 1e-12 * np.abs(A).max() * 10
    _set(L, x, xv); _set(L, b, bv); _set(L, uc, ucv)
    # 1. a deferred residual whose operand changes before it is read: the OLD x counts
    L.MatResidual(mA, b, x, r)
    L.VecScale(x, 2.0)
    assert np.max(np.abs(_get(L, r, nf) - (bv - A @ xv))) <= tol
    assert np.max(np.abs(_get(L, x, nf) - 2.0 * xv)) == 0.0
    # 2. deferred residual consumed by the restriction AND read afterwards
    _set(L, x, xv)
    L.MatResidual(mA, b, x, r)
    L.MatMult(mR, r, bc)
    assert np.max(np.abs(_get(L, bc, nc) - R @ (bv - A @ xv))) <= tol
    assert np.max(np.abs(_get(L, r, nf) - (bv - A @ xv))) <= tol
    # 2b. the restricted right-hand side is deferred as well: its fine-level operands change before it is read
    L.MatResidual(mA, b, x, r)
    L.MatMult(mR, r, bc)
    L.VecScale(x, 2.0)
    L.VecSet(b, 0.0)
    assert np.max(np.abs(_get(L, bc, nc) - R @ (bv - A @ xv))) <= tol
    assert np.max(np.abs(_get(L, r, nf) - (bv - A @ xv))) <= tol
    _set(L, x, xv); _set(L, b, bv)
    # 3. deferred prolongation + deferred correction, no sweep follows: u and rv read directly
    _set(L, u, xv)
    L.MatMult(mP, uc, rv)
    L.VecAXPY(u, 1.0, rv)
    assert np.max(np.abs(_get(L, u, nf) - (xv + P @ ucv))) <= 1e-13 * 10
    assert np.max(np.abs(_get(L, rv, nf) - P @ ucv)) <= 1e-13 * 10
    # 4. the coarse operand changes between the deferred prolongation, the deferred correction and their use
    _set(L, u, xv)
    L.MatMult(mP, uc, rv)
    L.VecAXPY(u, 1.0, rv)
    L.VecScale(uc, -3.0)                       # both deferred values were defined with the old u_c
    assert np.max(np.abs(_get(L, u, nf) - (xv + P @ ucv))) <= 1e-13 * 10
    assert np.max(np.abs(_get(L, rv, nf) - P @ ucv)) <= 1e-13 * 10
    # 5. ... or is destroyed
    _set(L, uc, ucv); _set(L, u, xv)
    L.MatMult(mP, uc, rv)
    L.VecAXPY(u, 1.0, rv)
    L.VecDestroy(C.byref(uc))
    assert np.max(np.abs(_get(L, u, nf) - (xv + P @ ucv))) <= 1e-13 * 10
    # 6. a deferred vector that is copied, and one that is overwritten unread
    L.MatResidual(mA, b, x, r)
    L.VecCopy(r, rv)
    assert np.max(np.abs(_get(L, rv, nf) - (bv - A @ xv))) <= tol
    L.MatResidual(mA, b, x, r)
    L.VecSet(r, 7.0)
    assert np.all(_get(L, r, nf) == 7.0)
    for v in (x, b, r, rv, u, bc):
        L.VecDestroy(C.byref(v))
    for m in (mA, mP, mR):
        L.MatDestroy(C.byref(m))


def speculative_sweep_is_adopted_only_when_nothing_changed(L, orc):
    """KSPBuildResidual + VecNorm on a Richardson smoother with a nonzero guess also make the first sweep of the next KSPSolve (DESIGN.md 8b
    N2).  That sweep must be dropped when u or b is written between the norm and the solve, and adopted otherwise -- either way the
    iterate is what sweep-by-sweep execution gives."""
    L.PetscInitialize(None, None, None, None)
    L.PetscOptionsSetValue(None, b"-pc_type", b"jacobi")
    L.PetscOptionsSetValue(None, b"-ksp_richardson_scale", b"0.8")
    npts = 33
    A = _dense(orc, "A", npts, 0)
    mA = _assemble(L, A)
    n = A.shape[0]
    d = 1.0 / np.diag(A)
    rng = np.random.default_rng(11)
    uv, bv = rng.standard_normal(n), rng.standard_normal(n)
    u, b, r = C.c_void_p(), C.c_void_p(), C.c_void_p()
    L.MatCreateVecs(mA, C.byref(u), C.byref(b))
    L.VecDuplicate(u, C.byref(r))
    k = C.c_void_p()
    L.KSPCreate(1, C.byref(k))
    L.KSPSetType(k, b"richardson"); L.KSPSetOperators(k, mA, mA); L.KSPSetNormType(k, 0)
    L.KSPSetTolerances(k, 1e-7, -2.0, -2.0, 2)
    L.KSPSetFromOptions(k)
    L.KSPSetInitialGuessNonzero(k, 1)
    V, val = C.c_void_p(), C.c_double()

    def sweeps(x, rhs, m, s=0.8):
        for _ in range(m):
            x = x + s * (d * (rhs - A @ x))
        return x

    tol = 1e-12 * max(np.abs(uv).max(), 1.0) * 100
    # "scale" / "options": a PARAMETER of the solver changes between the norm and the solve: the sweep the norm pass made
    # with the old scale must be dropped, both sweeps of the next solve use the new one
    for change in ("nothing", "u", "b", "scale", "options"):
        _set(L, u, uv); _set(L, b, bv)
        L.KSPSolve(k, b, u)
        x = sweeps(uv, bv, 2)
        L.KSPBuildResidual(k, None, r, C.byref(V))
        L.VecNorm(V, NORM_2, C.byref(val))
        assert abs(val.value - np.linalg.norm(bv - A @ x)) <= 1e-12 * np.linalg.norm(bv - A @ x)
        rhs, s2 = bv, 0.8
        if change == "u":
            L.VecScale(u, 0.5); x = 0.5 * x
        if change == "b":
            L.VecScale(b, 2.0); rhs = 2.0 * bv
        if change == "scale":
            L.KSPRichardsonSetScale(k, 0.5); s2 = 0.5
        if change == "options":
            L.PetscOptionsSetValue(None, b"-ksp_richardson_scale", b"0.6")
            L.KSPSetFromOptions(k); s2 = 0.6
        L.KSPSolve(k, b, u)
        x2 = sweeps(x, rhs, 2, s2)
        if change in ("scale", "options"):          # back to the scale of the other rounds
            L.PetscOptionsSetValue(None, b"-ksp_richardson_scale", b"0.8")
            L.KSPRichardsonSetScale(k, 0.8)
        assert np.max(np.abs(_get(L, u, n) - x2)) <= tol, change
        assert np.max(np.abs(_get(L, r, n) - (bv - A @ sweeps(uv, bv, 2)))) <= tol * np.abs(A).max(), change     # r was stored by the norm pass
    L.KSPDestroy(C.byref(k))
    for v in (u, b, r):
        L.VecDestroy(C.byref(v))
    L.MatDestroy(C.byref(mA))


def residual_left_deferred_by_the_norm_pass(L, orc):
    """max_it = 3 (the reference's -v 3,3): the pass that evaluates || b - A u || makes ALL three sweeps of the next KSPSolve and does not
    store r (round 3): r stays deferred, and when the solve adopts the sweeps (x and the work vector swap buffers) r's dependency follows
    the OLD iterate into the work vector.  Whatever is done next -- r read at once, after the adopting solve, after b was rewritten, after
    another solve reused the work vector, after the solver died, the norm asked twice -- r must be b - A (old u) and u three sweeps further."""
    L.PetscInitialize(None, None, None, None)
    L.PetscOptionsSetValue(None, b"-pc_type", b"jacobi")
    L.PetscOptionsSetValue(None, b"-ksp_richardson_scale", b"0.8")
    npts = 33
    A = _dense(orc, "A", npts, 0)
    mA = _assemble(L, A)
    n = A.shape[0]
    d = 1.0 / np.diag(A)
    rng = np.random.default_rng(17)
    uv, bv = rng.standard_normal(n), rng.standard_normal(n)

    def sweeps(x, rhs, m, s=0.8):
        for _ in range(m):
            x = x + s * (d * (rhs - A @ x))
        return x

    tol = 1e-12 * max(np.abs(uv).max(), 1.0) * 100
    rtol = tol * np.abs(A).max()
    for case in ("reference order", "read r at once", "read r after the solve", "b rewritten after the solve", "second solve reuses the work vector",
                 "solver destroyed", "norm twice", "norm again after the solve", "u written before the solve", "restriction of r after the solve"):
        u, b, r = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.MatCreateVecs(mA, C.byref(u), C.byref(b))
        L.VecDuplicate(u, C.byref(r))
        k = C.c_void_p()
        L.KSPCreate(1, C.byref(k))
        L.KSPSetType(k, b"richardson"); L.KSPSetOperators(k, mA, mA); L.KSPSetNormType(k, 0)
        L.KSPSetTolerances(k, 1e-7, -2.0, -2.0, 3)
        L.KSPSetFromOptions(k)
        L.KSPSetInitialGuessNonzero(k, 1)
        V, val = C.c_void_p(), C.c_double()
        _set(L, u, uv); _set(L, b, bv)
        L.KSPSolve(k, b, u)
        x = sweeps(uv, bv, 3)
        r_old = bv - A @ x
        L.KSPBuildResidual(k, None, r, C.byref(V))
        L.VecNorm(V, NORM_2, C.byref(val))
        assert abs(val.value - np.linalg.norm(r_old)) <= 1e-12 * np.linalg.norm(r_old), case
        if case == "read r at once":
            assert np.max(np.abs(_get(L, r, n) - r_old)) <= rtol, case
        if case == "norm twice":
            L.VecNorm(V, NORM_2, C.byref(val))
            assert abs(val.value - np.linalg.norm(r_old)) <= 1e-12 * np.linalg.norm(r_old), case
        if case == "u written before the solve":
            L.VecScale(u, 0.5)
            x = 0.5 * x
        L.KSPSolve(k, b, u)                                  # adopts the three sweeps (except after the write to u)
        x3 = sweeps(x, bv, 3)
        if case == "reference order":
            L.KSPBuildResidual(k, None, r, C.byref(V))       # r overwritten unread (src/solver.c:1534)
            L.VecNorm(V, NORM_2, C.byref(val))
            assert abs(val.value - np.linalg.norm(bv - A @ x3)) <= 1e-11 * np.linalg.norm(bv - A @ x3), case
        elif case == "b rewritten after the solve":
            L.VecSet(b, 1.0)                                 # r was defined with the old b
            assert np.max(np.abs(_get(L, r, n) - r_old)) <= rtol, case
            assert np.all(_get(L, b, n) == 1.0)
        elif case == "second solve reuses the work vector":
            L.KSPSolve(k, b, u)                              # writes the work vector that holds the old iterate
            assert np.max(np.abs(_get(L, r, n) - r_old)) <= rtol, case
            x3 = sweeps(x3, bv, 3)
        elif case == "solver destroyed":
            L.KSPDestroy(C.byref(k))
            assert np.max(np.abs(_get(L, r, n) - r_old)) <= rtol, case
        elif case == "norm again after the solve":
            L.VecNorm(V, NORM_2, C.byref(val))
            assert abs(val.value - np.linalg.norm(r_old)) <= 1e-12 * np.linalg.norm(r_old), case
            assert np.max(np.abs(_get(L, r, n) - r_old)) <= rtol, case
        elif case == "restriction of r after the solve":
            Rm = _dense(orc, "R", npts, 0)
            mR = _assemble(L, Rm)
            bc = C.c_void_p()
            L.MatCreateVecs(mR, None, C.byref(bc))
            L.MatMult(mR, r, bc)
            assert np.max(np.abs(_get(L, bc, Rm.shape[0]) - Rm @ r_old)) <= rtol, case
            L.VecDestroy(C.byref(bc)); L.MatDestroy(C.byref(mR))
        elif case != "u written before the solve":
            assert np.max(np.abs(_get(L, r, n) - r_old)) <= rtol, case
        if case == "u written before the solve":
            assert np.max(np.abs(_get(L, r, n) - r_old)) <= rtol, case      # r belongs to u BEFORE the scaling
        assert np.max(np.abs(_get(L, u, n) - x3)) <= tol, case
        if k:
            L.KSPDestroy(C.byref(k))
        for v in (u, b, r):
            L.VecDestroy(C.byref(v))
    L.MatDestroy(C.byref(mA))


def recorded_coarse_subcycle_keeps_petsc_semantics(L, orc):
    """the drop-in RECORDS the reference's calls on the levels from 63^2 down and runs them as ONE tail launch when the pattern completes
    (DESIGN.md 8b N2, round 3).  Here the pattern is completed, interrupted, broken and abused in ways the reference never does; after every
    program each vector must hold what call-by-call execution gives (a numpy model runs the same program)."""
    L.PetscInitialize(None, None, None, None)
    L.PetscOptionsSetValue(None, b"-pc_type", b"jacobi")
    L.PetscOptionsSetValue(None, b"-ksp_richardson_scale", b"0.8")
    L.MatScale.argtypes = [C.c_void_p, C.c_double]
    npts, nlev = 33, 3
    A = [_dense(orc, "A", npts, l) for l in range(nlev)]
    R = [_dense(orc, "R", npts, l) for l in range(nlev - 1)]
    P = [_dense(orc, "P", npts, l) for l in range(nlev - 1)]
    rng = np.random.default_rng(23)
    b0v = rng.standard_normal(A[0].shape[0])

    programs = {
        "complete, then every intermediate read": ["cycle", "read_all"],
        "complete, b0 overwritten, then the intermediates read": ["cycle", ("set", "b0", 1.5), "read_all"],
        "complete, b0 scaled, then the intermediates read": ["cycle", ("scale", "b0", 2.0), "read_all"],
        "two cycles, the second overwrites the first's intermediates unread": ["cycle", ("scale", "b0", 0.5), "cycle", "read_all"],
        "b1 read in the middle of the descent": [("cycle", {"after_restrict0": [("read", "b1")]}), "read_all"],
        "u0 scaled in the middle of the descent": [("cycle", {"after_restrict0": [("scale", "u0", 2.0)]}), "read_all"],
        "u1 overwritten on the way up": [("cycle", {"after_prolong1": [("set", "u1", 0.25)]}), "read_all"],
        "another scale on level 1": [("cycle", {"before_solve1": [("kscale", 1, 0.5)]}), "read_all"],
        "two sweeps on level 0, three below": [("cycle", {"start": [("kits", 0, 2)]}), "read_all"],
        "A1 scaled in the middle of the ascent": [("cycle", {"after_prolong1": [("matscale", 1, 1.25)]}), "read_all", ("matscale", 1, 0.8)],
        "solver of level 1 destroyed after the cycle": ["cycle", ("kdestroy", 1), "read_all"],
        "descent abandoned after the first restriction": [("cycle", {"after_restrict0": ["stop"]}), "read_all"],
    }
    for name, prog in programs.items():
        mA, mR, mP = [_assemble(L, a) for a in A], [_assemble(L, r) for r in R], [_assemble(L, q) for q in P]
        V, M = {}, {}                                        # PETSc vectors / the numpy model's values
        for l in range(nlev):
            for nm in ("u", "b", "rv"):
                v = C.c_void_p()
                L.MatCreateVecs(mA[l], C.byref(v), None)
                V[nm + str(l)] = v
                M[nm + str(l)] = np.zeros(A[l].shape[0])
        _set(L, V["b0"], b0v); M["b0"] = b0v.copy()
        K, par = [], []
        for l in range(nlev):
            k = C.c_void_p()
            L.KSPCreate(1, C.byref(k))
            L.KSPSetType(k, b"richardson"); L.KSPSetOperators(k, mA[l], mA[l]); L.KSPSetNormType(k, 0)
            L.KSPSetTolerances(k, 1e-7, -2.0, -2.0, 3)
            L.KSPSetFromOptions(k)
            K.append(k); par.append({"its": 3, "scale": 0.8, "guess": 0, "A": A[l].copy(), "alive": True})

        def solve(l):
            x = M["u%d" % l] if par[l]["guess"] else np.zeros_like(M["u%d" % l])
            d = 1.0 / np.diag(par[l]["A"])
            for _ in range(par[l]["its"]):
                x = x + par[l]["scale"] * (d * (M["b%d" % l] - par[l]["A"] @ x))
            M["u%d" % l] = x
            L.KSPSolve(K[l], V["b%d" % l], V["u%d" % l])

        def extra(steps):
            for st in steps or []:
                if st == "stop":
                    return True
                if st[0] == "read":
                    assert np.max(np.abs(_get(L, V[st[1]], M[st[1]].size) - M[st[1]])) <= tol(st[1]), (name, st)
                elif st[0] == "scale":
                    L.VecScale(V[st[1]], st[2]); M[st[1]] = st[2] * M[st[1]]
                elif st[0] == "set":
                    L.VecSet(V[st[1]], st[2]); M[st[1]] = np.full_like(M[st[1]], st[2])
                elif st[0] == "kscale":
                    L.KSPRichardsonSetScale(K[st[1]], st[2]); par[st[1]]["scale"] = st[2]
                elif st[0] == "kits":
                    L.KSPSetTolerances(K[st[1]], 1e-7, -2.0, -2.0, st[2]); par[st[1]]["its"] = st[2]
                elif st[0] == "matscale":
                    L.MatScale(mA[st[1]], st[2]); par[st[1]]["A"] = st[2] * par[st[1]]["A"]
                elif st[0] == "kdestroy":
                    L.KSPDestroy(C.byref(K[st[1]])); par[st[1]]["alive"] = False
            return False

        def tol(nm):
            return 1e-11 * max(1.0, np.abs(M[nm]).max())

        def cycle(hooks):
            # the reference's loop on these levels (src/solver.c:1531-1544), every solver back to the zero guess first
            for l in range(nlev):
                L.KSPSetInitialGuessNonzero(K[l], 0); par[l]["guess"] = 0
            if extra(hooks.get("start")):
                return
            solve(0)
            L.KSPSetInitialGuessNonzero(K[0], 1); par[0]["guess"] = 1
            for l in range(1, nlev):
                Vp = C.c_void_p()
                L.KSPBuildResidual(K[l - 1], None, V["rv%d" % (l - 1)], C.byref(Vp))
                M["rv%d" % (l - 1)] = M["b%d" % (l - 1)] - par[l - 1]["A"] @ M["u%d" % (l - 1)]
                L.MatMult(mR[l - 1], Vp, V["b%d" % l])
                M["b%d" % l] = R[l - 1] @ M["rv%d" % (l - 1)]
                if extra(hooks.get("after_restrict%d" % (l - 1))):
                    return
                if extra(hooks.get("before_solve%d" % l)):
                    return
                solve(l)
                if l != nlev - 1:
                    L.KSPSetInitialGuessNonzero(K[l], 1); par[l]["guess"] = 1
            for l in range(nlev - 2, -1, -1):
                L.MatMult(mP[l], V["u%d" % (l + 1)], V["rv%d" % l])
                M["rv%d" % l] = P[l] @ M["u%d" % (l + 1)]
                if extra(hooks.get("after_prolong%d" % l)):
                    return
                L.VecAXPY(V["u%d" % l], 1.0, V["rv%d" % l])
                M["u%d" % l] = M["u%d" % l] + M["rv%d" % l]
                solve(l)
                if l != 0:
                    L.KSPSetInitialGuessNonzero(K[l], 0); par[l]["guess"] = 0

        for st in prog:
            if st == "cycle":
                cycle({})
            elif st == "read_all":
                for nm in sorted(M):
                    assert np.max(np.abs(_get(L, V[nm], M[nm].size) - M[nm])) <= tol(nm), (name, nm)
            elif st[0] == "cycle":
                cycle(st[1])
            else:
                extra([st])
        for l in range(nlev):
            if par[l]["alive"]:
                L.KSPDestroy(C.byref(K[l]))
        for v in V.values():
            L.VecDestroy(C.byref(v))
        for m in mA + mR + mP:
            L.MatDestroy(C.byref(m))


def pcmg_level_vectors_after_the_tail_launch(L, orc):
    """-cycle 8 set up as the reference does (src/solver.c:1918-1956: its own vectors handed to PCMG through PCMGSetRhs / PCMGSetX / PCMGSetR), full
    depth, PETSc's default exact coarse solve: the drop-in runs PCMG's levels from 63^2 down -- here all of them -- as ONE tail launch and does
    not compute the level vectors.  Reading them after the solve must give what the level-by-level cycle leaves there (numpy model)."""
    L.PetscInitialize(None, None, None, None)
    for k_, v_ in ((b"-mg_levels_ksp_type", b"richardson"), (b"-mg_levels_pc_type", b"jacobi"), (b"-mg_levels_ksp_max_it", b"3"),
                   (b"-mg_levels_ksp_richardson_scale", b"0.8")):
        L.PetscOptionsSetValue(None, k_, v_)
    npts, nlev = 33, 5
    A = [_dense(orc, "A", npts, l) for l in range(nlev)]
    R = [_dense(orc, "R", npts, l) for l in range(nlev - 1)]
    P = [_dense(orc, "P", npts, l) for l in range(nlev - 1)]
    mA, mR, mP = [_assemble(L, a) for a in A], [_assemble(L, r) for r in R], [_assemble(L, q) for q in P]
    rng = np.random.default_rng(31)
    b0v = rng.standard_normal(A[0].shape[0])
    u, b, r = [], [], []
    for l in range(nlev):
        for lst in (u, b, r):
            v = C.c_void_p()
            L.MatCreateVecs(mA[l], C.byref(v), None)
            lst.append(v)
    _set(L, b[0], b0v)
    ksp, pc, kt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    L.KSPCreate(1, C.byref(ksp))
    L.KSPSetType(ksp, b"richardson"); L.KSPSetOperators(ksp, mA[0], mA[0]); L.KSPSetNormType(ksp, 2)
    L.KSPSetTolerances(ksp, 1e-30, -2.0, -2.0, 2)                       # two applications of the preconditioner
    L.KSPGetPC(ksp, C.byref(pc))
    L.PCSetType(pc, b"mg")
    L.PCMGSetLevels(pc, nlev, None)
    L.PCMGGetCoarseSolve(pc, C.byref(kt)); L.KSPSetOperators(kt, mA[nlev - 1], mA[nlev - 1])
    for i in range(1, nlev):
        L.PCMGGetSmoother(pc, i, C.byref(kt)); L.KSPSetOperators(kt, mA[nlev - i - 1], mA[nlev - i - 1])
        L.PCMGSetInterpolation(pc, i, mP[nlev - i - 1]); L.PCMGSetRestriction(pc, i, mR[nlev - i - 1])
    L.PCMGSetR(pc, nlev - 1, r[0])
    for i in range(1, nlev - 1):
        L.PCMGSetRhs(pc, i, b[nlev - i - 1]); L.PCMGSetX(pc, i, u[nlev - i - 1]); L.PCMGSetR(pc, i, r[nlev - i - 1])
    L.PCMGSetRhs(pc, 0, b[nlev - 1]); L.PCMGSetX(pc, 0, u[nlev - 1])
    L.KSPSetFromOptions(ksp)
    L.KSPSolve(ksp, b[0], u[0])
    # the numpy model: outer Richardson (scale 1), PCMG V-cycle with 3 + 3 damped Jacobi sweeps, exact solve on the 1 x 1 grid
    lev = {}

    def sm(l, x, rhs, m):
        d = 1.0 / np.diag(A[l])
        for _ in range(m):
            x = x + 0.8 * (d * (rhs - A[l] @ x))
        return x

    def cyc(l, rhs):
        if l == nlev - 1:
            x = np.linalg.solve(A[l], rhs)
        else:
            x = sm(l, np.zeros_like(rhs), rhs, 3)
            bc = R[l] @ (rhs - A[l] @ x)
            x = x + P[l] @ cyc(l + 1, bc)
            x = sm(l, x, rhs, 3)
        if l > 0:
            lev[l] = (rhs.copy(), x.copy())
        return x

    x = np.zeros_like(b0v)
    for _ in range(2):
        x = x + cyc(0, b0v - A[0] @ x)
    tol = 1e-11 * max(1.0, np.abs(x).max())
    assert np.max(np.abs(_get(L, u[0], x.size) - x)) <= tol
    for l in range(nlev - 1, 0, -1):                                     # the level vectors, coarsest first, then once more in the other order
        assert np.max(np.abs(_get(L, u[l], lev[l][1].size) - lev[l][1])) <= 1e-11 * max(1.0, np.abs(lev[l][1]).max()), ("u", l)
        assert np.max(np.abs(_get(L, b[l], lev[l][0].size) - lev[l][0])) <= 1e-11 * max(1.0, np.abs(lev[l][0]).max()), ("b", l)
    for l in range(1, nlev):
        assert np.max(np.abs(_get(L, u[l], lev[l][1].size) - lev[l][1])) <= 1e-11 * max(1.0, np.abs(lev[l][1]).max()), ("u again", l)
    L.KSPDestroy(C.byref(ksp))
    for v in u + b + r:
        L.VecDestroy(C.byref(v))
    for m in mA + mR + mP:
        L.MatDestroy(C.byref(m))


def richardson_with_lu_is_damped_not_exact(L, orc):
    """-pc_type lu (the dense inverse of a small operator): preonly and richardson with scale 1 return A^-1 b whatever the guess;
    richardson with scale s != 1 makes max_it DAMPED steps x <- (1 - s) x + s A^-1 b, as PETSc would (it used to come
    back as the exact solution, silently)."""
    L.PetscInitialize(None, None, None, None)
    L.PetscOptionsSetValue(None, b"-pc_type", b"lu")
    A = _dense(orc, "A", 17, 0)
    mA = _assemble(L, A)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    bv, x0 = rng.standard_normal(n), rng.standard_normal(n)
    y = np.linalg.solve(A, bv)
    u, b = C.c_void_p(), C.c_void_p()
    L.MatCreateVecs(mA, C.byref(u), C.byref(b))
    _set(L, b, bv)
    tol = 1e-10 * np.abs(y).max()
    for ktype, scale, maxit, guess in ((b"preonly", 1.0, 1, 0), (b"richardson", 1.0, 2, 1), (b"richardson", 1.0, 1, 0),
                                       (b"richardson", 0.5, 3, 0), (b"richardson", 0.5, 3, 1), (b"richardson", 1.3, 2, 1), (b"richardson", 0.7, 0, 1)):
        k = C.c_void_p()
        L.KSPCreate(1, C.byref(k))
        L.KSPSetType(k, ktype); L.KSPSetOperators(k, mA, mA); L.KSPSetNormType(k, 0)
        L.KSPSetTolerances(k, 1e-7, -2.0, -2.0, maxit)
        L.KSPSetFromOptions(k)
        L.KSPRichardsonSetScale(k, scale)
        L.KSPSetInitialGuessNonzero(k, guess)
        _set(L, u, x0)
        L.KSPSolve(k, b, u)
        x = x0.copy() if guess else np.zeros(n)
        if ktype == b"preonly":
            x = y
        else:
            for _ in range(maxit):
                x = (1.0 - scale) * x + scale * y
        got = _get(L, u, n)
        assert np.max(np.abs(got - x)) <= tol, (ktype, scale, maxit, guess, np.max(np.abs(got - x)))
        L.KSPDestroy(C.byref(k))
    for v in (u, b):
        L.VecDestroy(C.byref(v))
    L.MatDestroy(C.byref(mA))


def random_programs_keep_petsc_semantics(L, orc, seed=1, nprog=40, nops=60):
    """Programs of PETSc calls drawn at random over a two-level set-up (fine / coarse 5-point operators, full weighting, bilinear prolongation,
    two Richardson + Jacobi solvers, six fine and three coarse vectors) against a numpy model that executes call by call: residuals, transfers,
    corrections, BLAS-1, norms, solves with changing max_it / scale / guess flag, reads, destroy + re-create -- with the fragments of the
    reference's loop (src/solver.c:1531-1546) among the draws, so that the drop-in's deferred temporaries, speculative sweeps and fused passes
    start and are then interrupted wherever the draw says.  Every value read, every norm and every vector at the end of a program must be what
    call-by-call execution gives (tolerance: 1e-11 of a running magnitude bound of the vector; the arithmetic order differs from numpy's)."""
    L.PetscInitialize(None, None, None, None)
    L.PetscOptionsSetValue(None, b"-pc_type", b"jacobi")
    rng = np.random.default_rng(seed)
    for prog in range(nprog):
        npts = int(rng.choice([17, 33, 33, 65]))
        A0, A1, P, R = _dense(orc, "A", npts, 0), _dense(orc, "A", npts, 1), _dense(orc, "P", npts, 0), _dense(orc, "R", npts, 0)
        mA0, mA1, mP, mR = _assemble(L, A0), _assemble(L, A1), _assemble(L, P), _assemble(L, R)
        nf, nc = A0.shape[0], A1.shape[0]
        amax = {0: 5.0 * np.abs(A0).max(), 1: 5.0 * np.abs(A1).max()}
        dinv = {0: 1.0 / np.diag(A0), 1: 1.0 / np.diag(A1)}
        Aop = {0: A0, 1: A1}
        fine, coarse = ["x", "b", "r", "rv", "u", "w"], ["uc", "bc", "rc"]
        lev = {**{v: 0 for v in fine}, **{v: 1 for v in coarse}}
        H, M, mag = {}, {}, {}
        first_f, first_c = C.c_void_p(), C.c_void_p()
        L.MatCreateVecs(mA0, C.byref(first_f), None)
        L.MatCreateVecs(mA1, C.byref(first_c), None)
        for v in fine + coarse:
            h = C.c_void_p()
            L.VecDuplicate(first_f if lev[v] == 0 else first_c, C.byref(h))
            H[v] = h
            M[v] = rng.standard_normal(nf if lev[v] == 0 else nc)
            mag[v] = float(np.abs(M[v]).max())
            _set(L, h, M[v])
        K, kst = {}, {}
        for q, mat in ((0, mA0), (1, mA1)):
            k = C.c_void_p()
            L.KSPCreate(1, C.byref(k))
            L.KSPSetType(k, b"richardson"); L.KSPSetOperators(k, mat, mat); L.KSPSetNormType(k, 0)
            kst[q] = {"maxit": int(rng.integers(0, 5)), "scale": float(rng.choice([0.5, 0.8, 1.0])), "guess": int(rng.integers(0, 2)), "b": None, "x": None}
            L.KSPSetTolerances(k, 1e-7, -2.0, -2.0, kst[q]["maxit"])
            L.KSPSetFromOptions(k)
            L.KSPRichardsonSetScale(k, kst[q]["scale"])
            L.KSPSetInitialGuessNonzero(k, kst[q]["guess"])
            K[q] = k
        trace = []

        def pick(level, n=1, exclude=()):
            pool = [v for v in (fine if level == 0 else coarse) if v not in exclude]
            return [str(t) for t in rng.choice(pool, size=n, replace=False)]

        def check(v, what):
            n = nf if lev[v] == 0 else nc
            got = _get(L, H[v], n)
            err = float(np.max(np.abs(got - M[v])))
            assert err <= 1e-11 * max(mag[v], 1e-30) + 1e-300, (f"seed {seed} program {prog}: {what} of {v}: error {err:.3e} against magnitude {mag[v]:.3e}", trace)

        def residual(q, bn, xn, rn_, via_ksp):
            if via_ksp:
                V = C.c_void_p()
                L.KSPBuildResidual(K[q], None, H[rn_], C.byref(V))
            else:
                L.MatResidual(mA0 if q == 0 else mA1, H[bn], H[xn], H[rn_])
            M[rn_] = M[bn] - Aop[q] @ M[xn]
            mag[rn_] = mag[bn] + amax[q] * mag[xn]

        def solve(q, bn, xn):
            st = kst[q]
            L.KSPSolve(K[q], H[bn], H[xn])
            xm = M[xn].copy() if st["guess"] else np.zeros_like(M[xn])
            mg = mag[xn] if st["guess"] else 0.0
            for _ in range(st["maxit"]):
                xm = xm + st["scale"] * (dinv[q] * (M[bn] - Aop[q] @ xm))
                mg = mg + st["scale"] * float(dinv[q].max()) * (mag[bn] + amax[q] * mg)
            M[xn], mag[xn] = xm, max(mg, float(np.abs(xm).max()))
            st["b"], st["x"] = bn, xn

        for step in range(nops):
            op = str(rng.choice(["resid", "kresid", "restrict", "prolong", "apply", "axpy", "scale", "set", "copy", "norm", "solve", "solve", "param", "read",
                                 "recreate", "frag_down", "frag_up", "frag_up", "frag_norm", "frag_norm", "frag_cycle"]))
            q = int(rng.integers(0, 2))
            if op == "resid":
                bn, xn, rn_ = pick(q, 3)
                trace.append(f"MatResidual(A{q}, {bn}, {xn}, {rn_})")
                residual(q, bn, xn, rn_, False)
            elif op == "kresid":
                if kst[q]["b"] is None:
                    continue
                (rn_,) = pick(q, 1, exclude=(kst[q]["b"], kst[q]["x"]))
                trace.append(f"KSPBuildResidual(k{q}, {rn_})   [b={kst[q]['b']}, x={kst[q]['x']}]")
                residual(q, kst[q]["b"], kst[q]["x"], rn_, True)
            elif op == "restrict":
                (f,), (c,) = pick(0), pick(1)
                trace.append(f"MatMult(R, {f}, {c})")
                L.MatMult(mR, H[f], H[c]); M[c] = R @ M[f]; mag[c] = mag[f]
            elif op == "prolong":
                (f,), (c,) = pick(0), pick(1)
                trace.append(f"MatMult(P, {c}, {f})")
                L.MatMult(mP, H[c], H[f]); M[f] = P @ M[c]; mag[f] = mag[c]
            elif op == "apply":
                a, bb = pick(q, 2)
                trace.append(f"MatMult(A{q}, {a}, {bb})")
                L.MatMult(mA0 if q == 0 else mA1, H[a], H[bb]); M[bb] = Aop[q] @ M[a]; mag[bb] = amax[q] * mag[a]
            elif op == "axpy":
                y, xx = pick(q, 2)
                al = float(rng.choice([1.0, -1.0, 0.5, 2.0]))
                trace.append(f"VecAXPY({y}, {al}, {xx})")
                L.VecAXPY(H[y], al, H[xx]); M[y] = M[y] + al * M[xx]; mag[y] = mag[y] + abs(al) * mag[xx]
            elif op == "scale":
                (v,) = pick(q)
                al = float(rng.choice([0.5, -1.0, 2.0, 0.0]))
                trace.append(f"VecScale({v}, {al})")
                L.VecScale(H[v], al); M[v] = al * M[v]; mag[v] = abs(al) * mag[v]
            elif op == "set":
                (v,) = pick(q)
                al = float(rng.choice([0.0, 1.0, -2.5]))
                trace.append(f"VecSet({v}, {al})")
                L.VecSet(H[v], al); M[v] = np.full_like(M[v], al); mag[v] = abs(al)
            elif op == "copy":
                a, bb = pick(q, 2)
                trace.append(f"VecCopy({a}, {bb})")
                L.VecCopy(H[a], H[bb]); M[bb] = M[a].copy(); mag[bb] = mag[a]
            elif op == "norm":
                (v,) = pick(q)
                trace.append(f"VecNorm({v})")
                val = C.c_double()
                L.VecNorm(H[v], NORM_2, C.byref(val))
                assert abs(val.value - np.linalg.norm(M[v])) <= 1e-11 * np.sqrt(M[v].size) * max(mag[v], 1e-30) + 1e-300, (f"seed {seed} program {prog}: norm of {v}", trace)
            elif op == "solve":
                bn, xn = pick(q, 2)
                trace.append(f"KSPSolve(k{q}, {bn}, {xn})   [max_it={kst[q]['maxit']} scale={kst[q]['scale']} guess={kst[q]['guess']}]")
                solve(q, bn, xn)
            elif op == "param":
                what = str(rng.choice(["maxit", "scale", "guess"]))
                if what == "maxit":
                    kst[q]["maxit"] = int(rng.integers(0, 5))
                    L.KSPSetTolerances(K[q], 1e-7, -2.0, -2.0, kst[q]["maxit"])
                elif what == "scale":
                    kst[q]["scale"] = float(rng.choice([0.5, 0.8, 1.0]))
                    L.KSPRichardsonSetScale(K[q], kst[q]["scale"])
                else:
                    kst[q]["guess"] = int(rng.integers(0, 2))
                    L.KSPSetInitialGuessNonzero(K[q], kst[q]["guess"])
                trace.append(f"k{q}.{what} = {kst[q][what]}")
            elif op == "read":
                (v,) = pick(q)
                trace.append(f"read {v}")
                check(v, "read")
            elif op == "recreate":
                (v,) = pick(q, 1, exclude=(kst[q]["b"], kst[q]["x"]))       # (the solver's vectors stay: KSPBuildResidual refers to them)
                trace.append(f"VecDestroy({v}); VecDuplicate; VecSet(1.5)")
                L.VecDestroy(C.byref(H[v]))
                h = C.c_void_p()
                L.VecDuplicate(first_f if q == 0 else first_c, C.byref(h))
                H[v] = h
                L.VecSet(h, 1.5); M[v] = np.full_like(M[v], 1.5); mag[v] = 1.5
            elif op == "frag_down":          # src/solver.c:1534-1536
                if kst[0]["b"] is None:
                    continue
                (rn_,) = pick(0, 1, exclude=(kst[0]["b"], kst[0]["x"]))
                bc_, uc_ = pick(1, 2)
                trace.append(f"down: KSPBuildResidual(k0, {rn_}); MatMult(R, {rn_}, {bc_}); KSPSolve(k1, {bc_}, {uc_})")
                residual(0, kst[0]["b"], kst[0]["x"], rn_, True)
                L.MatMult(mR, H[rn_], H[bc_]); M[bc_] = R @ M[rn_]; mag[bc_] = mag[rn_]
                solve(1, bc_, uc_)
            elif op == "frag_up":            # :1540-1542
                (uc_,) = pick(1)
                rv_, u_, b_ = pick(0, 3)
                if kst[0]["x"] is not None and rng.integers(0, 2):
                    u_, b_ = kst[0]["x"], kst[0]["b"]
                    if rv_ in (u_, b_):
                        (rv_,) = pick(0, 1, exclude=(u_, b_))
                trace.append(f"up: MatMult(P, {uc_}, {rv_}); VecAXPY({u_}, 1, {rv_}); KSPSolve(k0, {b_}, {u_})")
                L.MatMult(mP, H[uc_], H[rv_]); M[rv_] = P @ M[uc_]; mag[rv_] = mag[uc_]
                L.VecAXPY(H[u_], 1.0, H[rv_]); M[u_] = M[u_] + M[rv_]; mag[u_] = mag[u_] + mag[rv_]
                solve(0, b_, u_)
            elif op == "frag_norm":          # :1545-1546 and the next :1531
                if kst[0]["b"] is None:
                    continue
                (rn_,) = pick(0, 1, exclude=(kst[0]["b"], kst[0]["x"]))
                trace.append(f"norm: KSPBuildResidual(k0, {rn_}); VecNorm({rn_})" + ("; KSPSolve(k0, same b, same x)" if step % 2 else ""))
                residual(0, kst[0]["b"], kst[0]["x"], rn_, True)
                val = C.c_double()
                L.VecNorm(H[rn_], NORM_2, C.byref(val))
                assert abs(val.value - np.linalg.norm(M[rn_])) <= 1e-11 * np.sqrt(nf) * max(mag[rn_], 1e-30) + 1e-300, (f"seed {seed} program {prog}: residual norm", trace)
                if step % 2:
                    solve(0, kst[0]["b"], kst[0]["x"])
            elif op == "frag_cycle":         # one whole two-level cycle of the reference's loop on fixed vectors
                trace.append("cycle: KSPSolve(k0,b,u); KSPBuildResidual(k0,r); MatMult(R,r,bc); KSPSolve(k1,bc,uc); MatMult(P,uc,rv); VecAXPY(u,1,rv); KSPSolve(k0,b,u); "
                             "KSPBuildResidual(k0,r); VecNorm(r)")
                solve(0, "b", "u")
                residual(0, "b", "u", "r", True)
                L.MatMult(mR, H["r"], H["bc"]); M["bc"] = R @ M["r"]; mag["bc"] = mag["r"]
                solve(1, "bc", "uc")
                L.MatMult(mP, H["uc"], H["rv"]); M["rv"] = P @ M["uc"]; mag["rv"] = mag["uc"]
                L.VecAXPY(H["u"], 1.0, H["rv"]); M["u"] = M["u"] + M["rv"]; mag["u"] = mag["u"] + mag["rv"]
                solve(0, "b", "u")
                residual(0, "b", "u", "r", True)
                val = C.c_double()
                L.VecNorm(H["r"], NORM_2, C.byref(val))
                assert abs(val.value - np.linalg.norm(M["r"])) <= 1e-11 * np.sqrt(nf) * max(mag["r"], 1e-30) + 1e-300, (f"seed {seed} program {prog}: cycle norm", trace)
            if max(mag.values()) > 1e200:        # (a draw that keeps multiplying by A: start the next program)
                break
        for v in fine + coarse:
            check(v, "final value")
        for q in (0, 1):
            L.KSPDestroy(C.byref(K[q]))
        for v in fine + coarse:
            L.VecDestroy(C.byref(H[v]))
        L.VecDestroy(C.byref(first_f)); L.VecDestroy(C.byref(first_c))
        for m in (mA0, mA1, mP, mR):
            L.MatDestroy(C.byref(m))
    import os
    if os.environ.get("MGPETSC_LAZY_STATS"):       # which of the drop-in's fast paths the draws went through (printed by PetscFinalize)
        L.PetscFinalize.argtypes = []
        L.PetscFinalize()


if __name__ == "__main__":      # python tests/shim_semantics.py <shared library> <function name>: one check in a process of its own
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import Oracle
    lib = type_shim(C.CDLL(sys.argv[1], mode=os.RTLD_LOCAL))
    {"lazy": lazy_temporaries_keep_petsc_semantics, "spec": speculative_sweep_is_adopted_only_when_nothing_changed,
     "keepr": residual_left_deferred_by_the_norm_pass, "tailrec": recorded_coarse_subcycle_keeps_petsc_semantics, "pcmgtail": pcmg_level_vectors_after_the_tail_launch, "lu": richardson_with_lu_is_damped_not_exact,
     "random": lambda L_, o_: random_programs_keep_petsc_semantics(L_, o_, seed=int(sys.argv[3]) if len(sys.argv) > 3 else 1,
                                                                   nprog=int(sys.argv[4]) if len(sys.argv) > 4 else 40)}[sys.argv[2]](lib, Oracle())
    print("SEMANTICS_OK", sys.argv[2])
