#!/usr/bin/env python3
"""Generator of tests/golden/vcycle_golden.npz.

An INDEPENDENT third restatement of the `-cycle 0` V-cycle (src/solver.c:1414-1575), written with
scipy.sparse Kronecker products instead of the reference's MatSetValue loops or the oracle's stencil
loops.  It shares no code with oracle/ or with the product; it runs only in the development container
(scipy is not needed on the GPU box) and its outputs are committed as data.

    python tests/golden/make_golden.py          # rewrites tests/golden/vcycle_golden.npz

What is restated (reference file:line):
  operator rows      src/solver.c:185-253 + src/problem.c:3-22 (uniform mesh: 1/h^2 second differences)
  restriction        src/solver.c:1035-1094, weights src/matbuild.c:422-431 ([1 2 1]/4 per axis)
  prolongation       src/solver.c:1096-1154, weights src/matbuild.c:398-407 (= 2^d R^T)
  right-hand side    src/solver.c:558-620, src/problem.c:24-28, coordinates by repeated addition src/mesh.c:150-152
  smoother           KSPRICHARDSON + PCJACOBI, KSP_NORM_NONE, max_it sweeps, zero-fill on a zero guess
  cycle + stop rule  src/solver.c:1511-1550
  error norms        src/solver.c:1211-1237

PETSc itself is not available, so these vectors do not come from the reference
binary: they pin the oracle and the GPU path against a restatement made with different tools.
The 3-D cases extend the 2-D semantics (the reference has DIMENSION 2 only).
"""
import math
import os

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))


def second_difference(n):
    return sp.diags([np.ones(n - 1), -2.0 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr")


def full_weighting_1d(nf):
    nc = (nf - 1) // 2
    rows = np.repeat(np.arange(nc), 3)
    cols = (2 * np.arange(nc)[:, None] + np.arange(3)[None, :]).ravel()
    vals = np.tile([0.25, 0.5, 0.25], nc)
    return sp.csr_matrix((vals, (rows, cols)), shape=(nc, nf))


def kron_all(mats):
    out = mats[0]
    for m in mats[1:]:
        out = sp.kron(out, m, format="csr")
    out.sum_duplicates()
    out.sort_indices()
    return out


def level_operator(dim, n):
    h = 1.0 / (n + 1)
    T, I = second_difference(n) * (1.0 / (h * h)), sp.identity(n, format="csr")
    A = None
    for axis in range(dim):                       # slowest axis first, x (fastest) last
        term = kron_all([T if a == axis else I for a in range(dim)])
        A = term if A is None else A + term
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    return A


def coords_y(npts, mesh):
    """y coordinates of the stretched meshes, src/mesh.c:154-176 (x stays uniform)"""
    if mesh == 0:
        return coords(npts)
    c = [0.0] * npts
    c[-1] = 1.0
    for j in range(1, npts - 1):
        eta = j / float(npts - 1)
        c[j] = 1.0 - 1.0 * math.cos(math.pi * 0.5 * eta) if mesh == 1 else 0.0 + 1.0 * ((math.exp(2 * eta) - 1) / (math.exp(2) - 1))
    return c


def metrics(mesh, y):
    """MetricsNonUniform1 / 2 on [0,1]^2, src/mesh.c:45-107"""
    if mesh == 1:
        temp = (1.0 - 0.0) * (1.0 - 0.0) - (1.0 - y) * (1.0 - y)
        return [1.0, 4.0 / (math.pi * math.pi * temp), 0.0, (-2.0 * (1.0 - y)) / (math.pi * math.sqrt(temp * temp * temp)), 0.0]
    e2 = math.exp(2) - 1
    temp = (e2 * e2) / (((y - 0.0) * e2 + 1.0) * ((y - 0.0) * e2 + 1.0))
    return [1.0, 0.25 * temp, 0.0, (-0.5) * temp, 0.0]


def level_operator_mesh(npts, l, mesh):
    """rows of level l on a stretched mesh (2-D): metrics at the fine-grid point of the grid row (src/solver.c:227-232),
    OpA with the level's computational spacing (src/problem.c:3-22)"""
    n = (npts - 1) // 2 ** l - 1
    h = 1.0 / (n + 1)
    cy = coords_y(npts, mesh)
    a = np.zeros((5, n))
    for i in range(n):
        m = metrics(mesh, cy[2 ** l * (i + 1)])
        a[0, i] = (m[1] / (h * h)) - (m[3] / (2 * h))
        a[1, i] = (m[0] / (h * h)) - (m[2] / (2 * h))
        a[2, i] = -2.0 * ((m[0] / (h * h)) + (m[1] / (h * h)))
        a[3, i] = (m[0] / (h * h)) + (m[2] / (2 * h))
        a[4, i] = (m[1] / (h * h)) + (m[3] / (2 * h))
    I = sp.identity(n, format="csr")
    lo, up = sp.diags([np.ones(n - 1)], [-1], format="csr"), sp.diags([np.ones(n - 1)], [1], format="csr")
    A = (sp.kron(sp.diags(a[0]) @ lo, I) + sp.kron(sp.diags(a[1]), lo) + sp.kron(sp.diags(a[2]), I)
         + sp.kron(sp.diags(a[3]), up) + sp.kron(sp.diags(a[4]) @ up, I))
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    return A


def coords(npts):
    c = [0.0]
    d = 1.0 / (npts - 1)
    for _ in range(1, npts):
        c.append(c[-1] + d)                       # repeated addition, src/mesh.c:150-152
    return c


def rhs(dim, npts):
    c = coords(npts)
    n = npts - 2
    s = [math.sin(math.pi * c[j + 1]) for j in range(n)]
    b = np.empty(n ** dim)
    f = -dim * math.pi * math.pi
    if dim == 2:
        for i in range(n):
            for j in range(n):
                b[i * n + j] = f * s[j] * s[i]
    else:
        for k in range(n):
            for i in range(n):
                for j in range(n):
                    b[(k * n + i) * n + j] = f * s[j] * s[i] * s[k]
    return b


def rhs_mesh(npts, mesh):
    cx, cy = coords(npts), coords_y(npts, mesh)
    n = npts - 2
    b = np.empty(n * n)
    f = -2 * math.pi * math.pi
    for i in range(n):
        for j in range(n):
            b[i * n + j] = f * math.sin(math.pi * cx[j + 1]) * math.sin(math.pi * cy[i + 1])
    return b


def exact_mesh(npts, mesh):
    cx, cy = coords(npts), coords_y(npts, mesh)
    n = npts - 2
    sx = np.array([math.sin(math.pi * cx[j + 1]) for j in range(n)])
    sy = np.array([math.sin(math.pi * cy[i + 1]) for i in range(n)])
    return np.multiply.outer(sy, sx).ravel()


def exact(dim, npts):
    c = coords(npts)
    n = npts - 2
    s = np.array([math.sin(math.pi * c[j + 1]) for j in range(n)])
    u = s
    for _ in range(dim - 1):
        u = np.multiply.outer(s, u)
    return u.ravel()


def richardson(A, dinv, b, x, maxit, scale, guess_nonzero):
    if maxit <= 0:
        return x
    if not guess_nonzero:
        x = np.zeros_like(b)
        r = b.copy()
    else:
        r = b - A @ x
    for it in range(maxit):
        z = r * dinv
        x = x + scale * z
        if it + 1 < maxit:
            r = b - A @ x
    return x


def vcycle(dim, npts, levels, v0, v1, scale, maxiter=400, rtol=1e-7, mesh=0):
    ns = [(npts - 1) // 2 ** l - 1 for l in range(levels)]
    A = [level_operator(dim, n) for n in ns] if mesh == 0 else [level_operator_mesh(npts, l, mesh) for l in range(levels)]
    dinv = [1.0 / a.diagonal() for a in A]
    R = [kron_all([full_weighting_1d(ns[l])] * dim) for l in range(levels - 1)]
    P = [sp.csr_matrix((2.0 ** dim) * r.T) for r in R]
    for p in P:
        p.sort_indices()
    b = [rhs(dim, npts) if mesh == 0 else rhs_mesh(npts, mesh)] + [None] * (levels - 1)
    u = [np.zeros(n ** dim) for n in ns]
    bnorm = math.sqrt(float(np.dot(b[0], b[0])))
    r0 = b[0] - A[0] @ u[0]
    rnorm = [math.sqrt(float(np.dot(r0, r0)))]
    it = 0
    nonzero = [False] * levels
    while it < maxiter and rnorm[-1] > rtol * bnorm:
        u[0] = richardson(A[0], dinv[0], b[0], u[0], v0, scale, nonzero[0])
        nonzero[0] = True                                          # src/solver.c:1532-1533
        for l in range(1, levels):
            b[l] = R[l - 1] @ (b[l - 1] - A[l - 1] @ u[l - 1])
            u[l] = richardson(A[l], dinv[l], b[l], u[l], v0 if l < levels - 1 else v1, scale, False)
        for l in range(levels - 2, -1, -1):
            u[l] = u[l] + P[l] @ u[l + 1]
            u[l] = richardson(A[l], dinv[l], b[l], u[l], v0, scale, True)
        r = b[0] - A[0] @ u[0]
        rnorm.append(math.sqrt(float(np.dot(r, r))))
        it += 1
    return dict(iters=it, bnorm=bnorm, rnorm=np.array(rnorm), u=u[0], b=b[0])


def error_norms(dim, npts, u):
    d = np.abs(u - exact(dim, npts))              # max, plain sum, sqrt of plain sum of squares (no 1/N)
    return np.array([d.max(), d.sum(), math.sqrt(float(np.dot(d, d)))])


def maps(npts, levels, procs):
    """One grid per level (-grids == -levels): lexicographic identity map, ranges of src/matbuild.c:120-144."""
    out = []
    for l in range(levels):
        n = (npts - 1) // 2 ** l - 1
        tot = n * n
        q, rem = divmod(tot, procs)
        rg = [0]
        for p in range(procs):
            rg.append(rg[-1] + q + (1 if p < rem else 0))
        out.append(rg)
    return out


CASES = [  # dim, npts, levels, scale
    (2, 17, 2, 1.0), (2, 17, 2, 0.8), (2, 17, 4, 0.8),
    (2, 33, 2, 0.8), (2, 33, 5, 1.0), (2, 33, 5, 0.8),
    (2, 129, 2, 0.8), (2, 129, 7, 0.8), (2, 129, 7, 1.0),
    (3, 9, 2, 6.0 / 7.0), (3, 9, 3, 6.0 / 7.0), (3, 17, 2, 6.0 / 7.0), (3, 17, 4, 6.0 / 7.0), (3, 17, 4, 1.0),
    (3, 33, 5, 6.0 / 7.0), (3, 33, 2, 6.0 / 7.0),
]


def main():
    out = {}
    rng = np.random.default_rng(0x5EED0001)
    for dim, npts, levels, scale in CASES:
        maxiter = 60 if levels == 2 else 400       # two-level cycles converge slowly: keep a fixed prefix
        r = vcycle(dim, npts, levels, 3, 3, scale, maxiter=maxiter)
        key = "d%d_n%d_l%d_s%d" % (dim, npts, levels, round(scale * 1000))
        out[key + "_meta"] = np.array([dim, npts, levels, 3, 3, maxiter, r["iters"]], dtype=np.int64)
        out[key + "_scale"] = np.array([scale, r["bnorm"]])
        out[key + "_rnorm"] = r["rnorm"]
        out[key + "_err"] = error_norms(dim, npts, r["u"])
        out[key + "_u"] = r["u"]
        if levels == 2 and scale != 1.0:
            out["b0_d%d_n%d" % (dim, npts)] = r["b"]
        print(key, "cycles", r["iters"], "rel", r["rnorm"][-1] / r["rnorm"][0], "err", out[key + "_err"])
    for mesh, npts, levels in ((1, 33, 4), (2, 33, 4), (1, 129, 6), (2, 65, 5)):
        r = vcycle(2, npts, levels, 3, 3, 0.8, maxiter=1000, mesh=mesh)
        key = "mesh%d_n%d_l%d" % (mesh, npts, levels)
        d = np.abs(r["u"] - exact_mesh(npts, mesh))
        out[key + "_meta"] = np.array([2, npts, levels, 3, 3, 1000, r["iters"], mesh], dtype=np.int64)
        out[key + "_scale"] = np.array([0.8, r["bnorm"]])
        out[key + "_rnorm"] = r["rnorm"]
        out[key + "_err"] = np.array([d.max(), d.sum(), math.sqrt(float(np.dot(d, d)))])
        out[key + "_u"] = r["u"]
        print(key, "cycles", r["iters"], "rel", r["rnorm"][-1] / r["rnorm"][0])
    for dim, nf in ((2, 31), (3, 15)):
        x = rng.uniform(-1, 1, nf ** dim)
        nc = (nf - 1) // 2
        R = kron_all([full_weighting_1d(nf)] * dim)
        P = sp.csr_matrix((2.0 ** dim) * R.T)
        P.sort_indices()
        xc = rng.uniform(-1, 1, nc ** dim)
        y = rng.uniform(-1, 1, nf ** dim)
        A = level_operator(dim, nf)
        out["xfer_d%d_fine" % dim] = x
        out["xfer_d%d_restricted" % dim] = R @ x
        out["xfer_d%d_coarse" % dim] = xc
        out["xfer_d%d_base" % dim] = y
        out["xfer_d%d_prolonged" % dim] = y + P @ xc
        out["xfer_d%d_applied" % dim] = A @ x
        out["xfer_d%d_residual" % dim] = y - A @ x
    for npts, levels in ((9, 3), (17, 4), (33, 5)):
        for procs in (1, 2, 4, 8):
            m = maps(npts, levels, procs)
            out["ranges_n%d_p%d" % (npts, procs)] = np.array(m, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "vcycle_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "vcycle_golden.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
