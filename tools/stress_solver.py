#!/usr/bin/env python3
"""Randomised configurations of the own solver against the oracle (a one-off stress run, not part of the suite): dimension, size, depth,
sweep counts, damping, mesh, pair threshold and fuse bits drawn at random; the iteration count must agree and u must be bit-identical.
usage: stress_solver.py [count] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from multigrid_petsc_amd.solver import Solver
from oracle import Oracle

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = Oracle()
bad = 0
for q in range(count):
    dim = int(rng.choice([2, 2, 3]))
    npts = int(rng.choice([9, 17, 33, 65, 129, 257, 513, 1025] if dim == 2 else [9, 17, 33, 65, 129]))
    lmax = int(np.log2(npts - 1))
    levels = int(rng.integers(1, lmax + 1))
    v0, v1 = int(rng.integers(0, 5)), int(rng.integers(1, 5))
    mesh = int(rng.choice([0, 0, 1, 2])) if dim == 2 else 0
    scale = float(rng.choice([0.8, 1.0, 6.0 / 7.0, 0.5]))
    fuse = int(rng.choice([-1, -1, -1, 0, 63, 63 | 256 | 512, 63 | 256 | 512 | 1024 | 2048, 63 | 256 | 512 | 8192, int(rng.integers(0, 16384))]))
    pair = int(rng.choice([0, 0, 7, 15, 31]))
    graph = int(rng.choice([-1, -1, 0]))
    prec = str(rng.choice(["fp64", "fp64", "mixed"])) if dim == 3 else "fp64"
    if v0 == 0 and levels > 1:
        v0 = 1
    tag = f"dim={dim} npts={npts} levels={levels} v=({v0},{v1}) mesh={mesh} scale={scale:.4f} fuse={fuse} pair_min_n={pair} graph={graph} {prec}"
    try:
        s = Solver(dim, npts, levels, v=(v0, v1), maxiter=60, scale=scale, fuse=fuse, pair_min_n=pair, mesh=mesh, graph=graph, precision=prec)
        s.set_rhs_problem()
        it = s.solve()
        u = s.solution()
        rn = s.rnorm
        s.close()
    except Exception as e:                                   # a configuration the solver refuses is reported, not counted as a mismatch
        print("REFUSED", tag, str(e)[:120], flush=True)
        continue
    if prec == "mixed":
        ref = orc.vcycle_mixed(npts, levels, v0, v1, maxiter=60, scale=scale)
    else:
        ref = orc.vcycle(dim, npts, levels, v0, v1, maxiter=60, scale=scale, use_csr=1 if mesh else 0, mesh=mesh)
    ok = it == ref["iters"] and np.array_equal(u, ref["u"]) and np.max(np.abs(rn - ref["rnorm"]) / np.maximum(ref["rnorm"], 1e-300)) <= 1e-10
    if not ok:
        bad += 1
        print("MISMATCH", tag, "iters", it, ref["iters"], "max|du|", float(np.max(np.abs(u - ref["u"]))), flush=True)
print(f"{count} configurations, {bad} mismatches")
sys.exit(1 if bad else 0)
