#!/usr/bin/env python3
"""Randomised z-slab configurations (loopback ranks = threads sharing one GPU) against the oracle: rank count, size, depth, where the hierarchy
stops being distributed, sweep counts, fuse bits, overlap, chunk hint, pair threshold, precision drawn at random; every rank's iteration count must
agree with the oracle's and the concatenated slabs must be bit-identical to its u.  A one-off stress run, not part of the suite.
MG_STRESS_MOCK=1: the same draws on the CPU -- mg_solver.c + mg_comm.c over tests/mock_mgk.cpp (tools/stress_solver_mock.py injects the library),
sizes bounded by MG_STRESS_MAXN (the mock is scalar host code).
usage: stress_slabs.py [count] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("MG_STRESS_MOCK"):
    from stress_solver_mock import inject
    inject()
from multigrid_petsc_amd.solver import Solver
from multigrid_petsc_amd.comm import LoopbackWorld
from oracle import Oracle

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = Oracle()
bad = 0
for q in range(count):
    npts = int(rng.choice([n for n in (33, 65, 65, 129) if n <= int(os.environ.get("MG_STRESS_MAXN", "129"))]))
    lmax = int(np.log2(npts - 1))
    levels = int(rng.integers(2, lmax + 1))
    P = int(rng.choice([2, 3, 4, 8]))
    dist = int(rng.choice([7, 15, 31, 63]))
    while (npts - 2) // P < 4 or dist > npts - 2:
        P = int(rng.choice([2, 3, 4])); dist = int(rng.choice([7, 15, 31]))
    v0, v1 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    fuse = int(rng.choice([-1, -1, -1, 0, 63, 63 | 256 | 512, 63 | 256 | 512 | 1024 | 2048, int(rng.integers(0, 16384)), 32767, 32767, int(rng.integers(0, 32768)) | 16384]))   # bit 14: the 91-byte fine level on slabs
    kw = dict(fuse=fuse, overlap=int(rng.choice([-1, 0, 1])), pair_min_n=int(rng.choice([0, 0, 7, 15])), slab_chunk=int(rng.choice([-1, 0, 4, 8])),
              precision=str(rng.choice(["fp64", "fp64", "mixed"])))
    scale = 6.0 / 7.0
    tag = f"P={P} npts={npts} levels={levels} dist_min_n={dist} v=({v0},{v1}) {kw}"
    world = LoopbackWorld(P)

    # solve to the tolerance | the bench's loop: a fixed number of cycles in two calls, norms deferred to the end (what the N-GPU run executes)
    k1, k2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    fixed = (k1 + k2) if rng.integers(0, 3) == 0 else 0
    tag += f" {'cycles ' + str(k1) + '+' + str(k2) if fixed else 'solve'}"

    def fn(rank, comm):
        s = Solver(3, npts, levels, v=(v0, v1), scale=scale, maxiter=40, rank=rank, nranks=P, comm=comm, dist_min_n=dist, **kw)
        s.set_rhs_problem()
        if fixed:
            s.cycles(k1)
            s.cycles(k2)
            s.sync()
            it = s.iterations
        else:
            it = s.solve()
        res = (it, s.solution(), s.rnorm)
        s.close()
        return res

    try:
        res = world.run(fn)
    except Exception as e:
        print("REFUSED", tag, str(e)[:160], flush=True)
        continue
    finally:
        world.close()
    if kw["precision"] == "mixed":
        ref = orc.vcycle_mixed(npts, levels, v0, v1, maxiter=40, scale=scale, fixed_cycles=fixed)
    else:
        ref = orc.vcycle(3, npts, levels, v0, v1, maxiter=40, scale=scale, fixed_cycles=fixed)
    u = np.concatenate([r[1] for r in res])
    ok = (all(r[0] == ref["iters"] for r in res) and np.array_equal(u, ref["u"]) and
          all(np.max(np.abs(r[2] - ref["rnorm"]) / np.maximum(ref["rnorm"], 1e-300)) <= 1e-10 for r in res))
    if not ok:
        bad += 1
        print("MISMATCH", tag, "iters", [r[0] for r in res], ref["iters"], "max|du|", float(np.max(np.abs(u - ref["u"]))), flush=True)
    elif q % 10 == 0:
        print("ok", q, tag, flush=True)
print(f"{count} configurations, {bad} mismatches")
sys.exit(1 if bad else 0)
