#!/usr/bin/env python3
"""The 91-byte fine level on z-slabs (fuse bit 14; loopback ranks = threads) against the oracle: solve and the bench's fixed-count loop, overlap on / off.
MOCK=1 (default): over the host mock, small sizes; MOCK=0: on the GPU (cases like 513,5,4,255,255 = npts, levels, ranks, dist_min_n, pair_min_n; ';' separates cases).
usage: stress_slab91.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("MOCK", "1") == "1":
    from stress_solver_mock import inject
    inject()
import numpy as np
from multigrid_petsc_amd.solver import Solver
from multigrid_petsc_amd.comm import LoopbackWorld
from oracle import Oracle
orc = Oracle()
FUSE = 63 | 256 | 512 | 1024 | 2048 | 4096 | 8192 | 16384
bad = 0
REFS = {}
cases = [(33, 4, 2, 7, 7), (33, 5, 2, 15, 7), (33, 4, 3, 7, 15), (65, 5, 2, 15, 15), (65, 6, 4, 7, 7), (65, 5, 3, 15, 31), (65, 4, 8, 7, 7)] if len(sys.argv) < 2 else [tuple(int(x) for x in c.split(",")) for c in sys.argv[1].split(";")]
for npts, levels, P, dist, pair in cases:
    for overlap in (1, 0):
        for mode in ("solve", "cycles"):
            world = LoopbackWorld(P)
            def fn(rank, comm):
                s = Solver(3, npts, levels, v=(3, 3), scale=6.0 / 7.0, maxiter=40, rank=rank, nranks=P, comm=comm, dist_min_n=dist, fuse=FUSE, pair_min_n=pair, overlap=overlap)
                s.set_rhs_problem()
                if mode == "solve":
                    it = s.solve()
                else:
                    s.cycles(2); s.cycles(3); s.sync(); it = s.iterations
                r = (it, s.solution(), s.rnorm)
                s.close()
                return r
            try:
                res = world.run(fn)
            finally:
                world.close()
            key = (npts, levels, mode)
            if key not in REFS:
                REFS[key] = orc.vcycle(3, npts, levels, 3, 3, maxiter=40, scale=6.0 / 7.0, fixed_cycles=0 if mode == "solve" else 5)
            ref = REFS[key]
            u = np.concatenate([r[1] for r in res])
            ok = all(r[0] == ref["iters"] for r in res) and np.array_equal(u, ref["u"]) and all(np.max(np.abs(r[2] - ref["rnorm"]) / ref["rnorm"]) < 1e-10 for r in res)
            print("OK " if ok else "BAD", npts, levels, P, dist, pair, "overlap", overlap, mode, [r[0] for r in res], ref["iters"], float(np.max(np.abs(u - ref["u"]))), flush=True)
            bad += 0 if ok else 1
print("bad", bad)
sys.exit(1 if bad else 0)
