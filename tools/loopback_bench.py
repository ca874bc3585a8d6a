"""P loopback ranks (threads sharing ONE GPU) over the slab solver: a functional timing of the multi-rank code path
(all ranks serialise on the one device, so the time is the SUM of the per-rank work plus the exchange overheads --
useful to see regressions of the slab path, not to predict multi-GPU speed).   usage: loopback_bench.py [npts] [P ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.solver import Solver
from multigrid_petsc_amd.comm import LoopbackWorld

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 513
Ps = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
levels = {1025: 10, 513: 9, 257: 8, 129: 7}[npts]
for P in Ps:
    for fuse in (31, 63):                         # 31: one sweep per launch; 63: sweeps in pairs where possible
        def fn(rank, comm):
            s = Solver(3, npts, levels, scale=6 / 7, maxiter=40, rank=rank, nranks=P, comm=comm, fuse=fuse)
            s.set_rhs_problem(); s.cycles(2); s.sync()
            t = time.perf_counter(); s.cycles(8); s.sync()
            dt = (time.perf_counter() - t) / 8
            r = s.rnorm[-1]
            s.close()
            return dt, r
        if P == 1:
            s = Solver(3, npts, levels, scale=6 / 7, maxiter=40, fuse=fuse)
            s.set_rhs_problem(); s.cycles(2); s.sync()
            t = time.perf_counter(); s.cycles(8); s.sync()
            res = [((time.perf_counter() - t) / 8, s.rnorm[-1])]; s.close()
        else:
            w = LoopbackWorld(P)
            try:
                res = w.run(fn)
            finally:
                w.close()
        print(f"npts {npts} P {P} fuse {fuse}: {max(r[0] for r in res) * 1e3:.2f} ms/cycle  rnorm {res[0][1]:.6e}", flush=True)
