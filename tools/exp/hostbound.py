"""is the 2-D 4097^2 cycle bound by the host's launch rate?  enqueue time of N cycles against their total time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.solver import Solver
for dim, npts, levels, scale in ((2, 4097, 12, 0.8), (2, 2049, 11, 0.8), (3, 257, 8, 6.0 / 7.0)):
    for graph in (1, 0):
        s = Solver(dim, npts, levels, scale=scale, maxiter=100000, graph=graph)
        s.set_rhs_problem(); s.cycles(5); s.sync()
        t0 = time.perf_counter(); s.cycles(100); t1 = time.perf_counter(); s.sync(); t2 = time.perf_counter()
        print(f"dim {dim} npts {npts} graph {graph}: enqueue {(t1 - t0) * 10:.3f} ms/cycle, total {(t2 - t0) * 10:.3f} ms/cycle", flush=True)
        s.close()
