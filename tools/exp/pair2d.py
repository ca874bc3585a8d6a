import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.solver import Solver
for npts, levels in ((4097, 12), (1025, 10)):
    for pm in (0, 1023, 511, 255, 127):
        s = Solver(2, npts, levels, scale=0.8, maxiter=40, pair_min_n=pm)
        s.set_rhs_problem(); s.cycles(3); s.sync()
        t0 = time.perf_counter(); s.cycles(20); s.sync(); ms = 1e3 * (time.perf_counter() - t0) / 20
        print(f"2-D npts {npts} pair_min_n {pm or 'default'}: {ms:.4f} ms/cycle", flush=True)
        s.close()
