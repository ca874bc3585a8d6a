import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.solver import Solver
for npts, levels in ((1025, 10), (513, 9)):
    for ksp, kw in (("richardson", dict(scale=6.0 / 7.0)), ("chebyshev", dict(eigenvalues=(0.3, 2.0)))):
        s = Solver(3, npts, levels, maxiter=40, ksp_type=ksp, **kw)
        s.set_rhs_problem(); s.cycles(2); s.sync()
        t0 = time.perf_counter(); s.cycles(8); s.sync(); ms = 1e3 * (time.perf_counter() - t0) / 8
        rn = s.rnorm
        print(f"3-D npts {npts} {ksp}: {ms:.3f} ms/cycle, contraction {(rn[-1]/rn[-5])**0.25:.3f}", flush=True)
        s.close()
