import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.solver import Solver
s = Solver(3, 1025, 10, maxiter=40, ksp_type="chebyshev", eigenvalues=(0.3, 2.0))
s.set_rhs_problem(); s.cycles(2); s.sync()
t0 = time.perf_counter(); s.cycles(5); s.sync(); print(1e3 * (time.perf_counter() - t0) / 5)
s.close()
