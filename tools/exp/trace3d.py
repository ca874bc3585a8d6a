"""30 cycles of a 3-D V(3,3) solver (default npts 257) for a rocprofv3 kernel trace"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.solver import Solver
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 257
lv = {129: 7, 257: 8, 513: 9}[npts]
s = Solver(3, npts, lv, scale=6.0 / 7.0, maxiter=100000)
s.set_rhs_problem(); s.cycles(5); s.sync(); s.cycles(30); s.sync(); s.close()
