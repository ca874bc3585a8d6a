"""30 cycles of the 2-D 4097^2 V(3,3) solver for a rocprofv3 kernel trace"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.solver import Solver
s = Solver(2, 4097, 12, scale=0.8, maxiter=100000)
s.set_rhs_problem(); s.cycles(5); s.sync(); s.cycles(30); s.sync(); s.close()
