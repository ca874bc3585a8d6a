"""rank 3 of 8 (phantom neighbours, free link model) for a rocprofv3 kernel trace: 12 cycles of the 1023^3 V(3,3) cycle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.comm import phantom_comm
from multigrid_petsc_amd.solver import Solver
import sys as _s
c = phantom_comm(3, 8, float(_s.argv[1]) if len(_s.argv) > 1 else 0.0, float(_s.argv[2]) if len(_s.argv) > 2 else 0.0)
s = Solver(3, 1025, 10, scale=6.0 / 7.0, maxiter=100, rank=3, nranks=8, comm=c.handle)
s.set_rhs_problem(); s.cycles(3); s.sync(); s.cycles(12); s.sync(); s.close(); c.close()
