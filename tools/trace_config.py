#!/usr/bin/env python3
"""a few cycles of one BASELINE configuration for a rocprofv3 kernel trace:  rocprofv3 --kernel-trace -d DIR -- python3 tools/trace_config.py 2 4097 fp64
then  python3 tools/trace_cycle.py DIR/.../*_results.db 5"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.solver import Solver
dim, npts, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "fp64"
levels = 0
while (npts - 1) % (2 ** levels) == 0 and (npts - 1) // (2 ** levels) - 1 >= 1:
    levels += 1
s = Solver(dim, npts, levels, scale=6.0 / 7.0 if dim == 3 else 0.8, maxiter=100, precision=prec, pair_min_n=int(os.environ.get("MG_PAIR_MIN_N", "0")), mesh=int(os.environ.get("MG_MESH", "0")))
if os.environ.get("MG_TUNE"):          # "variant,zchunk": a tuning experiment over every launch of the cycle (mgk_set_tuning)
    from multigrid_petsc_amd.mgk import load_mgk as load
    v, z = (int(t) for t in os.environ["MG_TUNE"].split(","))
    load().mgk_set_tuning(v, z)
s.set_rhs_problem()
s.cycles(3)
s.sync()
s.cycles(8)
s.sync()
s.close()
