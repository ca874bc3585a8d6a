"""debugging aid: mixed-precision cycle time vs the marching chunk length (global override)"""
import sys, time
sys.path.insert(0, ".")
from multigrid_petsc_amd.solver import Solver
from multigrid_petsc_amd._lib import load_mgk
L = load_mgk()
for zc in (-1, 512, 342, 256):
    L.mgk_set_tuning(-1, zc)
    s = Solver(3, 1025, 10, scale=6 / 7, maxiter=30, precision="mixed")
    s.set_rhs_problem(); s.cycles(2); s.sync()
    t = time.perf_counter(); s.cycles(10); s.sync()
    print("zchunk", zc, "%.2f ms/cycle" % ((time.perf_counter() - t) * 100), flush=True)
    s.close()
