"""stretched meshes (-mesh 1/2, 2-D) in the own driver: ms per V(3,3) cycle with the fused row-table cycle (default) and with the
kernel-per-operation cycle (fuse=0), beside the uniform mesh.  Usage: python tools/bench_mesh.py [npts] [levels]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.solver import Solver   # noqa: E402

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for mesh in (0, 1, 2):
    for fuse in (-1, 0):
        s = Solver(2, npts, levels, scale=0.8, mesh=mesh, fuse=fuse, maxiter=100000)
        s.set_rhs_problem()
        s.cycles(5)
        s.sync()
        t0 = time.perf_counter()
        s.cycles(50)
        s.sync()
        ms = (time.perf_counter() - t0) * 1e3 / 50
        s.reset()
        it = s.solve()
        print(f"mesh={mesh} fuse={'default' if fuse < 0 else 0}: {ms:.4f} ms/cycle; solve: {it} cycles in {s.solve_seconds:.4f} s", flush=True)
        s.close()
