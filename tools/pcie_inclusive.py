"""The headline workload with its inputs and outputs in HOST memory: b handed over as a host array (mg_solver_set_rhs_host: h2d + pack into
the padded layout), solved to the reference's stopping rule, u fetched back (mg_solver_get_solution: unpack + d2h).  bench.py's `value`
starts with b resident in HBM; this prints the rate with both transfers inside the clock.
Usage: python tools/pcie_inclusive.py [npts] [levels]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.solver import Solver   # noqa: E402

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 10
s = Solver(3, npts, levels, scale=6.0 / 7.0)
n = npts - 2
x = np.sin(np.pi * np.arange(1, n + 1) / (n + 1))
b = (x[:, None, None] * x[None, :, None] * x[None, None, :]).reshape(-1)      # smooth separable right-hand side, host memory
s.set_rhs(b); s.solve(); s.sync(); s.reset()                                     # warm-up (allocations, graph capture)
t0 = time.perf_counter()
s.set_rhs(b)
s.sync()
t1 = time.perf_counter()
it = s.solve()
s.sync()
t2 = time.perf_counter()
u = s.solution()
t3 = time.perf_counter()
dof = s.dof_updates_per_cycle * it
gb = b.nbytes / 1e9
print(json.dumps({"npts": npts, "cycles": it, "upload_s": t1 - t0, "upload_GBs": gb / (t1 - t0), "solve_s": t2 - t1, "download_s": t3 - t2,
                  "download_GBs": gb / (t3 - t2), "dof_updates_per_s_resident": dof / (t2 - t1),
                  "dof_updates_per_s_host_to_host": dof / (t3 - t0), "rel_residual": float(s.rnorm[it] / s.rnorm[0])}), flush=True)
s.close()
