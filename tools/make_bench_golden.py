#!/usr/bin/env python3
"""Writes tests/golden/bench_history.json: the normalised residual histories rnorm[i]/rnorm[0], i = 0..14, of the four
single-GPU bench configurations, produced on an MI355X by the KERNEL-PER-OPERATION cycle (fuse=0: one launch per PETSc
operation of the reference loop, the form the oracle tests pin bit for bit at small sizes).  bench.py compares every run's
history with these; the fused default cycle must reproduce them (fields are bit-identical, norms equal to rounding)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigrid_petsc_amd.solver import Solver          # noqa: E402

out = {}
for dim, npts, precision in ((3, 1025, "fp64"), (2, 4097, "fp64"), (3, 513, "fp64"), (3, 1025, "mixed")):
    levels = 0
    while (npts - 1) % (2 ** levels) == 0 and (npts - 1) // (2 ** levels) - 1 >= 1:
        levels += 1
    s = Solver(dim, npts, levels, v=(3, 3), maxiter=20, scale=6.0 / 7.0 if dim == 3 else 0.8, precision=precision, fuse=0)
    s.set_rhs_problem()
    s.cycles(14)
    s.sync()
    rn = s.rnorm
    out[f"{dim}d_{npts}_{precision}"] = [float(x / rn[0]) for x in rn]
    s.close()
    print(dim, npts, precision, ["%.3e" % x for x in out[f"{dim}d_{npts}_{precision}"][:6]], file=sys.stderr)
with open(os.path.join(ROOT, "tests", "golden", "bench_history.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
