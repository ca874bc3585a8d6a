"""The fine-level kernels of the 2-D cycle at 4095^2 for the PMC (HBM traffic) passes of rocprofv3 (as tools/pmc_sweep.py for 3-D):
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 tools/pmc_sweep_2d.py ; again with --pmc WRITE_SIZE.
NOTE: a 4095^2 field is 134 MB -- two of the three operands of a pass fit the 256 MB Infinity Cache, whose hits the counters include
(MI355X_MICROARCH.md): the figures say how many bytes cross the L2's memory side, not how many come from HBM."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4095
m = Mgk(0); L = m.L
g, gc = m.geom(2, n), m.geom(2, (n - 1) // 2)
rng = np.random.default_rng(0)
r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(2)]
u, b, out, uc, bc = m.field(g), m.field(g), m.field(g), m.field(gc), m.field(gc)
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[1], u, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[1], r1[0], r1[0], b, None))
q = float((n + 1) ** 2)
coef, dinv = m.coef([q, q, -4 * q, q, q]), -1.0 / (4 * q)
ss = C.c_double()
G, GC = C.byref(g), C.byref(gc)
for _ in range(3):
    m._chk(L.mgk_jacobi_f64(m.ctx, G, coef, dinv, 0.8, b, u, out, None))
    m._chk(L.mgk_jacobi2_2d_f64(m.ctx, G, coef, dinv, 0.8, b, u, out, None))
    m._chk(L.mgk_jacobi3_2d_f64(m.ctx, G, coef, dinv, 0.8, None, None, b, u, out, None))
    m._chk(L.mgk_jacobi3_2d_sumsq_f64(m.ctx, G, coef, dinv, 0.8, None, None, b, u, out, C.byref(ss), None))
    m._chk(L.mgk_jacobi3_2d_zero_f64(m.ctx, G, coef, dinv, 0.8, None, None, b, out, None))
    m._chk(L.mgk_prolong_jacobi3_2d_f64(m.ctx, G, GC, coef, dinv, 0.8, None, None, b, uc, u, out, None))
    m._chk(L.mgk_residual_restrict_2d_f64(m.ctx, G, GC, coef, b, u, bc, None, 0.0, 0.0, None))
    m._chk(L.mgk_prolong_jacobi_f64(m.ctx, G, GC, coef, dinv, 0.8, b, uc, u, out, None))
m.sync()
m.close()
