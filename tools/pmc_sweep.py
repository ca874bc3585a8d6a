"""A handful of fine-level kernels at 1023^3 for the PMC (HBM traffic) passes of rocprofv3.
Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 tools/pmc_sweep.py
and again with --pmc WRITE_SIZE (the two do not fit one pass on gfx950: MI355X_MICROARCH.md, PMC slots)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
m = Mgk(0); L = m.L
g = m.geom(3, n)
rng = np.random.default_rng(0)
r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(3)]
u, b, out = m.field(g), m.field(g), m.field(g)
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[2], u, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[2], r1[0], r1[1], b, None))
h = 1.0 / (n + 1); c = 1.0 / (h * h)
As = [c] * 7; As[3] = -6 * c
coef = m.coef(As); dinv = 1.0 / As[3]
ss = C.c_double()
for _ in range(3):
    m._chk(L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, u, out, None))
    m._chk(L.mgk_residual_f64(m.ctx, C.byref(g), coef, b, u, out, None))
    m._chk(L.mgk_residual_sumsq_f64(m.ctx, C.byref(g), coef, b, u, C.byref(ss), None))
    m._chk(L.mgk_jacobi2_f64(m.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, u, out, None))
    m._chk(L.mgk_jacobi_sumsq_f64(m.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, u, out, C.byref(ss), None))
m.sync()
gc = m.geom(3, (n - 1) // 2)
uc, bc = m.field(gc), m.field(gc)
for _ in range(2):
    m._chk(L.mgk_restrict_fw_f64(m.ctx, C.byref(g), C.byref(gc), out, bc, None))
    m._chk(L.mgk_prolong_add_f64(m.ctx, C.byref(g), C.byref(gc), uc, u, None))
    m._chk(L.mgk_residual_restrict_f64(m.ctx, C.byref(g), C.byref(gc), coef, b, u, bc, None))
    m._chk(L.mgk_prolong_jacobi_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 6.0 / 7.0, b, uc, u, out, None))
    # round 2: last pre-smoothing sweep + residual + restriction (+ coarse first sweep) in one pass; two sweeps + norm
    m._chk(L.mgk_sweep_residual_restrict_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 6.0 / 7.0, b, u, out, bc, uc, dinv, 6.0 / 7.0, None))
    m._chk(L.mgk_jacobi2_sumsq_f64(m.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, u, out, C.byref(ss), None))
    # end of round 2 / round 3: the passes of the 91-byte fine level (prolongation + two sweeps, two sweeps + norm of the mid iterate, residual
    # + restriction + coarse zero-guess sweep) and the three-sweep prototype
    m._chk(L.mgk_prolong_jacobi2_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 6.0 / 7.0, b, uc, u, out, None))
    m._chk(L.mgk_jacobi2_sumsq_mid_f64(m.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, u, out, C.byref(ss), None))
    m._chk(L.mgk_residual_restrict_jz_f64(m.ctx, C.byref(g), C.byref(gc), coef, b, u, bc, uc, dinv, 6.0 / 7.0, None))
    m._chk(L.mgk_jacobi3_f64(m.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, u, out, None))
m.sync()
m.close()
