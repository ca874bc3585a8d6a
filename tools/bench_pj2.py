#!/usr/bin/env python3
"""HIP-event timing of the prolongation + two-sweep pass (mgk_prolong_jacobi2_f64) and its neighbours at 511^3 / 1023^3, tuning variants.
usage: bench_pj2.py n [variant ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk

m = Mgk(0); L = m.L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 511
variants = [int(v) for v in sys.argv[2:]] or [-1, 46, 47]
g, gc = m.geom(3, n), m.geom(3, (n - 1) // 2)
u, b, o, uc, bc = m.field(g), m.field(g), m.field(g), m.field(gc), m.field(gc)
q = float((n + 1) ** 2)
coef, dinv = m.coef([q, q, q, -6 * q, q, q, q]), -1.0 / (6 * q)
G, GC = C.byref(g), C.byref(gc)
ss = C.c_double()


def timeit(fn, reps=8):
    t = C.c_void_p()
    m._chk(L.mgk_timer_create(m.ctx, C.byref(t)))
    m._chk(fn())
    m._chk(L.mgk_timer_start(m.ctx, t, None))
    for _ in range(reps):
        m._chk(fn())
    m._chk(L.mgk_timer_stop(m.ctx, t, None))
    ms = C.c_double()
    m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms)))
    L.mgk_timer_destroy(m.ctx, t)
    return ms.value / reps


N = float(n) ** 3
for var in variants:
    L.mgk_set_tuning(var, -1)
    for name, byts, fn in (
            ("prolong + two sweeps", 25, lambda: L.mgk_prolong_jacobi2_f64(m.ctx, G, GC, coef, dinv, 0.8, b, uc, u, o, None)),
            ("prolong + sweep", 25, lambda: L.mgk_prolong_jacobi_f64(m.ctx, G, GC, coef, dinv, 0.8, b, uc, u, o, None)),
            ("three sweeps from zero", 16, lambda: L.mgk_jacobi2_zero_f64(m.ctx, G, coef, dinv, 0.8, b, o, None)),
            ("two sweeps", 24, lambda: L.mgk_jacobi2_f64(m.ctx, G, coef, dinv, 0.8, b, u, o, None)),
            ("residual + restriction + jz", 18, lambda: L.mgk_residual_restrict_jz_f64(m.ctx, G, GC, coef, b, u, bc, uc, dinv, 0.8, None))):
        ms = timeit(fn)
        print(f"n={n} variant={var:3d} {name:28s} {ms:7.3f} ms  {byts * N / ms / 1e9:6.2f} TB/s of its {byts} B", flush=True)
L.mgk_set_tuning(-1, -1)
