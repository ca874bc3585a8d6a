#!/usr/bin/env python3
"""HIP-event timings of the 3-D three-sweep pass beside the sweep and the two-sweep pass.  usage: bench_j3_3d.py [n ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk

m = Mgk(0)
L = m.L


def timeit(fn, reps=5):
    t = C.c_void_p()
    m._chk(L.mgk_timer_create(m.ctx, C.byref(t)))
    fn()
    m._chk(L.mgk_timer_start(m.ctx, t, None))
    for _ in range(reps):
        fn()
    m._chk(L.mgk_timer_stop(m.ctx, t, None))
    ms = C.c_double()
    m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms)))
    L.mgk_timer_destroy(m.ctx, t)
    return ms.value / reps


for n in [int(x) for x in sys.argv[1:]] or [1023, 511]:
    g = m.geom(3, n)
    u, b, o = m.field(g), m.field(g), m.field(g)
    for f in (u, b, o):
        m._chk(L.mgk_memset0(m.ctx, f, 8 * g.total, None))
    q = float((n + 1) ** 2)
    coef, dinv = m.coef([q, q, q, -6 * q, q, q, q]), -1.0 / (6 * q)
    G = C.byref(g)
    N = float(n) ** 3
    print(f"n={n}: sweep {timeit(lambda: m._chk(L.mgk_jacobi_f64(m.ctx, G, coef, dinv, 0.8, b, u, o, None))):.3f} ms, "
          f"two sweeps {timeit(lambda: m._chk(L.mgk_jacobi2_f64(m.ctx, G, coef, dinv, 0.8, b, u, o, None))):.3f} ms", flush=True)
    for var, zc in ((-1, -1), (64, -1), (-1, 256), (64, 256), (63, 256), (62, 256), (-1, 64), (64, 64)):
        L.mgk_set_tuning(var, zc)
        ms = timeit(lambda: m._chk(L.mgk_jacobi3_f64(m.ctx, G, coef, dinv, 0.8, b, u, o, None)))
        print(f"n={n} var={var} zc={zc}: THREE sweeps {ms:.3f} ms = {24 * N / ms / 1e9:.2f} TB/s of its 24 B", flush=True)
    L.mgk_set_tuning(-1, -1)
    for f in (u, b, o):
        m.free(f)
