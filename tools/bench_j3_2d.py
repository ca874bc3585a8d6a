#!/usr/bin/env python3
"""HIP-event timings of the 2-D three-sweep passes beside the kernels they replace.  usage: bench_j3_2d.py [n ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk

m = Mgk(0)
L = m.L


def timeit(fn, reps=20):
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
    return 1e3 * ms.value / reps


for n in [int(x) for x in sys.argv[1:]] or [4095, 2047, 1023, 511, 255, 127]:
    g, gc = m.geom(2, n), m.geom(2, (n - 1) // 2)
    u, b, o, uc = m.field(g), m.field(g), m.field(g), m.field(gc)
    for f, gg in ((u, g), (b, g), (o, g), (uc, gc)):
        m._chk(L.mgk_memset0(m.ctx, f, 8 * gg.total, None))
    q = float((n + 1) ** 2)
    coef, dinv = m.coef([q, q, -4 * q, q, q]), -1.0 / (4 * q)
    ss = C.c_double()
    G, GC = C.byref(g), C.byref(gc)
    rows = [
        ("sweep", 24, lambda: L.mgk_jacobi_f64(m.ctx, G, coef, dinv, 0.8, b, u, o, None)),
        ("two sweeps", 24, lambda: L.mgk_jacobi2_2d_f64(m.ctx, G, coef, dinv, 0.8, b, u, o, None)),
        ("THREE sweeps", 24, lambda: L.mgk_jacobi3_2d_f64(m.ctx, G, coef, dinv, 0.8, None, None, b, u, o, None)),
        ("THREE sweeps + norm", 24, lambda: L.mgk_jacobi3_2d_sumsq_f64(m.ctx, G, coef, dinv, 0.8, None, None, b, u, o, C.byref(ss), None)),
        ("THREE from zero", 16, lambda: L.mgk_jacobi3_2d_zero_f64(m.ctx, G, coef, dinv, 0.8, None, None, b, o, None)),
        ("prolong + sweep", 25, lambda: L.mgk_prolong_jacobi_f64(m.ctx, G, GC, coef, dinv, 0.8, b, uc, u, o, None)),
        ("prolong + THREE", 25, lambda: L.mgk_prolong_jacobi3_2d_f64(m.ctx, G, GC, coef, dinv, 0.8, None, None, b, uc, u, o, None)),
        ("residual+restrict THREE", 18, lambda: L.mgk_residual_restrict_2d_f64(m.ctx, G, GC, coef, b, u, uc, None, 0.0, 0.0, None)),
    ]
    # MG_J3_TUNE="var,zc;var,zc" replaces the list of (tuning variant, rows per chunk) pairs
    tune = [tuple(int(t) for t in q.split(",")) for q in os.environ["MG_J3_TUNE"].split(";")] if os.environ.get("MG_J3_TUNE") else [(-1, -1), (57, -1), (58, -1)]
    for var, zc in tune:
        L.mgk_set_tuning(var, zc)
        for name, byts, fn in rows:
            if (var, zc) != (-1, -1) and "THREE" not in name:
                continue
            if "norm" in name:
                continue                 # (the host round trip of the sum dominates the small levels; the cycle defers it)
            us = timeit(lambda: m._chk(fn()))
            print(f"n={n:5d} var={var:3d} zc={zc:3d} {name:22s} {us:8.1f} us  {byts * n * n / us / 1e6:7.2f} TB/s of its {byts} B")
    L.mgk_set_tuning(-1, -1)
    for f in (u, b, o, uc):
        m.free(f)
