"""Micro-benchmark of the marching stencil kernel (variants x z-chunk) -- tuning aid, not the bench contract."""
import ctypes as C
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd import _lib
if os.environ.get("MGK_LIB"):
    _lib._cache["mgk"] = C.CDLL(os.environ["MGK_LIB"], mode=C.RTLD_GLOBAL)
from multigrid_petsc_amd.mgk import Mgk

def main():
    dim = int(sys.argv[1]); n = int(sys.argv[2])
    variants = [int(v) for v in sys.argv[3].split(",")]
    zcs = [int(v) for v in sys.argv[4].split(",")]
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    m = Mgk(0); L = m.L
    g = m.geom(dim, n)
    rng = np.random.default_rng(0)
    r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(3)]
    u, b, out = m.field(g), m.field(g), m.field(g)
    m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[2], u, None))
    m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[2], r1[0], r1[1], b, None))
    h = 1.0 / (n + 1); c = 1.0 / (h * h)
    As = [c] * 7; As[3] = -6 * c
    if dim == 2: As = [c, c, -4 * c, c, c]
    coef = m.coef(As); dinv = 1.0 / As[3 if dim == 3 else 2]
    N = float(n) ** dim
    t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t)))
    ms = C.c_double()
    for v in variants:
        for zc in zcs:
            L.mgk_set_tuning(v, zc)
            for mode in ("jacobi",):
                m._chk(L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.8, b, u, out, None)); m.sync()
                best = 1e9; tot = 0
                for _ in range(reps):
                    m._chk(L.mgk_timer_start(m.ctx, t, None))
                    m._chk(L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.8, b, u, out, None))
                    m._chk(L.mgk_timer_stop(m.ctx, t, None))
                    m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms)))
                    best = min(best, ms.value); tot += ms.value
                print(f"dim={dim} n={n} variant={v} zc={zc:4d} {mode}: best {best:8.3f} ms avg {tot/reps:8.3f} ms  "
                      f"{24*N/best/1e6:8.1f} GB/s best  {24*N/(tot/reps)/1e6:8.1f} GB/s avg", flush=True)
    L.mgk_set_tuning(-1, -1)
    ss = C.c_double()
    for name, fn, byts in (("jacobi_zero(read b, write out)", lambda: L.mgk_jacobi_zero_f64(m.ctx, C.byref(g), dinv, 0.8, b, out, None), 16),
                           ("sumsq(read only)", lambda: L.mgk_sumsq_f64(m.ctx, C.byref(g), u, C.byref(ss), None), 8),
                           ("residual_sumsq(read u,b)", lambda: L.mgk_residual_sumsq_f64(m.ctx, C.byref(g), coef, b, u, C.byref(ss), None), 16),
                           ("residual(read u,b write r)", lambda: L.mgk_residual_f64(m.ctx, C.byref(g), coef, b, u, out, None), 24)):
        best = 1e9
        for _ in range(reps):
            m._chk(L.mgk_timer_start(m.ctx, t, None))
            m._chk(fn())
            m._chk(L.mgk_timer_stop(m.ctx, t, None))
            m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms)))
            best = min(best, ms.value)
        print(f"{name}: best {best:.3f} ms  {byts*N/best/1e6:.1f} GB/s algorithmic", flush=True)
    # streaming references with the same buffers: copy (d2d) and memset
    nbytes = 8 * g.total
    for name in ("d2d", ):
        best = 1e9
        for _ in range(reps):
            m._chk(L.mgk_timer_start(m.ctx, t, None))
            m._chk(L.mgk_d2d(m.ctx, out, u, nbytes, None))
            m._chk(L.mgk_timer_stop(m.ctx, t, None))
            m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms)))
            best = min(best, ms.value)
        print(f"{name} copy of one field ({nbytes/1e9:.2f} GB): best {best:.3f} ms  {2*nbytes/best/1e6:.1f} GB/s (read+write)")
    m.close()

main()
