"""micro-benchmark: 2-D two-sweep pass vs two plain 2-D sweeps -- tuning aid"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk
m = Mgk(0); L = m.L
t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double()
def timeit(fn, reps=20):
    fn(); m.sync(); best = 1e9
    for _ in range(reps):
        m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
        m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
    return best
for n in (4095, 2047, 1023, 511):
    g = m.geom(2, n)
    u, b, out = m.field(g), m.field(g), m.field(g)
    for f in (u, b, out):
        m._chk(L.mgk_memset0(m.ctx, f, 8 * g.total, None))
    q = float((n + 1) ** 2); coef = m.coef([q, q, -4 * q, q, q]); dinv = -1.0 / (4 * q)
    one = timeit(lambda: L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.8, b, u, out, None))
    for zc in (-1, 16, 32, 64, 128):
        L.mgk_set_tuning(-1, zc)
        two = timeit(lambda: L.mgk_jacobi2_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, b, u, out, None))
        print(f"n={n} zc={zc}: one sweep {one * 1e3:.1f} us, two-in-one {two * 1e3:.1f} us ({two / (2 * one):.2f} x two sweeps)", flush=True)
    L.mgk_set_tuning(-1, -1)
    for f in (u, b, out):
        m.free(f)
m.close()
