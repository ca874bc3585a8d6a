"""two fp64 sweeps in one pass at 511^3 / 255^3: LDS-ring kernel before / after the instruction diet"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.mgk import Mgk
m = Mgk(0); L = m.L
for n in (511, 255):
    g = m.geom(3, n)
    u, b, out = m.field(g), m.field(g), m.field(g)
    c = float((n + 1) ** 2); coef = m.coef([c, c, c, -6 * c, c, c, c]); dinv = -1.0 / (6 * c)
    t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double()
    def timeit(fn, reps=10):
        m._chk(fn()); m.sync(); best = 1e9
        for _ in range(reps):
            m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
            m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
        return best
    for var in (-1, 39, 2):
        L.mgk_set_tuning(var, -1)
        two = timeit(lambda: L.mgk_jacobi2_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
        z3 = timeit(lambda: L.mgk_jacobi2_zero_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, out, None)) if var != 2 else float("nan")
        print(f"n={n} variant {var}: two sweeps {two:.4f} ms ({24 * n ** 3 / two / 1e6:.0f} GB/s); three from zero {z3:.4f} ms", flush=True)
    L.mgk_set_tuning(-1, -1)
    for p in (u, b, out): m.free(p)
m.close()
