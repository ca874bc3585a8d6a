"""A/B on one box: plain sweep and sweep+norm, LDS-tile kernel (tuning variant 12 / 2) against the register / shuffle form (-1)"""
import ctypes as C, sys, os
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
coef = m.coef([c, c, c, -6 * c, c, c, c]); dinv = -1.0 / (6 * c)
t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double(); ss = C.c_double()
def timeit(fn, reps=6):
    fn(); m.sync(); best = 1e9
    for _ in range(reps):
        m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
        m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
    return best
N = float(n) ** 3
old = 12 if n >= 1023 else (6 if n >= 511 else 2)
for rnd in range(2):
    for v, name in ((old, "LDS tiles"), (-1, "row form")):
        L.mgk_set_tuning(v, -1)
        a = timeit(lambda: L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
        bn = timeit(lambda: L.mgk_jacobi_sumsq_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, C.byref(ss), None))
        print(f"n={n} {name:9s}: sweep {a:.3f} ms {24 * N / a / 1e6:.0f} GB/s | sweep+norm {bn:.3f} ms {24 * N / bn / 1e6:.0f} GB/s", flush=True)
m.close()
