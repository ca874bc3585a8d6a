import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.mgk import Mgk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
m = Mgk(0); L = m.L
g = m.geom(3, n)
rng = np.random.default_rng(0)
r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(3)]
u, b, pm, out = m.field(g), m.field(g), m.field(g), m.field(g)
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[2], u, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[2], r1[0], r1[1], b, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[1], r1[2], r1[0], pm, None))
h = 1.0 / (n + 1); c = 1.0 / (h * h)
coef = m.coef([c, c, c, -6 * c, c, c, c]); dinv = -1.0 / (6 * c)
t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double()
def timeit(fn, reps=4):
    fn(); m.sync(); best = 1e9
    for _ in range(reps):
        m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
        m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
    return best
N = float(n) ** 3
for v in (-1, 12, 9, 6, 3, 2, 13):
    L.mgk_set_tuning(v, -1)
    a = timeit(lambda: L.mgk_cheby_f64(m.ctx, C.byref(g), coef, dinv, -0.3, 1.3, 0.4, b, u, pm, out, None))
    print(f"n={n} cheby variant {v:3d}: {a:.3f} ms {32 * N / a / 1e6:.0f} GB/s", flush=True)
m.close()
