"""experiment: the prolongation fused into a two-sweep pass at 1023^3 against its two-pass form"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multigrid_petsc_amd.mgk import Mgk
n = 1023
m = Mgk(0); L = m.L
g = m.geom(3, n); gc = m.geom(3, (n - 1) // 2)
rng = np.random.default_rng(0)
r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(3)]
u, b, out, uc = m.field(g), m.field(g), m.field(g), m.field(gc)
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[2], u, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[2], r1[0], r1[1], b, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(gc), r1[1], r1[2], r1[0], uc, None))
c = float((n + 1) ** 2); coef = m.coef([c, c, c, -6 * c, c, c, c]); dinv = -1.0 / (6 * c)
t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double()
def timeit(fn, reps=4):
    m._chk(fn()); m.sync(); best = 1e9
    for _ in range(reps):
        m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
        m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
    return best
N = float(n) ** 3
pj = timeit(lambda: L.mgk_prolong_jacobi_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.85, b, uc, u, out, None))
sw = timeit(lambda: L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
p2 = timeit(lambda: L.mgk_jacobi2_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
print(f"prolongation sweep {pj:.3f} ms, sweep {sw:.3f} ms, two-sweep pass {p2:.3f} ms", flush=True)
for var, zc in ((-1, -1), (46, -1), (-1, 512), (46, 512)):
    L.mgk_set_tuning(var, zc)
    x = timeit(lambda: L.mgk_prolong_jacobi2_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.85, b, uc, u, out, None))
    print(f"prolongation + two sweeps, variant {var}, planes per chunk {zc}: {x:.3f} ms ({25 * N / x / 1e6:.0f} GB/s)", flush=True)
m.close()
