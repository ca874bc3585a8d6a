"""micro-benchmark: two sweeps in one pass (mgk_jacobi2_f64) vs two plain sweeps -- tuning aid"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
zcs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
m = Mgk(0); L = m.L
g = m.geom(3, n)
rng = np.random.default_rng(0)
r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(3)]
u, b, out = m.field(g), m.field(g), m.field(g)
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[2], u, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[2], r1[0], r1[1], b, None))
h = 1.0 / (n + 1); c = 1.0 / (h * h)
coef = m.coef([c, c, c, -6 * c, c, c, c]); dinv = -1.0 / (6 * c)
t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double()
def timeit(fn, reps=5):
    fn(); m.sync(); best = 1e9
    for _ in range(reps):
        m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
        m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
    return best
N = float(n) ** 3
one = timeit(lambda: L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
print(f"n={n} plain sweep: {one:.3f} ms ({24 * N / one / 1e6:.0f} GB/s) -> two sweeps {2 * one:.3f} ms", flush=True)
for var, zc in [(v, z) for v in (2, 36, 1, 37) for z in zcs]:
    L.mgk_set_tuning(var, zc)
    two = timeit(lambda: L.mgk_jacobi2_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
    print(f"n={n} variant={ {2: 'one-barrier', 36: 'one-barrier, predicated loads + ds_bpermute', 1: 'ring', 37: 'ring, predicated loads + ds_bpermute'}[var]} zc={zc}: two-in-one {two:.3f} ms  ({24 * N / two / 1e6:.0f} GB/s of the 24 B/unknown minimum; {two / (2 * one):.2f} x the two plain sweeps)", flush=True)
m.close()
