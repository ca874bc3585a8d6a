"""Does the relative placement of u / b / out in HBM matter for the sweep?  (tuning experiment)"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk
n = 1023
m = Mgk(0); L = m.L
g = m.geom(3, n)
fb = 8 * g.total
big = m.alloc(3 * fb + (256 << 20))
base = big.value
rng = np.random.default_rng(0)
r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(3)]
h = 1.0 / (n + 1); c = 1.0 / (h * h)
coef = m.coef([c, c, c, -6 * c, c, c, c]); dinv = -1.0 / (6 * c)
t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double()
def timeit(fn, reps=3):
    m._chk(fn()); m.sync(); best = 1e9
    for _ in range(reps):
        m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
        m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
    return best
def rnd(x, a): return (x + a - 1) // a * a
print("field bytes", fb, "mod 2MiB", fb % (2 << 20), "mod 4KiB", fb % 4096)
for d1, d2 in [(0, 0), (4096, 8192), (1 << 20, 2 << 20), (65536, 131072), (512, 1024), (2048, 4096 + 2048), (8192 + 256, 16384 + 512),
               ((2 << 20) - fb % (2 << 20), (2 << 20) - fb % (2 << 20)), (3 * 4096, 7 * 4096), (256, 512), (128 * 33, 128 * 77)]:
    u = C.c_void_p(base)
    b = C.c_void_p(rnd(base + fb, 256) + d1)
    o = C.c_void_p(rnd(b.value + fb, 256) + d2)
    m._chk(L.mgk_memset0(m.ctx, big, 3 * fb + (128 << 20), None))
    m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[2], u, None))
    m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[2], r1[0], r1[1], b, None))
    f = timeit(lambda: L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, o, None))
    r = timeit(lambda: L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, o, u, None))
    print(f"d1={d1:9d} d2={d2:9d}  (b-u)%2MiB={(b.value-u.value)%(2<<20):8d} (o-u)%2MiB={(o.value-u.value)%(2<<20):8d}  u->o {f:6.3f} ms  o->u {r:6.3f} ms", flush=True)
m.close()
