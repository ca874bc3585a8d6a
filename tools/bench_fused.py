"""micro-benchmark of the fused kernels (prolong+sweep, residual+restrict) at 1023^3 -- tuning aid"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd import _lib
if os.environ.get("MGK_LIB"):
    _lib._cache["mgk"] = C.CDLL(os.environ["MGK_LIB"], mode=C.RTLD_GLOBAL)
from multigrid_petsc_amd.mgk import Mgk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
m = Mgk(0); L = m.L
g = m.geom(3, n); gc = m.geom(3, (n - 1) // 2)
rng = np.random.default_rng(0)
r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(3)]
u, b, out, uc, bc = m.field(g), m.field(g), m.field(g), m.field(gc), m.field(gc)
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[2], u, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[2], r1[0], r1[1], b, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(gc), r1[1], r1[2], r1[0], uc, None))
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
for v in [int(x) for x in os.environ.get("PJ_VARIANTS", "9,31,32").split(",")]:
    for zc in (-1, 512):
        L.mgk_set_tuning(v, zc)
        pj = timeit(lambda: L.mgk_prolong_jacobi_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.85, b, uc, u, out, None))
        print(f"PJ variant {v:2d} zc {zc:4d}: {pj:7.3f} ms  {25 * N / pj / 1e6:7.1f} GB/s", flush=True)
for v in [int(x) for x in os.environ.get("RR_VARIANTS", "30,31,32").split(",")]:
    for zc in (-1, 256):
        L.mgk_set_tuning(v, zc)
        rr = timeit(lambda: L.mgk_residual_restrict_f64(m.ctx, C.byref(g), C.byref(gc), coef, b, u, bc, None))
        print(f"RR variant {v:2d} coarse-planes-per-chunk {zc:4d}: {rr:7.3f} ms  {17 * N / rr / 1e6:7.1f} GB/s", flush=True)
m.close()
