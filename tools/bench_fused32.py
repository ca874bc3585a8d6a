"""micro-benchmark of the fp32 fused kernels (prolong+sweep, residual+restrict, plain sweep) at 1023^3 -- tuning aid"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
m = Mgk(0); L = m.L
g = m.geom32(n); gc = m.geom32((n - 1) // 2)
u, b, out, uc, bc = (m.alloc(4 * g.total) for _ in range(3)).__iter__(), None, None, None, None
u = m.alloc(4 * g.total); b = m.alloc(4 * g.total); out = m.alloc(4 * g.total); uc = m.alloc(4 * gc.total); bc = m.alloc(4 * gc.total)
for p, gg in ((u, g), (b, g), (out, g), (uc, gc), (bc, gc)):
    m._chk(L.mgk_memset0(m.ctx, p, 4 * gg.total, None))
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
for v in [int(x) for x in os.environ.get("PJ_VARIANTS", "-1,34").split(",")]:
    for zc in (-1,):
        L.mgk_set_tuning(v, zc)
        sw = timeit(lambda: L.mgk_jacobi_f32(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
        pj = timeit(lambda: L.mgk_prolong_jacobi_f32(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.85, b, uc, u, out, None))
        print(f"variant {v} zc {zc:4d}: sweep {sw:6.3f} ms {12 * N / sw / 1e6:7.1f} GB/s | PJ {pj:6.3f} ms {12.5 * N / pj / 1e6:7.1f} GB/s", flush=True)
for v in [int(x) for x in os.environ.get("RR_VARIANTS", "-1,34").split(",")]:
    for zc in (-1,):
        L.mgk_set_tuning(v, zc)
        rr = timeit(lambda: L.mgk_residual_restrict_f32(m.ctx, C.byref(g), C.byref(gc), coef, b, u, bc, None))
        print(f"RR32 variant {v:2d} coarse-planes-per-chunk {zc:4d}: {rr:6.3f} ms  {8.5 * N / rr / 1e6:7.1f} GB/s", flush=True)
m.close()
