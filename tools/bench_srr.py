"""micro-benchmark of the fused sweep + residual + restriction (mgk_sweep_residual_restrict_f64) against the two passes it
replaces, at n^3 (default 1023) -- tuning aid.  Usage: python tools/bench_srr.py [n]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
m = Mgk(0); L = m.L
g = m.geom(3, n); gc = m.geom(3, (n - 1) // 2)
rng = np.random.default_rng(0)
r1 = [m.upload(rng.uniform(-1, 1, n)) for _ in range(3)]
u, b, out, bc, uc0 = m.field(g), m.field(g), m.field(g), m.field(gc), m.field(gc)
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[0], r1[1], r1[2], u, None))
m._chk(L.mgk_fill_separable_f64(m.ctx, C.byref(g), r1[2], r1[0], r1[1], b, None))
h = 1.0 / (n + 1); c = 1.0 / (h * h)
coef = m.coef([c, c, c, -6 * c, c, c, c]); dinv = -1.0 / (6 * c)
t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double()
def timeit(fn, reps=4):
    m._chk(fn()); m.sync(); best = 1e9
    for _ in range(reps):
        m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
        m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
    return best
N = float(n) ** 3
sw = timeit(lambda: L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
rr = timeit(lambda: L.mgk_residual_restrict_jz_f64(m.ctx, C.byref(g), C.byref(gc), coef, b, u, bc, uc0, dinv, 0.85, None))
p2 = timeit(lambda: L.mgk_jacobi2_f64(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
print(f"n={n}: sweep {sw:.3f} ms, residual+restriction(+jz) {rr:.3f} ms, sum {sw + rr:.3f} ms; two-sweep pass {p2:.3f} ms", flush=True)
for v in [int(x) for x in os.environ.get('SRR_VARIANTS', '-1,40').split(',')]:
    for zc in [int(x) for x in os.environ.get("SRR_ZC", "-1,256,128,64").split(",")]:
        L.mgk_set_tuning(v, zc)
        s = timeit(lambda: L.mgk_sweep_residual_restrict_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.85, b, u, out, bc, uc0, dinv, 0.85, None))
        print(f"fused sweep+residual+restriction, variant {v}, coarse planes per chunk {zc:4d}: {s:7.3f} ms  {26 * N / s / 1e6:7.1f} GB/s", flush=True)
L.mgk_set_tuning(-1, -1)
m.close()
