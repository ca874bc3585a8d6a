"""micro-benchmark of the 2-D fine-level passes at n^2 (default 4095) -- tuning aid.  Usage: python tools/bench_2d.py [n]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4095
m = Mgk(0); L = m.L
g = m.geom(2, n); gc = m.geom(2, (n - 1) // 2)
rng = np.random.default_rng(0)
u, b, out, bc, uc0 = [m.to_field(g, rng.uniform(-1, 1, n * n)) for _ in range(2)] + [m.field(g), m.field(gc), m.field(gc)]
c = float((n + 1) ** 2)
coef = m.coef([c, c, -4 * c, c, c]); dinv = -1.0 / (4 * c)
t = C.c_void_p(); m._chk(L.mgk_timer_create(m.ctx, C.byref(t))); ms = C.c_double(); ss = C.c_double()
def timeit(fn, reps=20):
    m._chk(fn()); m.sync(); best = 1e9
    for _ in range(reps):
        m._chk(L.mgk_timer_start(m.ctx, t, None)); m._chk(fn()); m._chk(L.mgk_timer_stop(m.ctx, t, None))
        m._chk(L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))); best = min(best, ms.value)
    return best * 1e3
N = float(n) ** 2
zcs = [int(x) for x in os.environ.get("ZC", "-1").split(",")]
for zc in zcs:
    L.mgk_set_tuning(-1, zc)
    r = {
        "sweep": (24, timeit(lambda: L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 0.8, b, u, out, None))),
        "sweep+norm": (24, timeit(lambda: L.mgk_jacobi_sumsq_f64(m.ctx, C.byref(g), coef, dinv, 0.8, b, u, out, C.byref(ss), None))),
        "two sweeps": (24, timeit(lambda: L.mgk_jacobi2_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, b, u, out, None))),
        "two sweeps+norm": (24, timeit(lambda: L.mgk_jacobi2_2d_sumsq_f64(m.ctx, C.byref(g), coef, dinv, 0.8, b, u, out, C.byref(ss), None))),
        "residual+restriction+jz": (18, timeit(lambda: L.mgk_residual_restrict_2d_f64(m.ctx, C.byref(g), C.byref(gc), coef, b, u, bc, uc0, dinv, 0.8, None))),
        "sweep+residual+restriction+jz": (26, timeit(lambda: L.mgk_sweep_residual_restrict_2d_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.8, b, u, out, bc, uc0, dinv, 0.8, None))),
        "prolongation+sweep": (25, timeit(lambda: L.mgk_prolong_jacobi_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.8, b, bc, u, out, None))),
    }
    print(f"n={n} chunk={zc}: " + "; ".join(f"{k} {v[1]:.1f} us ({v[0] * N / v[1] / 1e3:.0f} GB/s)" for k, v in r.items()), flush=True)
L.mgk_set_tuning(-1, -1)
m.close()
