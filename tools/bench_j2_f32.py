"""micro-benchmark: two fp32 sweeps in one pass vs two plain fp32 sweeps -- tuning aid"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
zcs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
m = Mgk(0); L = m.L
g = m.geom32(n)
u, b, out = (m.alloc(4 * g.total) for _ in range(3))
for p in (u, b, out):
    m._chk(L.mgk_memset0(m.ctx, p, 4 * g.total, None))
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
one = timeit(lambda: L.mgk_jacobi_f32(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
print(f"n={n} plain fp32 sweep: {one:.3f} ms ({12 * N / one / 1e6:.0f} GB/s) -> two sweeps {2 * one:.3f} ms", flush=True)
NAMES = {-1: "default (ring, instruction diet)", 39: "ring before the diet", 37: "ring, predicated loads + ds_bpermute"}
for var, zc in [(v, z) for v in (-1, 39, 37) for z in zcs]:
    L.mgk_set_tuning(var, zc)
    two = timeit(lambda: L.mgk_jacobi2_f32(m.ctx, C.byref(g), coef, dinv, 0.85, b, u, out, None))
    print(f"n={n} variant={NAMES[var]} zc={zc}: two-in-one {two:.3f} ms  ({12 * N / two / 1e6:.0f} GB/s of the 12 B/unknown minimum; {two / (2 * one):.2f} x two plain sweeps)", flush=True)
L.mgk_set_tuning(-1, -1)
z3 = timeit(lambda: L.mgk_jacobi2_zero_f32(m.ctx, C.byref(g), coef, dinv, 0.85, b, out, None))
L.mgk_set_tuning(39, -1)
z3o = timeit(lambda: L.mgk_jacobi2_zero_f32(m.ctx, C.byref(g), coef, dinv, 0.85, b, out, None))
print(f"n={n} three sweeps from the zero guess: {z3:.3f} ms (before the diet {z3o:.3f})", flush=True)
L.mgk_set_tuning(-1, -1)
m.close()
