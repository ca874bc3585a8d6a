"""Triad probe over launch shapes (debugging aid for bench.py's measured_ceiling)."""
import ctypes as C, sys
sys.path.insert(0, ".")
from multigrid_petsc_amd.mgk import Mgk
m = Mgk(0); L = m.L
n = 1023 ** 3 & ~1
a, b, c = (m.alloc(8 * n) for _ in range(3))
L.mgk_flat_fill(m.ctx, n, 1.0, b, None); L.mgk_flat_fill(m.ctx, n, 2.0, c, None)
t = C.c_void_p(); L.mgk_timer_create(m.ctx, C.byref(t))
for blocks in (256, 512, 1024, 2048, 4096, 16384, 65536):
    for nt in (1, 0):
        L.mgk_stream_triad_f64(m.ctx, n, a, b, c, 0.5, blocks, nt, None)
        L.mgk_timer_start(m.ctx, t, None)
        for _ in range(5):
            L.mgk_stream_triad_f64(m.ctx, n, a, b, c, 0.5, blocks, nt, None)
        L.mgk_timer_stop(m.ctx, t, None)
        ms = C.c_double(); L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms))
        print(blocks, nt, "%.1f GB/s" % (24.0 * n * 5 / ms.value / 1e6), flush=True)
