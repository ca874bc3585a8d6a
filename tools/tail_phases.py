#!/usr/bin/env python3
"""Where the LDS tail kernel spends its time: (s_memrealtime, s_memtime) stamps at each of its barriers (mgk_debug_tail_stamps).
usage: tail_phases.py dim [fp64|fp32]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = Mgk(0); L = m.L
n0 = L.mgk_tail_max_n(dim)
ns = []
n = n0
while n >= 1:
    ns.append(n); n = (n - 1) // 2
g = m.geom(dim, n0)
b = m.to_field(g, np.random.default_rng(0).uniform(-1, 1, n0 ** dim))
u = m.field(g)
coef, dinv = [], []
for n in ns:
    q = float((n + 1) ** 2)
    coef += ([q, q, q, -6 * q, q, q, q] if dim == 3 else [q, q, -4 * q, q, q, 0, 0]); dinv.append(-1.0 / (2 * dim * q))
stamps = m.alloc(8 * 257)
L.mgk_debug_tail_stamps(stamps)
na = (C.c_int * len(ns))(*ns)
for rep in range(3):
    m._chk(L.mgk_tail_cycle_f64(m.ctx, C.byref(g), len(ns), na, m.coef(coef), m.coef(dinv), 0.8, 3, 3, b, u, None))
m.sync()
L.mgk_debug_tail_stamps(None)
raw = np.empty(257, dtype=np.int64)
m._chk(L.mgk_d2h(m.ctx, raw.ctypes.data_as(C.c_void_p), stamps, raw.nbytes))
k = int(raw[256])
wall, clk = raw[0:2 * k:2], raw[1:2 * k:2]
print(f"levels {ns}: {k} stamps, total {(wall[-1] - wall[0]) / 100:.2f} us, shader clock {(clk[-1] - clk[0]) / ((wall[-1] - wall[0]) / 100) / 1e3:.2f} GHz")
for i in range(1, k):
    print(f"  phase {i:3d}: {(wall[i] - wall[i - 1]) / 100:6.2f} us  {clk[i] - clk[i - 1]:7d} clk")
