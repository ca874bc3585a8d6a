#!/usr/bin/env python3
"""Why does a 4095^2 pass take longer inside the cycle than alone?  One sequence per process, rocprofv3 --kernel-trace around it:
   exp_incycle_2d.py alone|fine|cycle zero|rand
 alone: the three-sweep + norm pass 30 times;  alt: the same on two sets of fields in turn (nothing of a set survives in the 256 MB Infinity Cache);  fine: the three fine-level passes of a cycle in turn;  cycle: the same with the 2047^2 passes
 and 12 small launches in between (the shape of the real cycle)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.mgk import Mgk

mode, data = sys.argv[1], sys.argv[2]
m = Mgk(0); L = m.L
rng = np.random.default_rng(1)


def level(n):
    g = m.geom(2, n)
    f = []
    for _ in range(3):
        p = m.field(g)
        m._chk(L.mgk_memset0(m.ctx, p, 8 * g.total, None))
        f.append(p)
    if data == "rand":
        for p in f[:2]:
            tmp = m.to_field(g, rng.uniform(-1, 1, n * n))
            m._chk(L.mgk_d2d(m.ctx, p, tmp, 8 * g.total, None)); m.sync(); m.free(tmp)
    q = float((n + 1) ** 2)
    return g, f, m.coef([q, q, -4 * q, q, q]), -1.0 / (4 * q)


lv = [level(n) for n in (4095, 2047, 1023, 511)]
if mode == "alt":                       # a second set of fine-level fields: the two sets alternate, 800 MB in turn (beyond the Infinity Cache)
    lv.append(level(4095))
ss = C.c_double()


REVN = len(sys.argv) > 3 and sys.argv[3] == "rev"       # the norm pass marches its chunks downwards (tuning variant 59)


def j3n(k):
    g, (u, b, o), coef, dinv = lv[k]
    if REVN and k == 0:
        L.mgk_set_tuning(59, -1)
    m._chk(L.mgk_jacobi3_2d_sumsq_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, u, o, C.byref(ss), None))
    L.mgk_set_tuning(-1, -1)


def rr(k):
    g, (u, b, o), coef, dinv = lv[k]
    gc, (uc, bc, oc), _, dc = lv[k + 1]
    m._chk(L.mgk_residual_restrict_2d_f64(m.ctx, C.byref(g), C.byref(gc), coef, b, o, bc, None, 0.0, 0.0, None))


def pj(k):
    g, (u, b, o), coef, dinv = lv[k]
    gc, (uc, bc, oc), _, dc = lv[k + 1]
    m._chk(L.mgk_prolong_jacobi3_2d_f64(m.ctx, C.byref(g), C.byref(gc), coef, dinv, 0.8, None, None, b, uc, o, u, None))


for it in range(30):
    if mode == "alone":
        j3n(0)
    elif mode == "alt":
        j3n(0); j3n(4)
    elif mode == "chain":               # every launch reads the field its predecessor wrote (u -> o, o -> u)
        g, (u, b, o), coef, dinv = lv[0]
        m._chk(L.mgk_jacobi3_2d_sumsq_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, u, o, C.byref(ss), None))
        m._chk(L.mgk_jacobi3_2d_sumsq_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, o, u, C.byref(ss), None))
    elif mode == "chain_plain":         # the same without the norm (no host round trip between the launches)
        g, (u, b, o), coef, dinv = lv[0]
        m._chk(L.mgk_jacobi3_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, u, o, None))
        m._chk(L.mgk_jacobi3_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, o, u, None))
    elif mode == "chain_st":            # chain_plain with ordinary (not non-temporal) stores: tuning variant 60
        g, (u, b, o), coef, dinv = lv[0]
        L.mgk_set_tuning(60, -1)
        m._chk(L.mgk_jacobi3_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, u, o, None))
        m._chk(L.mgk_jacobi3_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, o, u, None))
        L.mgk_set_tuning(-1, -1)
    elif mode == "alone_st":
        g, (u, b, o), coef, dinv = lv[0]
        L.mgk_set_tuning(60, -1)
        m._chk(L.mgk_jacobi3_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, u, o, None))
        L.mgk_set_tuning(-1, -1)
    elif mode == "rot3":                # three fields in rotation: each launch reads what its predecessor wrote and writes the field read two launches ago
        g, (u, b, o), coef, dinv = lv[0]
        if it == 0:
            lv.append(level(4095))
        x = lv[4][1][0]
        f = [u, o, x]
        m._chk(L.mgk_jacobi3_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, f[it % 3], f[(it + 1) % 3], None))
    elif mode == "alone_plain":
        g, (u, b, o), coef, dinv = lv[0]
        m._chk(L.mgk_jacobi3_2d_f64(m.ctx, C.byref(g), coef, dinv, 0.8, None, None, b, u, o, None))
    elif mode == "fine":
        j3n(0); rr(0); pj(0)
    else:
        j3n(0); rr(0); j3n(1); rr(1)
        for _ in range(6):
            j3n(3)
        pj(1); pj(0)
m.sync()
