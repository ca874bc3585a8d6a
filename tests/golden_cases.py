"""Loader of tests/golden/vcycle_golden.npz (made by tests/golden/make_golden.py: an independent
scipy.sparse restatement of the cycle, run in the development container)."""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vcycle_golden.npz")
GOLD = np.load(_PATH, allow_pickle=False)
CYCLE_KEYS = sorted(k[:-5] for k in GOLD.files if k.endswith("_meta") and not k.startswith("mesh"))
MESH_KEYS = sorted(k[:-5] for k in GOLD.files if k.endswith("_meta") and k.startswith("mesh"))


def cycle_case(key):
    dim, npts, levels, v0, v1, maxiter, iters = (int(x) for x in GOLD[key + "_meta"])
    scale, bnorm = (float(x) for x in GOLD[key + "_scale"])
    return dict(dim=dim, npts=npts, levels=levels, v0=v0, v1=v1, maxiter=maxiter, iters=iters, scale=scale,
                bnorm=bnorm, rnorm=GOLD[key + "_rnorm"], err=GOLD[key + "_err"], u=GOLD[key + "_u"])


def mesh_case(key):
    """stretched-mesh cycles (-mesh 1/2, 2-D): same fields as cycle_case plus `mesh`"""
    dim, npts, levels, v0, v1, maxiter, iters, mesh = (int(x) for x in GOLD[key + "_meta"])
    scale, bnorm = (float(x) for x in GOLD[key + "_scale"])
    return dict(dim=dim, npts=npts, levels=levels, v0=v0, v1=v1, maxiter=maxiter, iters=iters, scale=scale, mesh=mesh,
                bnorm=bnorm, rnorm=GOLD[key + "_rnorm"], err=GOLD[key + "_err"], u=GOLD[key + "_u"])
