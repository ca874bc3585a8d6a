#!/usr/bin/env python3
"""The reference's UNMODIFIED driver over the drop-in (build/refdriver/poisson) with options drawn at random -- size, depth, -v, -mesh, -map,
damping -- against the oracle: iteration count equal, uData.dat bit-identical.  A one-off stress run (GPU box; needs the binary that
__graft_entry__.build() links where /root/reference exists).  MG_STRESS_EXE=<binary> runs another link of the same driver instead -- on the
CPU tests/_san/san_refdriver (the reference's objects + the drop-in's host C over tests/mock_mgk.cpp, sanitized; built by
tests/test_host_sanitized.py), with MG_STRESS_MAXN bounding the sizes the scalar mock has to sweep.  usage: stress_refdriver.py [count] [seed]"""
import os
import subprocess
import sys
import tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import Oracle

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
exe = os.environ.get("MG_STRESS_EXE") or os.path.join(ROOT, "build", "refdriver", "poisson")
maxn = int(os.environ.get("MG_STRESS_MAXN", "513"))
orc = Oracle()
bad = 0
for q in range(count):
    npts = int(rng.choice([n for n in (17, 33, 65, 129, 257, 513) if n <= maxn]))
    lmax = int(np.log2(npts - 1))
    levels = int(rng.integers(1, lmax + 1))
    v0, v1 = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    mesh = int(rng.choice([0, 0, 1, 2]))
    mp = int(rng.choice([0, 1, 2]))
    scale = float(rng.choice([0.8, 1.0, 0.6]))
    cyc = int(rng.choice([0, 0, 0, 8]))             # -cycle 8: the reference's PCMG set-up over the drop-in's PCMG (uniform mesh, >= 2 levels)
    if cyc == 8 and (mesh or levels < 2):
        cyc = 0
    env = dict(os.environ)
    for k, vals in (("MGPETSC_TAIL", ["1", "1", "0"]), ("MGPETSC_KEEP_R", ["1", "1", "0"]), ("MGPETSC_J3", ["1", "1", "0"]), ("MGPETSC_LAZY", ["1", "1", "1", "0"])):
        env[k] = str(rng.choice(vals))
    tag = f"cycle={cyc} npts={npts} levels={levels} v=({v0},{v1}) mesh={mesh} map={mp} scale={scale} TAIL={env['MGPETSC_TAIL']} KEEP_R={env['MGPETSC_KEEP_R']} J3={env['MGPETSC_J3']} LAZY={env['MGPETSC_LAZY']}"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "poisson.in"), "w").write(
            f"-npts {npts}\n-mesh {mesh}\n-iter 300\n-grids {levels}\n-levels {levels}\n-cycle {cyc}\n-map {mp}\n-v {v0},{v1}\n-moreNorm 0\n" +
            (f"-pc_type jacobi\n-ksp_richardson_scale {scale!r}\n" if cyc == 0 else
             f"-mg_levels_ksp_type richardson\n-mg_levels_pc_type jacobi\n-mg_levels_ksp_max_it {v0}\n-mg_levels_ksp_richardson_scale {scale!r}\n"
             f"-mg_coarse_ksp_type richardson\n-mg_coarse_pc_type jacobi\n-mg_coarse_ksp_max_it {v1}\n-mg_coarse_ksp_richardson_scale {scale!r}\n"))
        p = subprocess.run([exe], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        if p.returncode != 0:
            bad += 1
            print("FAILED", tag, p.stdout[-400:], flush=True)
            continue
        it = int([ln for ln in p.stdout.splitlines() if "Number of iterations" in ln][0].split()[-1])
        u = np.array(open(os.path.join(d, "uData.dat")).read().split(), dtype=np.float64)
    ref = (orc.pcmg(2, npts, levels, v0, v1, maxiter=300, scale=scale) if cyc == 8 else
           orc.vcycle(2, npts, levels, v0, v1, maxiter=300, scale=scale, use_csr=1 if mesh else 0, mesh=mesh))
    if it != ref["iters"] or not np.array_equal(u, ref["u"]):
        bad += 1
        print("MISMATCH", tag, "iters", it, ref["iters"], "max|du|", float(np.max(np.abs(u - ref["u"]))), flush=True)
print(f"{count} configurations, {bad} mismatches")
sys.exit(1 if bad else 0)
