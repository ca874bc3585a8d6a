#!/usr/bin/env python3
"""tools/stress_solver.py on the CPU: the PRODUCT's host logic (csrc/mg_solver.c, unchanged) over tests/mock_mgk.cpp (the kernel ABI in host
memory, canonical arithmetic) with configurations drawn at random against the oracle -- sweep counts, depth, damping, fuse bits, pair
threshold, recording on / off, precision.  What it can find is what the host decides: which pass runs when, buffer swaps, the coarse-level
recording, deferred sweeps (the odd-swap hole of the recording, round 3, shows here exactly as on the GPU).  Small sizes only: the mock is
scalar host code.  Test infrastructure: the mock library is injected into the package's loader cache HERE; the product has no such switch.
usage: stress_solver_mock.py [count] [seed]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_mock():
    out = os.path.join(ROOT, "tests", "_san")
    os.makedirs(out, exist_ok=True)
    inc = "-I" + os.path.join(ROOT, "include")
    objs = []
    for cc, std, src, obj in (("g++", "-std=c++17", os.path.join(ROOT, "tests", "mock_mgk.cpp"), "stress_mock_mgk.o"),
                              ("gcc", "-std=c99", os.path.join(ROOT, "multigrid_petsc_amd", "csrc", "mg_solver.c"), "stress_mg_solver.o"),
                              ("gcc", "-std=c99", os.path.join(ROOT, "multigrid_petsc_amd", "csrc", "mg_comm.c"), "stress_mg_comm.o")):
        o = os.path.join(out, obj)
        subprocess.run([cc, std, "-O2", "-fPIC", "-ffp-contract=off", "-D_POSIX_C_SOURCE=200809L", inc, "-c", src, "-o", o], check=True)
        objs.append(o)
    so = os.path.join(out, "libmgsolve_stress.so")
    subprocess.run(["g++", "-shared", "-Wl,-Bsymbolic", "-o", so] + objs + ["-lm", "-lpthread", "-ldl"], check=True)
    return so


def inject():
    """build the mock-backed library and make the package's loader hand it out (before multigrid_petsc_amd.solver / .comm are imported)"""
    import multigrid_petsc_amd._lib as loader
    lib = ctypes.CDLL(build_mock(), mode=ctypes.RTLD_GLOBAL)
    loader._cache["mgk"] = lib
    loader._cache["mgpetsc"] = lib


def main(mock=True):
    """mock=False: the same draws on the GPU over the real libraries, with the larger sizes (tools/stress_solver.py)"""
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    if mock:
        inject()
    from multigrid_petsc_amd.solver import Solver
    from oracle import Oracle
    orc = Oracle()
    bad = 0
    for q in range(count):
        dim = int(rng.choice([2, 3, 3]))
        if mock:
            npts = int(rng.choice([9, 17, 33, 65, 129] if dim == 2 else [9, 17, 33]))
        else:
            npts = int(rng.choice([9, 17, 33, 65, 129, 257, 513, 1025] if dim == 2 else [9, 17, 33, 65, 129]))
        lmax = int(np.log2(npts - 1))
        levels = int(rng.integers(1, lmax + 1))
        v0, v1 = int(rng.integers(0, 6)), int(rng.integers(1, 5))
        mesh = int(rng.choice([0, 0, 1, 2])) if dim == 2 else 0
        scale = float(rng.choice([0.8, 1.0, 6.0 / 7.0, 0.5]))
        fuse = int(rng.choice([-1, -1, 0, 32, 63, 63 | 256 | 512, 63 | 256 | 512 | 1024 | 2048, 63 | 256 | 512 | 8192, int(rng.integers(0, 16384)),
                               int(rng.integers(0, 16384)) | 32]))
        pair = int(rng.choice([0, 7, 7, 15, 31]))
        graph = int(rng.choice([-1, -1, 0]))
        prec = str(rng.choice(["fp64", "fp64", "mixed"])) if dim == 3 else "fp64"
        if v0 == 0 and levels > 1:
            v0 = 1
        # what is done with the solver: solve to the tolerance | the bench's loop (fixed count, norms deferred, in one or two calls) followed
        # by a reset and a solve on the same handle | the same with the counts the other way round
        mode = str(rng.choice(["solve", "solve", "cycles", "cycles+solve"]))
        k1, k2 = int(rng.integers(1, 5)), int(rng.integers(0, 4))
        cheb = dim == 2 and mesh == 0 and prec == "fp64" and bool(rng.integers(0, 4) == 0)
        if v1 == 0 and cheb:
            v1 = 1
        kso = dict(ksp_type="chebyshev", eigenvalues=(0.2, 2.0)) if cheb else {}
        kor = dict(ksp_type=1, emin=0.2, emax=2.0) if cheb else {}
        tag = (f"dim={dim} npts={npts} levels={levels} v=({v0},{v1}) mesh={mesh} scale={scale:.4f} fuse={fuse} pair_min_n={pair} graph={graph} {prec} "
               f"{'chebyshev ' if cheb else ''}{mode} {k1}+{k2}")

        def oracle_run(fixed):
            if prec == "mixed":
                return orc.vcycle_mixed(npts, levels, v0, v1, maxiter=max(40, fixed), scale=scale, fixed_cycles=fixed)
            return orc.vcycle(dim, npts, levels, v0, v1, maxiter=max(40, fixed), scale=scale, use_csr=1 if mesh else 0, mesh=mesh, fixed_cycles=fixed, **kor)

        def same(it, u, rn, ref):
            return (it == ref["iters"] and np.array_equal(u, ref["u"]) and
                    np.max(np.abs(rn - ref["rnorm"]) / np.maximum(ref["rnorm"], 1e-300)) <= 1e-10)
        try:
            s = Solver(dim, npts, levels, v=(v0, v1), maxiter=40, scale=scale, fuse=fuse, pair_min_n=pair, mesh=mesh, graph=graph, precision=prec, **kso)
            s.set_rhs_problem()
            results = []
            if mode != "solve":
                s.cycles(k1)
                if k2:
                    s.cycles(k2)
                s.sync()
                results.append((k1 + k2, s.iterations, s.solution(), s.rnorm))
                if mode == "cycles+solve":
                    s.reset()
            if mode != "cycles":
                it = s.solve()
                results.append((0, it, s.solution(), s.rnorm))
            s.close()
        except Exception as e:                               # a configuration the solver refuses is reported, not counted as a mismatch
            print("REFUSED", tag, str(e)[:120], flush=True)
            continue
        for fixed, it, u, rn in results:
            ref = oracle_run(fixed)
            if not same(it, u, rn, ref):
                bad += 1
                print("MISMATCH", tag, f"(leg: {'fixed ' + str(fixed) if fixed else 'solve'})", "iters", it, ref["iters"], "max|du|", float(np.max(np.abs(u - ref["u"]))),
                      flush=True)
                break
    print(f"{count} configurations, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
