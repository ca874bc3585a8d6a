"""CPU, world_size 2 (and 3), gloo: the N>1 path without a GPU, twice over.
  * dist_cpu_worker.py: an ORACLE-side model of the decomposition -- the cycle re-stated in Python over the oracle's slab operators;
    from the product it borrows mg_slab_range only.  It pins WHAT the slabs must compute.
  * dist_mock_worker.py: the PRODUCT's own slab cycle (csrc/mg_solver.c + csrc/mg_comm.c, unchanged) over the host-memory mock of the
    kernel ABI, one process per rank, halos and reductions through gloo (HostStagedComm)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,npts,levels,ldist", [(2, 33, 4, 2), (2, 33, 3, 3), (3, 33, 4, 1)])
def test_gloo_slab_vcycle_equals_single_rank(tmp_path, world, npts, levels, ldist):
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_cpu_worker.py"), str(r), str(world), port,
                               str(npts), str(levels), str(ldist), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=500)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    ref = Oracle().vcycle(3, npts, levels, 3, 3, maxiter=40, scale=6.0 / 7.0, use_csr=0)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert parts[0]["z0"] == 0 and parts[-1]["z1"] == npts - 2
    for a, b in zip(parts[:-1], parts[1:]):
        assert a["z1"] == b["z0"]
    for p in parts:
        assert int(p["it"]) == ref["iters"]
        assert np.allclose(p["rn"], ref["rnorm"], rtol=1e-13, atol=0)
    u = np.concatenate([p["u"] for p in parts])
    assert np.array_equal(u, ref["u"])


@pytest.fixture(scope="module")
def mock_solver_lib():
    """mg_solver.c + mg_comm.c + tests/mock_mgk.cpp as ONE shared library (test infrastructure, tests/_san/)"""
    import shutil
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no host compiler")
    root = os.path.dirname(HERE)
    out = os.path.join(HERE, "_san")
    os.makedirs(out, exist_ok=True)
    inc = "-I" + os.path.join(root, "include")
    objs = []
    for cc, std, src, obj in (("g++", "-std=c++17", os.path.join(HERE, "mock_mgk.cpp"), "dist_mock_mgk.o"),
                              ("gcc", "-std=c99", os.path.join(root, "multigrid_petsc_amd", "csrc", "mg_solver.c"), "dist_mg_solver.o"),
                              ("gcc", "-std=c99", os.path.join(root, "multigrid_petsc_amd", "csrc", "mg_comm.c"), "dist_mg_comm.o")):
        o = os.path.join(out, obj)
        p = subprocess.run([cc, std, "-O1", "-g", "-fPIC", "-ffp-contract=off", "-D_POSIX_C_SOURCE=200809L", inc, "-c", src, "-o", o],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout[-3000:]
        objs.append(o)
    so = os.path.join(out, "libmgsolve_mock.so")
    p = subprocess.run(["g++", "-shared", "-Wl,-Bsymbolic", "-o", so] + objs + ["-lm", "-lpthread", "-ldl"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    return so


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,npts,levels,dmin,precision", [(2, 33, 4, 15, "fp64"), (3, 33, 4, 15, "fp64"), (2, 65, 5, 15, "fp64"),
                                                             (2, 33, 4, 15, "mixed")])
def test_gloo_ranks_run_the_products_slab_cycle(tmp_path, mock_solver_lib, world, npts, levels, dmin, precision):
    """mg_solver.c under REAL ranks (processes, gloo) on the CPU: iteration count and residual history of every rank equal the oracle's
    (1e-12), the concatenated slabs equal the oracle's solution bit for bit (fp64), and the fixed-count loop with deferred norms that
    bench.py times gives the same history"""
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_mock_worker.py"), str(r), str(world), port, mock_solver_lib,
                               str(npts), str(levels), str(dmin), precision, str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=700)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    orc = Oracle()
    ref = orc.vcycle(3, npts, levels, 3, 3, maxiter=40, scale=6.0 / 7.0, use_csr=0) if precision == "fp64" else \
        orc.vcycle_mixed(npts, levels, 3, 3, maxiter=40, scale=6.0 / 7.0)
    parts = [np.load(tmp_path / f"mock_rank{r}.npz") for r in range(world)]
    assert int(parts[0]["z0"]) == 0 and sum(int(p["nz"]) for p in parts) == npts - 2
    for p in parts:
        assert int(p["it"]) == ref["iters"]
        assert np.abs(p["rn"] / ref["rnorm"] - 1).max() <= 1e-12
        assert np.abs(p["rn3"] / ref["rnorm"][:4] - 1).max() <= 1e-12
        assert np.array_equal(p["e"], parts[0]["e"])                    # all-reduced error sums: the same on every rank
    u = np.concatenate([p["u"] for p in parts])
    assert np.array_equal(u, ref["u"])
