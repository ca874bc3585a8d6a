"""CPU, world_size 2 (and 3), gloo: the N>1 host logic of the product without a GPU.  See dist_cpu_worker.py."""
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
