"""bench.py's CPU legs on the CPU tier: the reference's own loop over the host CSR backend (build/refdriver/poisson_cpu, a
bench-only artefact made by __graft_entry__.build() where the reference tree exists) must solve the reference's problem to the
reference's tolerance and yield a number; its iteration count is the oracle's."""
import importlib.util
import os

import pytest

from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_cpu_legs", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_loop_on_the_host_csr_backend():
    if not os.path.exists(os.path.join(ROOT, "build", "refdriver", "poisson_cpu")):
        pytest.skip("build/refdriver/poisson_cpu is not built (no reference tree at build time)")
    r = _bench().cpu_reference_loop(2, npts=129)
    assert r.get("value") and r["value"] > 0 and r["kind"] == "port" and r["cores"] >= 1, r
    ref = Oracle().vcycle(2, 129, 7, 3, 3, maxiter=1000, scale=0.8, use_csr=1)
    assert f"in {ref['iters']} cycles" in r["sample"], (r["sample"], ref["iters"])
