"""GPU tier: configurations drawn at random (fixed seeds) instead of picked by hand -- dimension, size, depth, sweep counts, damping, mesh, pair
threshold, fuse bits, graph on / off for the single-rank solver; rank count, distribution depth, overlap, chunk hint, precision for z-slab ranks
(loopback threads); options of the reference's unmodified driver over the drop-in.  Iteration count equal to the oracle's, u bit-identical.
tools/stress_solver.py / stress_slabs.py / stress_refdriver.py run the same draws with larger counts (round 3: 500 + 210 + 280, no mismatch)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, count, seed, timeout):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(count), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0 and f"{count} configurations, 0 mismatches" in p.stdout, p.stdout[-3000:]
    assert "REFUSED" not in p.stdout, p.stdout[-3000:]


@pytest.mark.timeout(600)
def test_random_solver_configurations_equal_the_oracle():
    _run("stress_solver.py", 60, 11, 500)


@pytest.mark.timeout(600)
def test_random_slab_configurations_equal_the_oracle():
    _run("stress_slabs.py", 20, 5, 500)


@pytest.mark.timeout(600)
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "build", "refdriver", "poisson")), reason="build/refdriver/poisson absent (linked where /root/reference exists)")
def test_random_reference_driver_options_equal_the_oracle():
    """the reference's unmodified driver over the drop-in: size, depth, -v, -mesh, -map, damping and the drop-in's fast-path switches drawn at random"""
    _run("stress_refdriver.py", 30, 21, 500)
