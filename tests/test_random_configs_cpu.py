"""CPU tier of tests/test_random_configs_gpu.py: the product's HOST logic under configurations drawn at random (fixed seeds) -- csrc/mg_solver.c over
tests/mock_mgk.cpp through the package's own Solver class (tools/stress_solver_mock.py), and the reference's unmodified driver over the drop-in's
host C on the same mock (tools/stress_refdriver.py with MG_STRESS_EXE = tests/_san/san_refdriver).  Iteration count equal to the oracle's, u
bit-identical.  Round 3: the coarse-level recording's odd-swap hole showed in 3 of 500 such draws on the code before the fix, in 0 of 2000 after."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_random_solver_configurations_on_the_host_mock_equal_the_oracle():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_solver_mock.py"), "150", "101"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=500, cwd=ROOT)
    assert p.returncode == 0 and "150 configurations, 0 mismatches" in p.stdout, p.stdout[-3000:]
    assert "REFUSED" not in p.stdout, p.stdout[-3000:]


@pytest.mark.timeout(900)
def test_random_reference_driver_options_on_the_host_mock_equal_the_oracle(tmp_path):
    exe = os.path.join(ROOT, "tests", "_san", "san_refdriver")
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("the reference's sources are not here: the sanitized link of its driver cannot be made")
    if not os.path.exists(exe):          # (tests/test_host_sanitized.py links it; alone, this test makes that module's fixture do it)
        subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_host_sanitized.py"), "-q", "-k", "bad_options"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert os.path.exists(exe)
    env = dict(os.environ, MG_STRESS_EXE=exe, MG_STRESS_MAXN="129", ASAN_OPTIONS="detect_leaks=0")     # (the reference's own leaks are not the subject)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_refdriver.py"), "60", "21"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=800, cwd=ROOT, env=env)
    assert p.returncode == 0 and "60 configurations, 0 mismatches" in p.stdout, p.stdout[-3000:]


@pytest.mark.timeout(900)
def test_ninety_one_byte_fine_level_on_slab_ranks_over_the_host_mock():
    """fuse bit 14 (round 3, second session): prolongation + two sweeps, the mid-iterate norm pass and the owed final sweep on z-slabs -- 2 / 3 / 4 / 8 loopback
    ranks over the mock, solve and the bench's fixed-count loop, overlap on / off: iteration counts, histories and the concatenated slabs equal the oracle's"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_slab91.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=800, cwd=ROOT,
                       env=dict(os.environ, MOCK="1", MOCK_MGK_STATS="1"))
    assert p.returncode == 0 and "bad 0" in p.stdout, p.stdout[-3000:]
    m = __import__("re").search(r"pj2_slab=(\d+) mid_slab=(\d+)", p.stdout)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0, p.stdout[-500:]          # the new passes did run
