"""CPU tier: the PETSc-surface drop-in's host logic (petsc_shim.c) linked against the host-memory mock of the kernel ABI
(tests/mock_mgk.cpp) as a shared library and driven through ctypes -- the lazy temporaries in call orders the reference's loop does not
use, and the adoption rule of the speculative sweep (tests/shim_semantics.py; the same functions run on the GPU tier over the real
libmgpetsc.so).  Test infrastructure: nothing under multigrid_petsc_amd/ links the mock."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_san")


@pytest.fixture(scope="module")
def mock_shim():
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no host compiler")
    os.makedirs(OUT, exist_ok=True)
    inc = "-I" + os.path.join(ROOT, "include")
    so = os.path.join(OUT, "libmgpetsc_mock.so")
    objs = []
    for cc, std, src, obj in (("g++", "-std=c++17", os.path.join(ROOT, "tests", "mock_mgk.cpp"), "mock_mgk_pic.o"),
                              ("gcc", "-std=c99", os.path.join(ROOT, "multigrid_petsc_amd", "csrc", "petsc_shim.c"), "petsc_shim_pic.o")):
        o = os.path.join(OUT, obj)
        p = subprocess.run([cc, std, "-O1", "-g", "-fPIC", "-ffp-contract=off", "-D_POSIX_C_SOURCE=200809L", inc, "-c", src, "-o", o],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout[-3000:]
        objs.append(o)
    # -Bsymbolic: the library's own mgk_* / PETSc-surface definitions, whatever else the process has loaded
    p = subprocess.run(["g++", "-shared", "-Wl,-Bsymbolic", "-o", so] + objs + ["-lm", "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    return so


def _check(so, which, *more):
    # a process of its own: the drop-in ends the process on a fatal error, and the test process may hold the real libraries
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shim_semantics.py"), so, which] + list(more), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "SEMANTICS_OK " + which in p.stdout, (p.returncode, p.stdout[-3000:])


def test_lazy_temporaries_keep_petsc_semantics_on_the_mock(mock_shim):
    _check(mock_shim, "lazy")


def test_speculative_sweep_adoption_rule_on_the_mock(mock_shim):
    _check(mock_shim, "spec")


def test_residual_left_deferred_by_the_norm_pass_on_the_mock(mock_shim):
    _check(mock_shim, "keepr")


def test_recorded_coarse_subcycle_keeps_petsc_semantics_on_the_mock(mock_shim):
    _check(mock_shim, "tailrec")


def test_pcmg_level_vectors_after_the_tail_launch_on_the_mock(mock_shim):
    _check(mock_shim, "pcmgtail")


def test_richardson_with_lu_is_damped_on_the_mock(mock_shim):
    _check(mock_shim, "lu")


def test_random_programs_of_petsc_calls_keep_petsc_semantics_on_the_mock(mock_shim):
    """random_programs_keep_petsc_semantics: 25 programs of up to 60 calls (seed 1); round 3 ran 7 seeds x 40 programs over the mock: no deviation"""
    _check(mock_shim, "random", "1", "25")
