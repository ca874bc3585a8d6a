"""multigrid_petsc_amd -- MI355X-native geometric-multigrid V-cycle (drop-in for the hot path of
SyamVangara/multigrid-petsc, `-cycle 0`).

The product is two C-ABI shared libraries built from ``csrc/``:

* ``libmgk.so``     hand-written gfx950 HIP kernels + thin runtime wrappers (``include/mgk.h``)
* ``libmgpetsc.so`` C99 host side: level hierarchy, the V-cycle driver that mirrors
  ``MultigridVcycle`` (reference ``src/solver.c:1414-1575``) and the PETSc ``Mat/Vec/KSP``
  call surface the reference driver links against (``include/petscksp.h``, ``include/mgsolve.h``)

This Python package is only plumbing around them (ctypes bindings used by ``bench.py`` and the
tests).  There is no CPU fallback: importing works anywhere, but creating a context without the
built libraries or without a HIP device raises.
"""
from ._lib import load_mgk, load_mgpetsc, LibraryMissing  # noqa: F401
