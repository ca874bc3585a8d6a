"""Locate and load the in-tree shared libraries.  Fails loudly when they are missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class LibraryMissing(RuntimeError):
    pass


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise LibraryMissing(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C multigrid_petsc_amd/csrc`). There is no CPU fallback for the product path.")
    return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


_cache = {}


def load_mgk():
    if "mgk" not in _cache:
        _cache["mgk"] = _load("libmgk.so")
    return _cache["mgk"]


def load_mgpetsc():
    if "mgpetsc" not in _cache:
        load_mgk()
        _cache["mgpetsc"] = _load("libmgpetsc.so")
    return _cache["mgpetsc"]
