"""ctypes binding of include/mg_comm.h: communicator handles for the slab-decomposed solver.

rccl_comm()      one process per GPU (bench.py under torch.distributed.run); the 128-byte RCCL unique id
                 is created on rank 0 and shipped through the caller's control-plane group.
LoopbackWorld    all ranks are threads of this process on one GPU (tests)."""
import ctypes as C
import threading

from ._lib import load_mgpetsc

ID_BYTES = 128


def _lib():
    L = load_mgpetsc()
    if getattr(L, "_comm_sigs", False):
        return L
    L.mg_comm_rccl_unique_id.restype = C.c_int
    L.mg_comm_rccl_unique_id.argtypes = [C.c_void_p]
    L.mg_comm_rccl_create.restype = C.c_void_p
    L.mg_comm_rccl_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.mg_comm_loopback_shared_create.restype = C.c_void_p
    L.mg_comm_loopback_shared_create.argtypes = [C.c_int]
    L.mg_comm_loopback_shared_destroy.argtypes = [C.c_void_p]
    L.mg_comm_loopback_create.restype = C.c_void_p
    L.mg_comm_loopback_create.argtypes = [C.c_void_p, C.c_int]
    L.mg_comm_last_error.restype = C.c_char_p
    L.mg_comm_destroy.argtypes = [C.c_void_p]
    L.mg_comm_halo.restype = C.c_int
    L.mg_comm_halo.argtypes = [C.c_void_p] * 4
    L.mg_comm_allreduce_sum.restype = C.c_int
    L.mg_comm_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L._comm_sigs = True
    return L


class Comm:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("communicator creation failed: " + _lib().mg_comm_last_error().decode())
        self.handle = C.c_void_p(handle)

    def close(self):
        if self.handle:
            _lib().mg_comm_destroy(self.handle)
            self.handle = None


def rccl_unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    rc = _lib().mg_comm_rccl_unique_id(buf)
    if rc:
        raise RuntimeError("ncclGetUniqueId failed: " + _lib().mg_comm_last_error().decode())
    return buf.raw


def rccl_comm(rank, world, device, dist=None, uid=None):
    """dist: an initialised torch.distributed module (any backend) used only to broadcast the id"""
    if uid is None:
        box = [rccl_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        uid = box[0]
    return Comm(_lib().mg_comm_rccl_create(rank, world, uid, device))


class LoopbackWorld:
    """Run `fn(rank, comm_handle)` on `nranks` threads that share one GPU."""

    def __init__(self, nranks):
        self.nranks = nranks
        self.shared = _lib().mg_comm_loopback_shared_create(nranks)

    def run(self, fn):
        results, errors = [None] * self.nranks, [None] * self.nranks

        def work(r):
            comm = None
            try:
                comm = Comm(_lib().mg_comm_loopback_create(self.shared, r))
                results[r] = fn(r, comm.handle)
            except BaseException as e:   # noqa: BLE001 - re-raised in the caller
                errors[r] = e
            finally:
                if comm is not None:
                    comm.close()

        th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(self.nranks)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in errors:
            if e is not None:
                raise e
        return results

    def close(self):
        if self.shared:
            _lib().mg_comm_loopback_shared_destroy(self.shared)
            self.shared = None
