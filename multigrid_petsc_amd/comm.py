"""ctypes binding of include/mg_comm.h: communicator handles for the slab-decomposed solver.

rccl_comm()      one process per GPU (bench.py under torch.distributed.run); the 128-byte RCCL unique id
                 is created on rank 0 and shipped through the caller's control-plane group.
LoopbackWorld    all ranks are threads of this process on one GPU (tests)."""
import ctypes as C
import sys
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
    L.mg_comm_loopback_inject_fault.argtypes = [C.c_void_p, C.c_int]
    L.mg_comm_last_error.restype = C.c_char_p
    L.mg_comm_destroy.argtypes = [C.c_void_p]
    L.mg_comm_halo.restype = C.c_int
    L.mg_comm_halo.argtypes = [C.c_void_p] * 4
    L.mg_comm_allreduce_sum.restype = C.c_int
    L.mg_comm_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.mg_comm_phantom_create.restype = C.c_void_p
    L.mg_comm_phantom_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
    L.mg_comm_peer_create.restype = C.c_void_p
    L.mg_comm_peer_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p]
    L.mg_comm_peer_connect.restype = C.c_int
    L.mg_comm_peer_connect.argtypes = [C.c_void_p, C.c_void_p]
    L.mg_comm_selftest.restype = C.c_int
    L.mg_comm_selftest.argtypes = [C.c_void_p, C.c_void_p]
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


def selftest(comm_handle, ctx):
    """mg_comm_selftest: rank-coded planes through every hook of the transport, checked on every rank (collective)."""
    rc = _lib().mg_comm_selftest(comm_handle, ctx)
    if rc:
        raise RuntimeError(f"transport self-test failed (rc={rc}): " + _lib().mg_comm_last_error().decode())


def phantom_comm(rank, world, lat_us=20.0, link_gbs=60.0):
    """one rank of a `world`-rank run alone on its GPU: exchanges are device copies + a modelled link time (timing aid)"""
    return Comm(_lib().mg_comm_phantom_create(rank, world, lat_us, link_gbs))


def rccl_unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    rc = _lib().mg_comm_rccl_unique_id(buf)
    if rc:
        raise RuntimeError("ncclGetUniqueId failed: " + _lib().mg_comm_last_error().decode())
    return buf.raw


def rccl_comm(rank, world, device, dist=None, uid=None):
    """dist: an initialised torch.distributed module (any backend) used only to broadcast the id.
    Collective-safe: rank 0 ALWAYS broadcasts a (status, id) pair, so a failure to make the id on rank 0 reaches every rank
    as an exception instead of leaving them in a broadcast that never comes."""
    if uid is None:
        box = [None]
        if rank == 0:
            try:
                box = [("ok", rccl_unique_id())]
            except Exception as e:      # noqa: BLE001 - shipped to every rank below
                box = [("error", str(e))]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        status, uid = box[0]
        if status != "ok":
            raise RuntimeError("rank 0 could not create the RCCL unique id: " + str(uid))
    return Comm(_lib().mg_comm_rccl_create(rank, world, uid, device))


PEER_BLOB_BYTES = 192


def peer_comm(rank, world, device, dist, plane_bytes_max, fields_max=5, gather_bytes=None):
    """The peer transport (include/mg_comm.h: IPC-mapped mailboxes + flag words, copy-engine plane copies): every rank creates its
    mailbox / flag block / gather box and publishes the 192 bytes of IPC handles; `dist` (an initialised torch.distributed module, any
    backend) all-gathers the blobs; then every rank maps its peers.  Collective-safe like rccl_comm: a rank that could not create its
    side still takes part in the all-gather (with an error marker), and every rank raises."""
    L = _lib()
    blob = C.create_string_buffer(PEER_BLOB_BYTES)
    h = L.mg_comm_peer_create(rank, world, device, int(plane_bytes_max), int(fields_max), int(gather_bytes or plane_bytes_max), blob)
    mine = ("ok", blob.raw) if h else ("error", L.mg_comm_last_error().decode())
    allb = [None] * world
    if world > 1:
        dist.all_gather_object(allb, mine)
    else:
        allb = [mine]
    bad = [(r, b[1]) for r, b in enumerate(allb) if b[0] != "ok"]
    if bad:
        if h:
            L.mg_comm_destroy(C.c_void_p(h))
        raise RuntimeError("peer transport: " + "; ".join(f"rank {r}: {m}" for r, m in bad))
    comm = Comm(h)
    joined = b"".join(b[1] for b in allb)
    rc = L.mg_comm_peer_connect(comm.handle, joined)
    ok = [None] * world
    if world > 1:
        dist.all_gather_object(ok, rc)
    else:
        ok = [rc]
    if any(ok):
        msg = L.mg_comm_last_error().decode()
        comm.close()
        raise RuntimeError(f"peer transport: mapping the peers' memory failed on rank(s) {[r for r, v in enumerate(ok) if v]}: {msg}")
    return comm


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

    def inject_fault(self, rank):
        """test aid: from now on `rank` (>= 1) receives a wrong plane as its lo ghost (-1: off)"""
        _lib().mg_comm_loopback_inject_fault(self.shared, rank)

    def close(self):
        if self.shared:
            _lib().mg_comm_loopback_shared_destroy(self.shared)
            self.shared = None


# ---------------------------------------------------------------------------------------------
# host-staged transport: the same mg_comm hooks implemented with device<->host copies and the
# caller's torch.distributed group (gloo).  A slow but dependency-free fallback for the case that
# RCCL cannot build its communicator (it is also what lets two ranks share ONE GPU in the tests).
# ---------------------------------------------------------------------------------------------
class _Geom(C.Structure):
    _fields_ = [("dim", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("pitch", C.c_int), ("plane", C.c_long), ("org", C.c_long), ("total", C.c_long)]


_HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Geom), C.c_int, C.c_void_p)
_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Geom), C.POINTER(C.c_int), C.c_int, C.c_void_p)
_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_void_p)
_BARRIER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
_DESTROY = C.CFUNCTYPE(None, C.c_void_p)


class _MgComm(C.Structure):
    # mirrors struct mg_comm (include/mg_comm.h); the three optional hooks stay NULL: the C side then falls back to
    # halo() per field (the far planes staged in their fields first) and to the host all-reduce
    _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("impl", C.c_void_p), ("halo", _HALO),
                ("allgather_planes", _GATHER), ("allreduce_sum", _REDUCE), ("barrier", _BARRIER), ("destroy", _DESTROY),
                ("halo_n", C.c_void_p), ("allreduce_sum_dev", C.c_void_p), ("exchange", C.c_void_p), ("check", C.c_void_p)]


class HostStagedComm:
    """mg_comm whose hooks run in Python: planes are copied to the host, exchanged with torch.distributed
    (any CPU-capable backend), and copied back.  Blocking; correctness fallback, not a performance path."""

    def __init__(self, rank, world, dist):
        import numpy as np
        import torch
        from ._lib import load_mgk
        self.np, self.torch, self.dist = np, torch, dist
        self.rank, self.world = rank, world
        self.K = load_mgk()
        self.K.mgk_sync.argtypes = [C.c_void_p, C.c_void_p]
        self.K.mgk_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        self.K.mgk_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        self._cb = (_HALO(self._halo), _GATHER(self._gather), _REDUCE(self._reduce), _BARRIER(self._barrier), _DESTROY(self._destroy))
        self.struct = _MgComm(rank, world, None, *self._cb, None, None, None, None)
        self.handle = C.c_void_p(C.addressof(self.struct))

    # device plane <-> host tensor
    def _get(self, ctx, addr, nbytes):
        t = self.torch.empty(nbytes, dtype=self.torch.uint8)
        if self.K.mgk_d2h(ctx, C.c_void_p(t.data_ptr()), C.c_void_p(addr), nbytes):
            raise RuntimeError("mgk_d2h failed")
        return t

    def _put(self, ctx, addr, t):
        if self.K.mgk_h2d(ctx, C.c_void_p(addr), C.c_void_p(t.data_ptr()), t.numel()):
            raise RuntimeError("mgk_h2d failed")

    def _halo(self, c, ctx, field, gptr, esz, stream):
        try:
            g = gptr.contents
            pb = esz * g.plane
            self.K.mgk_sync(ctx, stream)
            ops, recv = [], []
            if self.rank > 0:
                ops.append(self.dist.P2POp(self.dist.isend, self._get(ctx, field + pb, pb), self.rank - 1))
                lo = self.torch.empty(pb, dtype=self.torch.uint8)
                ops.append(self.dist.P2POp(self.dist.irecv, lo, self.rank - 1))
                recv.append((field, lo))
            if self.rank < self.world - 1:
                ops.append(self.dist.P2POp(self.dist.isend, self._get(ctx, field + g.nz * pb, pb), self.rank + 1))
                hi = self.torch.empty(pb, dtype=self.torch.uint8)
                ops.append(self.dist.P2POp(self.dist.irecv, hi, self.rank + 1))
                recv.append((field + (g.nz + 1) * pb, hi))
            for w in (self.dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            for addr, t in recv:
                self._put(ctx, addr, t)
            return 0
        except Exception as e:      # noqa: BLE001 - reported through the C return code
            print("[HostStagedComm.halo]", e, file=sys.stderr, flush=True)
            return 10003

    def _gather(self, c, ctx, field, gptr, zstart, esz, stream):
        try:
            g = gptr.contents
            pb = esz * g.plane
            self.K.mgk_sync(ctx, stream)
            for src in range(self.world):
                n = (zstart[src + 1] - zstart[src]) * pb
                if n == 0:
                    continue
                addr = field + (zstart[src] + 1) * pb
                t = self._get(ctx, addr, n) if src == self.rank else self.torch.empty(n, dtype=self.torch.uint8)
                self.dist.broadcast(t, src)
                if src != self.rank:
                    self._put(ctx, addr, t)
            return 0
        except Exception as e:      # noqa: BLE001
            print("[HostStagedComm.allgather]", e, file=sys.stderr, flush=True)
            return 10003

    def _reduce(self, c, ctx, vals, n, stream):
        try:
            t = self.torch.tensor([vals[q] for q in range(n)], dtype=self.torch.float64)
            self.dist.all_reduce(t)
            for q in range(n):
                vals[q] = float(t[q])
            return 0
        except Exception as e:      # noqa: BLE001
            print("[HostStagedComm.allreduce]", e, file=sys.stderr, flush=True)
            return 10003

    def _barrier(self, c, ctx):
        self.dist.barrier()
        return 0

    def _destroy(self, c):
        return None

    def close(self):
        self.handle = None
