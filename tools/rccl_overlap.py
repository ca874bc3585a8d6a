"""Does RCCL's send/recv kernel run BESIDE a marching kernel that fills the chip?  (last paragraphs: the question the
8-GPU node has to answer first.)  One GPU, a 1-rank RCCL communicator with this rank as its own peer -- the only send/recv a single GPU
can run, but the real RCCL kernel on the real high-priority comm stream -- moving two fine-level planes (2 x 8.4 MB) while the two-sweep
pass marches over a slab of 1023 x 1023 x 128 on the compute stream, for the per-context chunk hints 0 (long streams), 32, 8.
Prints the wall time of the kernel alone, the exchange alone, and both queued together (exchange first, as the cycle does)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_petsc_amd.comm import rccl_comm, _lib   # noqa: E402
from multigrid_petsc_amd.mgk import Mgk               # noqa: E402

m = Mgk(0)
c = rccl_comm(0, 1, 0)
L = _lib()
L.mg_comm_rccl_self_sendrecv_async.restype = C.c_int
L.mg_comm_rccl_self_sendrecv_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int]
m.L.mgk_ctx_set_chunk_planes.argtypes = [C.c_void_p, C.c_int]
n, nz = 1023, 128
g = m.geom(3, n, n, nz)
rng = np.random.default_rng(1)
u, b, out = m.field(g), m.field(g), m.field(g)
q = float((n + 1) ** 2)
As = [q, q, q, -6.0 * q, q, q, q]
planes = 2
cnt = int(g.plane) * planes
src, dst = m.alloc(8 * cnt), m.alloc(8 * cnt)


def kernel():
    m._chk(m.L.mgk_jacobi2_f64(m.ctx, C.byref(g), m.coef(As), 1.0 / As[3], 6.0 / 7.0, b, u, out, None))


def xchg():
    rc = L.mg_comm_rccl_self_sendrecv_async(c.handle, m.ctx, src, dst, cnt, 8)
    assert rc == 0, L.mg_comm_last_error()


def timed(fs, reps=20):
    for f in fs:
        f()
    m.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for f in fs:
            f()
        m.sync()
    return (time.perf_counter() - t0) / reps * 1e6


res = {"plane_MB": g.plane * 8 / 1e6, "planes_per_exchange": planes, "slab": [n, n, nz]}
# latency floor of one grouped send/recv on the comm stream (host call + RCCL kernel launch + completion), message sizes 128 B ... 2 planes
for nd in (16, 1 << 12, 1 << 16, 1 << 20, cnt):
    def small(nd=nd):
        rc = L.mg_comm_rccl_self_sendrecv_async(c.handle, m.ctx, src, dst, nd, 8)
        assert rc == 0, L.mg_comm_last_error()
    t = timed([small], reps=50)
    res[f"exchange_{nd * 8}_bytes_us"] = t
    print(f"grouped self send/recv of {nd * 8} B: {t:.1f} us", flush=True)
for chunk in (0, 32, 8):
    m._chk(m.L.mgk_ctx_set_chunk_planes(m.ctx, chunk))
    tk, tx = timed([kernel]), timed([xchg])
    tb, tb2 = timed([xchg, kernel]), timed([kernel, xchg])
    res[f"chunk_{chunk}"] = {"kernel_us": tk, "exchange_us": tx, "exchange_then_kernel_us": tb, "kernel_then_exchange_us": tb2,
                             "hidden_fraction_of_exchange": (tk + tx - tb) / tx}
    print(f"chunk {chunk}: kernel {tk:.0f} us, exchange {tx:.0f} us, together {tb:.0f} us (kernel first: {tb2:.0f}) -> "
          f"{100 * (tk + tx - tb) / tx:.0f} % of the exchange hidden", flush=True)
print(json.dumps(res))
c.close()
m.close()
