"""helper of test_multirank_gpu.py: one of two processes that share GPU 0 and solve a 2-slab problem over RCCL"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
from multigrid_petsc_amd.comm import rccl_comm, rccl_unique_id   # noqa: E402
from multigrid_petsc_amd.solver import Solver                    # noqa: E402

if rank == 0:
    uid = rccl_unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60:
            print("RCCL_REFUSED no id file")
            sys.exit(0)
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
try:
    comm = rccl_comm(rank, world, 0, uid=uid)
except RuntimeError as e:
    print("RCCL_REFUSED", e, flush=True)
    sys.exit(0)
print("COMM_UP", rank, flush=True)       # from here on a hang or a mismatch is a FAILURE of the transport, not a refusal
from multigrid_petsc_amd.comm import selftest       # noqa: E402
from multigrid_petsc_amd.mgk import Mgk             # noqa: E402
m = Mgk(0)
selftest(comm.handle, m.ctx)
m.close()
print("SELFTEST_OK", rank, flush=True)
s = Solver(3, 33, 4, scale=6.0 / 7.0, maxiter=40, rank=rank, nranks=world, comm=comm.handle, dist_min_n=15)
s.set_rhs_problem()
it = s.solve()
ref = Solver(3, 33, 4, scale=6.0 / 7.0, maxiter=40)
ref.set_rhs_problem()
it1 = ref.solve()
z0, nz = s.level_planes(0)
u1 = ref.solution().reshape(31, 31, 31)[z0:z0 + nz].ravel()
assert it == it1 and np.array_equal(s.solution(), u1) and np.allclose(s.rnorm, ref.rnorm, rtol=1e-13, atol=0)
print("PAIR_OK", rank, it)
s.close()
ref.close()
comm.close()
