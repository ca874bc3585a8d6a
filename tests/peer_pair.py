"""Worker of tests/test_multirank_gpu.py: one PROCESS of the peer transport (include/mg_comm.h) -- all ranks share GPU 0, so the IPC handles
are of the same device: mapping, flag words, mailbox slots, gather box and the all-reduce slots run exactly as between GPUs; only the plane
copies are same-device copies instead of copy-engine transfers over xGMI.  usage: peer_pair.py rank world port npts levels dist_min_n out"""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigrid_petsc_amd.comm import peer_comm, selftest          # noqa: E402
from multigrid_petsc_amd.mgk import Mgk                           # noqa: E402
from multigrid_petsc_amd.solver import Solver                     # noqa: E402


def main():
    rank, world, port, npts, levels, dmin, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = Mgk(0)
    g0 = m.geom(3, npts - 2)
    comm = peer_comm(rank, world, 0, dist, 8 * g0.plane, 5, 8 * g0.total)
    print("COMM_UP", flush=True)
    selftest(comm.handle, m.ctx)
    print("SELFTEST_OK", flush=True)
    m.close()
    s = Solver(3, npts, levels, scale=6.0 / 7.0, maxiter=60, rank=rank, nranks=world, comm=comm.handle, dist_min_n=dmin, pair_min_n=15)
    s.set_rhs_problem()
    it = s.solve()
    rn, u, e = s.rnorm, s.solution(), s.error_norms()
    s.reset()
    s.cycles(3)                     # bench.py's loop: norms deferred, all-reduced on the device at the end
    s.sync()
    rn3 = s.rnorm
    s.close()
    np.savez(os.path.join(out, f"peer_rank{rank}.npz"), it=it, rn=rn, u=u, e=e, rn3=rn3)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()
    print("PAIR_OK", flush=True)


if __name__ == "__main__":
    main()
