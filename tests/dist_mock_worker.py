"""Worker of tests/test_dist_cpu.py (second half): one gloo rank running the PRODUCT's slab cycle -- csrc/mg_solver.c and csrc/mg_comm.c,
unchanged -- over tests/mock_mgk.cpp (the kernel ABI in host memory) with the host-staged transport of multigrid_petsc_amd/comm.py
(HostStagedComm: the mg_comm hooks on torch.distributed).  World size 2 or 3 on the CPU: every decision mg_solver.c takes under ranks
(slab ranges, which exchange a pass needs, grouped exchanges, slab -> replicated all-gather, deferred norms) runs in separate processes
that only talk through gloo.  The library is injected into the package's loader cache HERE, in test code: the product itself has no
switch that makes it load anything but its HIP libraries."""
import ctypes
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, so = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    npts, levels, dmin, precision, outdir = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8], sys.argv[9]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import multigrid_petsc_amd._lib as loader
    lib = ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)
    loader._cache["mgk"] = lib
    loader._cache["mgpetsc"] = lib
    from multigrid_petsc_amd.comm import HostStagedComm
    from multigrid_petsc_amd.solver import Solver
    comm = HostStagedComm(rank, world, dist)
    s = Solver(3, npts, levels, v=(3, 3), maxiter=40, scale=6.0 / 7.0, rank=rank, nranks=world, comm=comm.handle,
               dist_min_n=dmin, pair_min_n=7, precision=precision)
    s.set_rhs_problem()
    it = s.solve()
    rn, u, e = s.rnorm, s.solution(), s.error_norms()
    planes = [s.level_planes(l) for l in range(levels)]
    # bench.py's loop: fixed count, norms deferred to the end
    s.reset()
    s.cycles(3)
    s.sync()
    rn3 = s.rnorm
    s.close()
    np.savez(os.path.join(outdir, f"mock_rank{rank}.npz"), it=it, rn=rn, u=u, e=e, z0=planes[0][0], nz=planes[0][1], rn3=rn3)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
