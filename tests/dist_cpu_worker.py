"""Worker of tests/test_dist_cpu.py: one gloo rank of the z-slab V-cycle evaluated with the CPU ORACLE's slab
operators.  It exercises, without a GPU, exactly the host logic the product uses on N GPUs: the nested slab
split (mg_slab_range from libmgpetsc.so), which ghost planes every operator needs, the switch to replicated
coarse levels (all-gather), and the norm all-reduce -- with torch.distributed (gloo) as the transport."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import Oracle                                 # noqa: E402
from multigrid_petsc_amd.solver import slab_range, _lib   # noqa: E402


def main():
    rank, world, port, npts, levels, ldist, outdir = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]),
                                                     int(sys.argv[5]), int(sys.argv[6]), sys.argv[7])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    L = _lib()
    scale, v0, v1, maxiter = 6.0 / 7.0, 3, 3, 40
    n = [L.mg_grid_n(npts, l) for l in range(levels)]
    As = [orc.level_stencil(3, npts, l)[0] for l in range(levels)]
    z = [slab_range(npts, ldist, l, rank, world) if l < ldist else (0, n[l]) for l in range(levels)]
    nz = [b - a for a, b in z]
    zs = [slab_range(npts, ldist, ldist, r, world) for r in range(world)] if ldist < levels else None

    def halo(x, l):
        """returns (lo ghost plane, hi ghost plane) of slab array x (nz,n,n) of level l"""
        if l >= ldist:
            return None, None
        x3 = x.reshape(nz[l], n[l], n[l])
        lo = torch.zeros(n[l] * n[l], dtype=torch.float64)
        hi = torch.zeros(n[l] * n[l], dtype=torch.float64)
        ops = []
        if rank > 0:
            ops += [dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(x3[0]).ravel()), rank - 1),
                    dist.P2POp(dist.irecv, lo, rank - 1)]
        if rank < world - 1:
            ops += [dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(x3[-1]).ravel()), rank + 1),
                    dist.P2POp(dist.irecv, hi, rank + 1)]
        for w in dist.batch_isend_irecv(ops) if ops else []:
            w.wait()
        return (lo.numpy() if rank > 0 else None), (hi.numpy() if rank < world - 1 else None)

    def allsum(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0])

    u = [np.zeros(nz[l] * n[l] * n[l]) for l in range(levels)]
    b = [np.zeros(nz[l] * n[l] * n[l]) for l in range(levels)]
    guess = [False] * levels
    b[0] = orc.rhs(3, npts).reshape(n[0], n[0], n[0])[z[0][0]:z[0][1]].ravel().copy()

    def smooth(l, its):
        for it in range(its):
            if it == 0 and not guess[l]:
                u[l] = orc.jacobi(3, n[l], As[l], scale, b[l], u[l], zero_guess=True, nz=nz[l])
            else:
                lo, hi = halo(u[l], l)
                u[l] = orc.jacobi(3, n[l], As[l], scale, b[l], u[l], nz=nz[l], zlo=lo, zhi=hi)

    def residual(l):
        lo, hi = halo(u[l], l)
        return orc.residual(3, n[l], As[l], b[l], u[l], nz=nz[l], zlo=lo, zhi=hi)

    bnorm = np.sqrt(allsum(orc.sumsq(b[0])))
    rn = [np.sqrt(allsum(orc.sumsq(residual(0))))]
    it = 0
    while it < maxiter and 1e8 * bnorm > rn[-1] and rn[-1] > 1e-7 * bnorm:
        smooth(0, v0)
        if it == 0:
            guess[0] = True
        for l in range(1, levels):
            r = residual(l - 1)
            _, rhi = halo(r, l - 1)
            if l - 1 < ldist <= l:           # slab -> replicated: my coarse planes, then all-gather
                c0, c1 = zs[rank]
                part = orc.restrict(3, n[l - 1], r, nzf=nz[l - 1], nzc=c1 - c0, fzhi=rhi)
                pieces = [torch.zeros((e - a) * n[l] * n[l], dtype=torch.float64) for a, e in zs]
                dist.all_gather(pieces, torch.from_numpy(part)) if len({p.numel() for p in pieces}) == 1 else None
                if len({p.numel() for p in pieces}) != 1:      # ragged: one broadcast per producer
                    for src in range(world):
                        if src == rank:
                            pieces[src] = torch.from_numpy(part.copy())
                        dist.broadcast(pieces[src], src)
                b[l] = torch.cat(pieces).numpy().copy()
            else:
                b[l] = orc.restrict(3, n[l - 1], r, nzf=nz[l - 1], nzc=nz[l], fzhi=rhi)
            smooth(l, v1 if l == levels - 1 else v0)
            if l != levels - 1:
                guess[l] = True
        for l in range(levels - 2, -1, -1):
            if l < ldist <= l + 1:           # replicated coarse -> my slab: ghosts are the neighbouring planes
                c0, c1 = zs[rank]
                U = u[l + 1].reshape(n[l + 1], n[l + 1], n[l + 1])
                lo = np.ascontiguousarray(U[c0 - 1]).ravel() if c0 > 0 else None
                hi = np.ascontiguousarray(U[c1]).ravel() if c1 < n[l + 1] else None
                u[l] = orc.prolong_add(3, n[l], np.ascontiguousarray(U[c0:c1]).ravel(), u[l], nzf=nz[l], nzc=c1 - c0,
                                       czlo=lo, czhi=hi)
            else:
                lo, hi = halo(u[l + 1], l + 1)
                u[l] = orc.prolong_add(3, n[l], u[l + 1], u[l], nzf=nz[l], nzc=nz[l + 1], czlo=lo, czhi=hi)
            smooth(l, v0)
            if l != 0:
                guess[l] = False
        rn.append(np.sqrt(allsum(orc.sumsq(residual(0)))))
        it += 1
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), it=it, rn=np.array(rn), u=u[0], z0=z[0][0], z1=z[0][1])
    dist.barrier()
    dist.destroy_process_group()


main()
