#!/usr/bin/env python3
"""One rank's share of the 8-GPU 1023^3 V-cycle, timed on ONE GPU.

Solver(3, 1025, 10, rank=r, nranks=8) runs exactly the launches, streams and events of rank r of the 8-GPU job; the
neighbours' planes are stood in for by the phantom communicator (include/mg_comm.h): device copies of the rank's own boundary
planes, and every exchange holds the comm stream for  latency + bytes_per_direction / link_bandwidth  (a one-wavefront
wait kernel), so exchanges that the cycle does not hide show up in ms/cycle.  Three link models:
  free        0 us, infinite bandwidth: the compute + launch share alone
  expected    20 us per exchange, 60 GB/s per direction and link (RCCL grouped send/recv over one xGMI link)
  pessimistic 40 us, 40 GB/s
Writes a JSON object (stdout and, with --out, a file).  Numbers from the phantom runs are timings only."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multigrid_petsc_amd.comm import phantom_comm          # noqa: E402
from multigrid_petsc_amd.solver import Solver              # noqa: E402

MODELS = {"free": (0.0, 0.0), "expected": (20.0, 60.0), "pessimistic": (40.0, 40.0)}


def run(rank, nranks, npts, levels, lat, gbs, cycles, warmup, overlap, precision, **kw):
    c = phantom_comm(rank, nranks, lat, gbs)
    s = Solver(3, npts, levels, scale=6.0 / 7.0, maxiter=cycles + warmup + 1, rank=rank, nranks=nranks, comm=c.handle,
               overlap=overlap, precision=precision, **kw)
    s.set_rhs_problem()
    s.cycles(warmup)
    s.sync()
    t0 = time.perf_counter()
    s.cycles(cycles)
    s.sync()
    ms = 1e3 * (time.perf_counter() - t0) / cycles
    planes = [s.level_planes(l)[1] for l in range(levels)]
    s.close()
    c.close()
    return ms, planes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npts", type=int, default=1025)
    ap.add_argument("--levels", type=int, default=10)
    ap.add_argument("--nranks", type=int, default=8)
    ap.add_argument("--ranks", default="0,3,7")
    ap.add_argument("--cycles", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="fp64")
    ap.add_argument("--single-ms", type=float, default=0.0, help="ms per cycle of the 1-GPU run on the same box (for the ratio)")
    ap.add_argument("--out", default="")
    ap.add_argument("--models", default="free,expected,pessimistic")
    ap.add_argument("--overlap", default="1,0")
    ap.add_argument("--pair-min-n", type=int, default=0)
    ap.add_argument("--dist-min-n", type=int, default=0)
    ap.add_argument("--fuse", type=int, default=-1, help="mg_config.fuse (-1: the default bits)")
    ap.add_argument("--slab-chunks", default="-1", help="comma list of mg_config.slab_chunk values (-1: default 32, 0: long streams)")
    a = ap.parse_args()
    res = {"workload": f"3-D npts={a.npts}, {a.levels} levels, V(3,3), one rank of {a.nranks} z-slabs, phantom neighbours",
           "precision": a.precision, "single_gpu_ms_per_cycle": a.single_ms or None,
           "ideal_share_ms": (a.single_ms / a.nranks) if a.single_ms else None, "models": {}}
    for name, overlap, chunk in [(n_, o_, c_) for n_ in a.models.split(",") for o_ in [int(x) for x in a.overlap.split(",")]
                                 for c_ in [int(x) for x in a.slab_chunks.split(",")]]:
        lat, gbs = MODELS[name]
        if True:
            key = f"{name}{'' if overlap else '_no_overlap'}{'' if chunk < 0 else '_chunk%d' % chunk}"
            res["models"][key] = {"latency_us": lat, "link_GBs": gbs or None, "overlap": bool(overlap),
                                  "slab_chunk_planes": max(32, ((a.npts - 2) // a.nranks) // 4) if chunk < 0 else chunk, "ranks": {}}     # (< 0: the solver's default, a quarter of the slab, >= 32)
            for r in [int(x) for x in a.ranks.split(",")]:
                ms, planes = run(r, a.nranks, a.npts, a.levels, lat, gbs, a.cycles, a.warmup, overlap, a.precision,
                                 pair_min_n=a.pair_min_n, dist_min_n=a.dist_min_n, slab_chunk=chunk, fuse=a.fuse)
                res["models"][key]["ranks"][str(r)] = {"ms_per_cycle": ms, "local_planes_per_level": planes}
                print(f"[slab_share] {key:24s} rank {r}: {ms:.3f} ms/cycle", file=sys.stderr, flush=True)
            worst = max(v["ms_per_cycle"] for v in res["models"][key]["ranks"].values())
            res["models"][key]["worst_rank_ms"] = worst
            if a.single_ms:
                res["models"][key]["predicted_speedup_1_to_N"] = a.single_ms / worst
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
