#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X multigrid V-cycle.

Metric (BASELINE.json): fp64 DOF-updates/sec per V-cycle at 1024^3; achieved HBM GB/s vs peak.
Workload: 3-D 7-point Poisson, npts = 1025 (1023^3 unknowns, the reference's vertex-centred
"1024^3"), 10 levels, V(3,3) with 3 coarsest sweeps (poisson.in:12), Richardson +
Jacobi smoother with scale 6/7, RHS f = -3 pi^2 sin sin sin, u0 = 0.  A "step" is one V-cycle
(src/solver.c:1531-1549), including its residual-norm reduction.

  python bench.py --gpus N --steps K --warmup W
  N > 1: launched by torch.distributed.run, one rank per GPU; the grid is z-slab decomposed
  (strong scaling: the total problem is fixed), halos travel over RCCL (C side, include/mg_comm.h).

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (fine-level Jacobi sweep, HIP-event timed
inside the timed region) and "cpu_baseline" (the CPU oracle's assembled-CSR V-cycle on a bounded
sample, timed on this box's host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# one process per GPU: the host driver of this pool supports dmabuf IPC only; RCCL's peer mapping fails without it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
JACOBI_BYTES_PER_DOF = 24.0  # read u, read b, write u' 


def cpu_baseline(args):
    """The oracle's assembled (AIJ/CSR) V-cycle -- the PETSc-equivalent data path of the reference --
    on a bounded sample of the same workload, OpenMP over this box's host cores."""
    threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import Oracle
    orc = Oracle()
    npts, levels, cycles = args.cpu_npts, args.cpu_levels, args.cpu_cycles
    r = orc.vcycle(3, npts, levels, 3, 3, maxiter=cycles, scale=6.0 / 7.0, use_csr=1,
                   fixed_cycles=cycles, want_u=False)
    dof = 0.0
    for l in range(levels):
        n = (npts - 1) // (2 ** l) - 1
        dof += (3 if l == levels - 1 else 6) * float(n) ** 3
    return {"value": dof * cycles / r["seconds"], "unit": "DOF-updates/s", "cores": orc.L.mgo_num_threads(),
            "kind": "port",
            "sample": f"oracle assembled-CSR V(3,3) cycle, 3-D npts={npts} ({npts - 2}^3 unknowns), "
                      f"{levels} levels, {cycles} cycles, solve loop only ({r['seconds']:.2f} s)"}


def cpu_reference_loop(threads, npts=2049):
    """the reference's OWN V-cycle loop (src/solver.c:1526-1553, unmodified objects of /root/reference/src/*.c) on the
    host cores, over a PETSc-equivalent CPU data path -- assembled CSR SpMV + Jacobi + BLAS-1 (MGPETSC_NO_RECOGNITION=1: the matrices
    the reference assembles are applied as matrices) -- timed by the reference's own `Solver walltime` window.  The binary
    (build/refdriver/poisson_cpu, made by __graft_entry__.build() where the reference tree exists) links the drop-in's host logic with
    a host-memory OpenMP backend of the kernel ABI; it is a bench-only artefact, PETSc itself being unavailable offline: kind "port".
    2-D, the family of BASELINE config 2, at a bounded npts."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "build", "refdriver", "poisson_cpu")
    if not os.path.exists(exe):
        return {"value": None, "error": "build/refdriver/poisson_cpu is not built (it needs the reference tree at build time)"}
    levels = 0
    while (npts - 1) % (2 ** levels) == 0 and (npts - 1) // (2 ** levels) - 1 >= 1:
        levels += 1
    threads = max(1, min(threads, os.cpu_count() or 1))
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "poisson.in"), "w") as f:
            f.write(f"-npts {npts}\n-mesh 0\n-iter 1000\n-grids {levels}\n-levels {levels}\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n"
                    "-pc_type jacobi\n-ksp_richardson_scale 0.8\n")
        env = dict(os.environ, MGPETSC_NO_RECOGNITION="1", OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close")
        p = subprocess.run([exe], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    m, it = re.search(r"Solver walltime:\s+(\S+)", p.stdout), re.search(r"Number of iterations:\s+(\d+)", p.stdout)
    if p.returncode != 0 or not m or not it:
        return {"value": None, "error": f"poisson_cpu rc={p.returncode}: {p.stdout[-300:]}"}
    wall, cycles = float(m.group(1)), int(it.group(1))
    dof = sum((3 if l == levels - 1 else 6) * float((npts - 1) // (2 ** l) - 1) ** 2 for l in range(levels))
    return {"value": dof * cycles / wall, "unit": "DOF-updates/s", "cores": threads, "kind": "port",
            "ms_per_cycle": 1e3 * wall / cycles,
            "sample": f"the reference's unmodified driver (src/solver.c:1526-1553 loop, its own Solver walltime {wall:.2f} s) over the drop-in's "
                      f"host logic + host CSR backend (assembled AIJ SpMV, Jacobi, BLAS-1; OpenMP), 2-D npts={npts} ({npts - 2}^2), "
                      f"{levels} levels, V(3,3), solved to 1e-7 in {cycles} cycles; PETSc itself is unavailable offline"}


def stream_ceiling(device, n_doubles):
    """Measured HBM ceiling on this box: STREAM triad a = b + s*c over arrays of the fine field's
    size, 24 B per element like a Jacobi sweep, HIP-event timed; best of a few launch shapes."""
    import ctypes as C
    from multigrid_petsc_amd.mgk import Mgk
    m = Mgk(device)
    n = int(n_doubles) & ~1
    a, b, c = (m.alloc(8 * n) for _ in range(3))
    m._chk(m.L.mgk_flat_fill(m.ctx, n, 1.0, b, None))
    m._chk(m.L.mgk_flat_fill(m.ctx, n, 2.0, c, None))
    t = C.c_void_p()
    m._chk(m.L.mgk_timer_create(m.ctx, C.byref(t)))
    best = None
    for blocks in (256, 4096, 65536):
        for nt in (1, 0):
            m._chk(m.L.mgk_stream_triad_f64(m.ctx, n, a, b, c, 0.5, blocks, nt, None))       # warm-up
            reps = 5
            m._chk(m.L.mgk_timer_start(m.ctx, t, None))
            for _ in range(reps):
                m._chk(m.L.mgk_stream_triad_f64(m.ctx, n, a, b, c, 0.5, blocks, nt, None))
            m._chk(m.L.mgk_timer_stop(m.ctx, t, None))
            ms = C.c_double()
            m._chk(m.L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms)))
            gbs = 24.0 * n * reps / (ms.value * 1e-3) / 1e9
            if best is None or gbs > best["GB/s"]:
                best = {"GB/s": gbs, "blocks": blocks, "nontemporal": bool(nt)}
    m.L.mgk_timer_destroy(m.ctx, t)
    for p_ in (a, b, c):
        m.free(p_)
    m.close()
    best["kernel"] = "STREAM triad a=b+s*c, fp64, 24 B/element, arrays of the fine field's size"
    return best


def plain_sweep_probe(device, n):
    """The one-sweep kernel (k_stencil<MODE_JACOBI>) on a fine grid of n^3, HIP-event timed: the cycle itself runs its
    sweeps in pairs (k_jacobi2) wherever it can, so the plain kernel is measured beside it."""
    import ctypes as C
    from multigrid_petsc_amd.mgk import Mgk
    m = Mgk(device)
    g = m.geom(3, n)
    u, b, out = m.field(g), m.field(g), m.field(g)
    for f in (u, b, out):
        m._chk(m.L.mgk_memset0(m.ctx, f, 8 * g.total, None))
    c = float((n + 1) ** 2)
    coef, dinv = m.coef([c, c, c, -6 * c, c, c, c]), -1.0 / (6 * c)
    t = C.c_void_p()
    m._chk(m.L.mgk_timer_create(m.ctx, C.byref(t)))
    m._chk(m.L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, u, out, None))
    reps = 6
    m._chk(m.L.mgk_timer_start(m.ctx, t, None))
    for _ in range(reps):
        m._chk(m.L.mgk_jacobi_f64(m.ctx, C.byref(g), coef, dinv, 6.0 / 7.0, b, u, out, None))
    m._chk(m.L.mgk_timer_stop(m.ctx, t, None))
    ms = C.c_double()
    m._chk(m.L.mgk_timer_elapsed_ms(m.ctx, t, C.byref(ms)))
    m.L.mgk_timer_destroy(m.ctx, t)
    for f in (u, b, out):
        m.free(f)
    m.close()
    per = ms.value / reps
    ach = JACOBI_BYTES_PER_DOF * float(n) ** 3 / (per * 1e-3) / 1e9
    return {"kernel": "k_stencil<double,3,..,MODE_JACOBI> one fine-level Jacobi sweep", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "launches": reps, "avg_launch_ms": per,
            "algorithmic_bytes_per_launch": JACOBI_BYTES_PER_DOF * float(n) ** 3}


# ---------------------------------------------------------------------------------------------
# whole-cycle roofline: the bytes ONE V(3,3) cycle of this implementation cannot avoid moving,
# per unknown of the level where a pass runs, times the level sizes
# ---------------------------------------------------------------------------------------------
def cycle_compulsory_bytes(dim, npts, levels, precision):
    """3-D fp64, fine level: norm + two sweeps in one pass 24 + last pre-smoothing sweep fused with residual/restriction 24+2*8/8 +
    fused prolongation sweep 24+8/8 + two sweeps in one pass 24 = 99 B/unknown on the full-row shapes those two kernels are built
    for (n = 255, 511, 1023; otherwise sweep+norm 24 + pair 24 + residual/restriction 17: 114); coarser levels have no norm pass and get their
    first (zero-guess) sweep from the restriction kernel (+8 written): 24 + 17 + 8/8 + 25 + 24 = 91.  2-D: the same four
    passes on levels >= 2047^2 (99 / 91), 124 on the levels >= 127^2 that sweep once per pass, the per-operation count of
    SURVEY 8(d3) (172) below.  Mixed precision: the fp32 inner cycle moves half of the 3-D figures (no norm pass: 45 / 45.5) plus the fp64 outer
    correction+residual pass, 32 B per fine unknown.  The coarsest level makes v1 = 3 plain sweeps (16 + 24 + 24)."""
    tot = 0.0
    for l in range(levels):
        n = (npts - 1) // (2 ** l) - 1
        N = float(n) ** dim
        if l == levels - 1 and levels > 1:
            per = 64.0
        elif dim == 3:
            # levels >= 1 that sweep in pairs (255 <= n <= 511 in fp64): the three pre-smoothing sweeps from the zero guess are ONE pass
            # that reads b alone (16 B) instead of the zero-guess sweep written by the restriction (8) + a two-sweep pass (24): 91 - 16
            # fine level, rows of 512 / 1024 (fuse bit 12): prolongation + two sweeps 25, third sweep + norm + next first sweep 24, two sweeps 24,
            # restriction 18 = 91 (the one sweep that materialises the last iterate when the iteration stops is not a per-cycle cost:
            # it is inside the timed region all the same)
            fine = 91.0 if (n + 1) in (512, 1024) else (99.0 if n + 1 == 256 else 114.0)
            per = (fine if precision != "mixed" else 114.0) if l == 0 else (75.0 if 255 <= n <= 511 else 91.0)
        else:
            # 2-D, round 2: levels >= 2047^2 (pairs of sweeps): norm + two sweeps 24, sweep + residual + restriction 26, prolongation
            # sweep 25, two sweeps 24 = 99 at level 0 (115 - 24 without the norm pass below it: 91); levels >= 127^2 without pairs:
            # zero-guess sweep from the restriction above (1), one sweep 24, sweep + residual + restriction 26, prolongation sweep 25,
            # two sweeps 48 = 124; below that the kernel-per-operation count
            # round 3 (fuse bit 13): THREE sweeps per pass on every level with kernels of its own (n >= 127): norm + three pre-smoothing
            # sweeps 24 (below level 0: the three from the zero guess over b alone, 16), residual + restriction 16 + 2, prolongation + three
            # post-smoothing sweeps 25: 67 at level 0, 59 below; the LDS tail (n <= 63): the per-operation count
            per = (67.0 if l == 0 else 59.0) if n >= 127 else 172.0
        if precision == "mixed":
            # fp32 inner cycle: half the fp64 bytes of a level that starts from the zero guess, with the three-sweep pass where the level
            # sweeps in pairs (n >= 255); level 0 adds the fp64 outer pass (u += e, r = b - A u -> fp32: 8 + 4 + 8 read, 8 + 4 written)
            half = (75.0 if n >= 255 else 91.0) / 2
            per = (half if l else half + 32.0) if not (l == levels - 1 and levels > 1) else 32.0
        tot += per * N
    return tot


def survey_cycle_bytes(dim, N0, seconds_per_cycle):
    """SURVEY 8(d3)'s per-OPERATION count for one V(3,3) cycle, [48 nu + 2 (16 + 8 / 2^d)] 2^d / (2^d - 1) + 16 bytes per fine unknown (219 B in 3-D,
    256 B in 2-D): what the cycle would move one kernel per PETSc call; the rate it implies may exceed the HBM peak -- that is what fusing and
    temporal blocking buy"""
    per = (48.0 * 3 + 2 * (16.0 + 8.0 / 2 ** dim)) * 2 ** dim / (2 ** dim - 1) + 16.0
    gbs = per * N0 / seconds_per_cycle / 1e9
    return {"bytes_per_fine_unknown": per, "GB/s": gbs, "of_peak": gbs / HBM_PEAK_GBS}


def golden_history(key):
    """normalised residual histories rnorm[i]/rnorm[0] of the bench configurations (tests/golden/bench_history.json, made by
    tools/make_bench_golden.py on an MI355X with THIS implementation's kernel-per-operation cycle, fuse=0: a regression guard for the
    fused cycles, NOT parity evidence -- that is what the oracle tests are: tests/test_headline_width_gpu.py compares the 511^3 solve
    and two cycles at 1023^3 with the CPU oracle bit for bit); None when the file has no entry"""
    path = os.path.join(ROOT, "tests", "golden", "bench_history.json")
    try:
        return json.load(open(path)).get(key)
    except Exception:
        return None


def check_history(key, rn):
    """every entry of the history this run produced against the committed one, while it is above the rounding floor"""
    gold = golden_history(key)
    if not gold:
        return None
    worst, ncmp = 0.0, 0
    for i in range(1, min(len(rn), len(gold))):
        g = gold[i]
        if g < 1e-9:
            break
        worst = max(worst, abs(rn[i] / rn[0] / g - 1.0))
        ncmp += 1
    return {"entries_compared": ncmp, "max_rel_dev": worst, "ok": bool(ncmp > 0 and worst <= 1e-9)}


def run_config(dim, npts, precision, steps, warmup, device):
    """one more BASELINE configuration on this GPU (single rank): ms per cycle, DOF-updates/s, the HIP-event timed dominant
    fine-level smoother launch, the whole-cycle roofline, and the residual history against the committed one"""
    from multigrid_petsc_amd.solver import Solver
    levels = 0
    while (npts - 1) % (2 ** levels) == 0 and (npts - 1) // (2 ** levels) - 1 >= 1:
        levels += 1
    scale = 6.0 / 7.0 if dim == 3 else 0.8
    s = Solver(dim, npts, levels, v=(3, 3), maxiter=steps + warmup + 1, scale=scale, device=device, precision=precision)
    s.set_rhs_problem()
    s.cycles(warmup)
    s.sync()
    s.profile(True)
    t0 = time.perf_counter()
    s.cycles(steps)
    s.sync()
    el = time.perf_counter() - t0
    pair_ms, pair_n = s.profile_read(1)
    one_ms, one_n = s.profile_read(0)
    n0 = npts - 2
    N0 = float(n0) ** dim
    esz = 8.0 if precision == "fp64" else 4.0
    out = {"workload": f"{dim}-D npts={npts} ({n0}^{dim}), {levels} levels, V(3,3), {precision}",
           "ms_per_cycle": 1e3 * el / steps, "dof_updates_per_s": s.dof_updates_per_cycle * steps / el, "steps": steps, "warmup": warmup}
    if pair_n:
        t = pair_ms / pair_n
        out["dominant_kernel"] = {"kernel": "three fine-level sweeps (+ norm) in one pass" if dim == 2 else "two fine-level sweeps in one pass",
                                  "avg_launch_ms": t, "launches": pair_n,
                                  "achieved_GBs": 3 * esz * N0 / (t * 1e-3) / 1e9, "frac": 3 * esz * N0 / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
    elif one_n:
        t = one_ms / one_n
        out["dominant_kernel"] = {"kernel": "one fine-level sweep", "avg_launch_ms": t, "launches": one_n,
                                  "achieved_GBs": 3 * esz * N0 / (t * 1e-3) / 1e9, "frac": 3 * esz * N0 / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
    cb = cycle_compulsory_bytes(dim, npts, levels, precision)
    gbs = cb / (el / steps) / 1e9
    out["cycle_roofline"] = {"compulsory_bytes": cb, "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS,
                             "bytes_per_fine_unknown": cb / N0,
                             "model": "bytes one cycle of THIS implementation must move (fused passes; fewer passes lower the bytes, not the fraction)",
                             "survey_8d3": survey_cycle_bytes(dim, N0, el / steps)}
    out["history"] = check_history(f"{dim}d_{npts}_{precision}", s.rnorm)
    s.close()
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD job (`python -m torch.distributed.run`, one rank per
    GPU, rendezvous on 127.0.0.1) and relay its output and exit code.  This process has not touched HIP (nothing imported so far
    does) and never will: it only waits -- no exec of, or from, a process that holds the GPU."""
    import socket
    import subprocess
    try:
        import torch.distributed.run  # noqa: F401  (importing the launcher module initialises no device)
    except Exception as e:  # noqa: BLE001
        print(f"bench.py --gpus {n}: torch.distributed.run is not importable ({e}); launch one rank per GPU yourself "
              "(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment)", file=sys.stderr, flush=True)
        return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")          # (the launcher would set it with a warning)
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode          # stdout / stderr are inherited: rank 0's JSON line passes straight through


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--npts", type=int, default=1025)
    ap.add_argument("--levels", type=int, default=0, help="0: down to one unknown")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--precision", choices=["fp64", "mixed"], default="fp64",
                    help="mixed = BASELINE config 5: fp32 smoother sweeps, fp64 residual/correction (1 GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (2-D 4097^2, 3-D 513^3, mixed 1025^3)")
    ap.add_argument("--cpu-npts", type=int, default=257)
    ap.add_argument("--cpu-levels", type=int, default=8)
    ap.add_argument("--cpu-cycles", type=int, default=120)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MG_BENCH_DEVICE"):      # debugging aid: pin every rank to one device ordinal
        local_rank = int(os.environ["MG_BENCH_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    from multigrid_petsc_amd.solver import Solver
    levels = args.levels
    if levels <= 0:     # coarsen down to a single unknown: n_l = (npts-1)/2^l - 1 >= 1
        levels = 0
        while (args.npts - 1) % (2 ** levels) == 0 and (args.npts - 1) // (2 ** levels) - 1 >= 1:
            levels += 1

    dist = None
    comm = None
    transport = "none"
    transport_fallback = False
    if world > 1:
        # watchdog: a transport that hangs (a collective some rank never enters) must end the job with a message instead of
        # sitting in the launcher until the driver's limit; the whole N-GPU run takes seconds
        import threading
        limit = float(os.environ.get("MG_BENCH_WATCHDOG_S", "420"))

        def _watchdog():
            print(f"[bench rank {rank}] no result after {limit:.0f} s -- a collective is stuck (transport "
                  f"{os.environ.get('MG_BENCH_TRANSPORT', 'rccl')}); aborting", file=sys.stderr, flush=True)
            os._exit(4)
        wd = threading.Timer(limit, _watchdog)
        wd.daemon = True
        wd.start()
    if world > 1:
        import torch.distributed as dist   # control plane only (id broadcast, barrier, max-reduce)
        import torch
        from multigrid_petsc_amd.comm import rccl_comm, HostStagedComm, selftest
        from multigrid_petsc_amd.mgk import Mgk
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        requested = os.environ.get("MG_BENCH_TRANSPORT", "rccl")       # rccl (default) | peer | host
        transport = requested

        def all_ok(ok):                     # every rank takes the same decision: MIN over ranks of a status flag
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag[0]) == 1

        def bring_up(make):
            """communicator + first-run gate with the SAME sequence of control-plane collectives on every rank whatever fails where
            (make() itself is collective-safe: rccl_comm / peer_comm ship a status with what they exchange); then one MIN for the
            communicator, one MIN for the gate.  Returns the communicator, or None after saying why on stderr."""
            c_, why = None, ""
            try:
                c_ = make()
            except Exception as e:          # noqa: BLE001
                why = f"communicator: {e}"
                print(f"[bench rank {rank}] {why}", file=sys.stderr, flush=True)
            up = all_ok(c_ is not None)
            if up:
                # first-run gate: rank-coded planes through every hook of the transport, checked on every rank, before anything is timed
                # (a wrong neighbour, plane or element type shows up here, not as a wrong residual); it returns on every rank
                ok = True
                try:
                    m = Mgk(local_rank)
                    try:
                        selftest(c_.handle, m.ctx)
                    finally:
                        m.close()
                except Exception as e:      # noqa: BLE001
                    ok, why = False, f"self-test: {e}"
                    # said BEFORE the agreement below: if that collective never completes, the first mismatch is on record
                    print(f"[bench rank {rank}] transport self-test FAILED: {why}", file=sys.stderr, flush=True)
                up = all_ok(ok)
            if up:
                return c_
            print(f"[bench rank {rank}] {'RCCL' if transport == 'rccl' else transport} transport unusable ({why or 'another rank failed'})", file=sys.stderr, flush=True)
            if c_ is not None:
                c_.close()
            return None

        def make_peer():
            # IPC-mapped mailboxes + flag words, plane copies by the copy engines (include/mg_comm.h)
            from multigrid_petsc_amd.comm import peer_comm
            m = Mgk(local_rank)
            g0, g2 = m.geom(args.dim, args.npts - 2), m.geom(args.dim, min(args.npts - 2, 255))
            m.close()
            return peer_comm(rank, world, local_rank, dist, 8 * g0.plane, 5, 8 * g2.total)

        if transport in ("rccl", "peer"):
            comm = bring_up((lambda: rccl_comm(rank, world, local_rank, dist)) if transport == "rccl" else make_peer)
            if comm is None:
                if os.environ.get("MG_BENCH_ALLOW_FALLBACK", "1") != "1":
                    raise SystemExit(3)
                # the run continues on the host-staged (gloo) transport so that the job still yields a correct line, but the
                # line says so at top level ("transport_fallback": true): it is NOT an RCCL/xGMI number
                transport, transport_fallback = "host", True
        if transport == "host":
            comm = HostStagedComm(rank, world, dist)

    scale = 6.0 / 7.0 if args.dim == 3 else 0.8
    s = Solver(args.dim, args.npts, levels, v=(3, 3), maxiter=args.steps + args.warmup + 1, scale=scale,
               device=local_rank, rank=rank, nranks=world, comm=comm.handle if comm else None,
               precision=args.precision,
               # the peer transport moves planes with the copy engines and one-wave flag kernels: nothing waits for a free CU, so the marching kernels keep
               # the long streams of a single GPU (MG_SLAB_CHUNK still overrides); RCCL's send / recv kernel gets the chunked interiors
               slab_chunk=(0 if (transport == "peer" and not os.environ.get("MG_SLAB_CHUNK")) else -1))
    s.set_rhs_problem()

    def barrier():
        s.sync()                           # device-wide synchronise through the C ABI (all streams)
        if dist is not None:
            dist.barrier()

    s.cycles(args.warmup)                  # untimed warm-up steps
    barrier()
    s.profile(True)
    t0 = time.perf_counter()
    s.cycles(args.steps)                   # EXACTLY K timed steps
    barrier()
    elapsed = time.perf_counter() - t0
    prof_ms, prof_n = s.profile_read()
    pair_ms, pair_n = s.profile_read(1)    # launches that make two sweeps in one pass (fp64, 3-D, whole grids >= 511^3)

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    dof_per_cycle = s.dof_updates_per_cycle
    value = dof_per_cycle * args.steps / elapsed
    n0 = args.npts - 2
    local_unknowns = s.local_unknowns
    rn = s.rnorm
    out = None
    if rank == 0:
        sweep_ms = prof_ms / max(prof_n, 1)
        bytes_per_dof = JACOBI_BYTES_PER_DOF if args.precision == "fp64" else JACOBI_BYTES_PER_DOF / 2
        achieved = bytes_per_dof * local_unknowns / (sweep_ms * 1e-3) / 1e9 if prof_n else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1 and args.npts == 1025 and args.precision == "fp64":
            try:
                traffic = json.load(open(tpath)).get("jacobi_sweep_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        tj = {}
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
            except Exception:
                tj = {}
        if pair_n:
            # dominant kernel of the cycle: k_jacobi2 (two sweeps per pass).  Its roofline is drawn against the traffic it
            # cannot avoid -- read u, read b, write the twice-swept field: 24 B/unknown per launch (SURVEY 8 d3 counts
            # 24 B per SWEEP, 48 B for what this launch does: reported beside it as per_sweep_equivalent).
            t_ms = pair_ms / pair_n
            comp = bytes_per_dof * local_unknowns
            if world > 1:          # on a slab the timed launch covers the interior planes 2 .. nz-3 (the rest waits for the halos)
                nzl = s.level_planes(0)[1]
                comp *= max(nzl - 4, 1) / float(nzl)
            ach = comp / (t_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": f"k_jacobi2{'r' if args.precision == 'fp64' else ''}<{'double' if args.precision == 'fp64' else 'float'},WX>: "
                                              "two fine-level Jacobi sweeps in one pass (temporal blocking)",
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "launches": pair_n, "avg_launch_ms": t_ms, "algorithmic_bytes_per_launch": comp,
                    "traffic": tj.get("jacobi2_hbm_bytes_per_launch") if (world == 1 and args.npts == 1025 and args.precision == "fp64") else None,
                    "traffic_source": "profiles/traffic.json: rocprofv3 PMC passes (FETCH_SIZE x2, WRITE_SIZE) of an earlier run of this kernel, "
                                      f"{tj.get('round', 'round 2')}; a constant, NOT measured in this run",
                    "per_sweep_equivalent": {"bytes_per_launch": 2 * comp, "GB/s": 2 * ach,
                                             "note": f"SURVEY 8(d3) accounting: {bytes_per_dof:g} B per unknown and SWEEP, two sweeps per launch"}}
        else:
            roof = {"bound": "hbm", "kernel": f"k_stencil<{'double' if args.precision == 'fp64' else 'float'},{args.dim},..,MODE_JACOBI> fine-level Jacobi sweep",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                    "launches": prof_n, "avg_launch_ms": sweep_ms if prof_n else None,
                    "algorithmic_bytes_per_launch": bytes_per_dof * local_unknowns,
                    "traffic": traffic}
        out = {
            "metric": "fp64 DOF-updates/sec per V-cycle at 1024^3",
            "value": value, "unit": "DOF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "transport": transport, "transport_fallback": transport_fallback,
            "dtype": "f64" if args.precision == "fp64" else "f32 sweeps + f64 residual/correction", "data": "synthetic",
            "config": {"workload": f"{args.dim}-D {2 * args.dim + 1}-point Poisson, npts={args.npts} "
                                   f"({n0}^{args.dim} unknowns), {levels} levels, V(3,3), Richardson+Jacobi "
                                   f"scale {scale:.6g}, one step = one V-cycle incl. residual norm",
                       "decomposition": f"z-slabs x{world}, halo transport {transport}" if world > 1 else "single GPU",
                       "fine_unknowns": float(n0) ** args.dim,
                       "dof_updates_per_cycle": dof_per_cycle},
            "cycle_unknowns_per_s": float(n0) ** args.dim * args.steps / elapsed,
            "residual_reduction_per_cycle": float((rn[-1] / rn[-1 - args.steps]) ** (1.0 / args.steps)) if len(rn) > args.steps else None,
            "roofline": roof,
        }
        if world == 1:
            cb = cycle_compulsory_bytes(args.dim, args.npts, levels, args.precision)
            gbs = cb / (elapsed / args.steps) / 1e9
            # the whole cycle against the bytes it cannot avoid (per-operation count of SURVEY 8(d3) beside it)
            out["cycle_roofline"] = {"compulsory_bytes": cb, "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS,
                                     "bytes_per_fine_unknown": cb / float(n0) ** args.dim,
                                     "model": "bytes one cycle of THIS implementation must move (fused passes; DESIGN.md section 4)",
                                     "survey_8d3": survey_cycle_bytes(args.dim, float(n0) ** args.dim, elapsed / args.steps)}
            out["history"] = check_history(f"{args.dim}d_{args.npts}_{args.precision}", rn)
    s.close()
    failures = []
    if rank == 0:
        if world == 1 and pair_n and args.dim == 3 and args.precision == "fp64":
            try:     # the plain one-sweep kernel beside it (north_star: >= 70 % of 8 TB/s on the fp64 smoother sweep)
                ps = plain_sweep_probe(local_rank, n0)
                ps["traffic"] = tj.get("jacobi_sweep_hbm_bytes_per_launch") if args.npts == 1025 else None
                out["roofline"]["plain_sweep"] = ps
                achieved = ps["achieved"]
            except Exception as e:   # reporting only
                out["roofline"]["plain_sweep"] = {"error": str(e)}
        if world == 1 and achieved:
            try:
                tri = stream_ceiling(local_rank, local_unknowns)
                tri["sweep_over_triad"] = achieved / tri["GB/s"]
                out["roofline"]["measured_ceiling"] = tri
            except Exception as e:   # reporting only
                out["roofline"]["measured_ceiling"] = {"error": str(e)}
        default_headline = (world == 1 and args.dim == 3 and args.npts == 1025 and args.precision == "fp64")
        if default_headline and not args.no_configs:
            # the other single-GPU BASELINE configurations ride in the same line (configs 2, 3, 5); config 4 is the N > 1 run
            out["configs"] = []
            for name, (d_, np_, pr_) in (("config2", (2, 4097, "fp64")), ("config3", (3, 513, "fp64")), ("config5", (3, 1025, "mixed"))):
                try:
                    c_ = run_config(d_, np_, pr_, 10, 2, local_rank)
                except Exception as e:   # reporting only
                    c_ = {"error": str(e)}
                c_["baseline_config"] = name
                if name == "config2" and not args.no_cpu_baseline:
                    try:     # the CPU column of the one configuration the reference itself can run (reporting only)
                        c_["cpu_baseline"] = cpu_reference_loop(args.cpu_threads)
                    except Exception as e:   # noqa: BLE001
                        c_["cpu_baseline"] = {"value": None, "error": str(e)}
                out["configs"].append(c_)
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:   # the baseline is reporting only; never fail the GPU line for it
                out["cpu_baseline"] = {"value": None, "error": str(e)}
        print(json.dumps(out), flush=True)
        # the line is printed whatever happened; a configuration that crashed or a residual history that left the committed one makes
        # the run FAIL (exit 5) -- only the cpu_baseline leg is reporting-only
        for c_ in out.get("configs", []):
            if "error" in c_:
                failures.append(f"{c_.get('baseline_config')}: {c_['error']}")
            elif c_.get("history") is not None and not c_["history"]["ok"]:
                failures.append(f"{c_.get('baseline_config')}: residual history deviates from tests/golden/bench_history.json ({c_['history']})")
        if out.get("history") is not None and not out["history"]["ok"]:
            failures.append(f"headline: residual history deviates from tests/golden/bench_history.json ({out['history']})")
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failures:
        for f_ in failures:
            print(f"[bench] FAILED: {f_}", file=sys.stderr, flush=True)
        raise SystemExit(5)


if __name__ == "__main__":
    main()
