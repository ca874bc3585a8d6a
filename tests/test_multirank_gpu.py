"""Multi-rank (z-slab) V-cycle on ONE GPU: every rank is a thread with its own context and streams,
halos move through the loopback communicator (device-to-device copies).  This runs every line of the
slab logic (nested split, halo protocol, replicated coarse levels, all-gather, norm all-reduce) that the
RCCL back end runs on 8 GPUs; only the transport differs.  Bar: the P-rank result equals the 1-rank
result -- fields bit-identical (Jacobi is order independent), norms to 1e-13 (summation order)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve_single(npts, levels, scale, maxiter, **kw):
    from multigrid_petsc_amd.solver import Solver
    s = Solver(3, npts, levels, scale=scale, maxiter=maxiter, **kw)
    s.set_rhs_problem()
    it = s.solve()
    out = (it, s.rnorm, s.solution(), s.error_norms())
    s.close()
    return out


def _solve_ranks(P, npts, levels, scale, maxiter, dist_min_n, **kw):
    from multigrid_petsc_amd.solver import Solver
    from multigrid_petsc_amd.comm import LoopbackWorld
    world = LoopbackWorld(P)

    def fn(rank, comm):
        s = Solver(3, npts, levels, scale=scale, maxiter=maxiter, rank=rank, nranks=P, comm=comm,
                   dist_min_n=dist_min_n, **kw)
        s.set_rhs_problem()
        it = s.solve()
        res = (it, s.rnorm, s.solution(), s.error_norms(), [s.level_planes(l) for l in range(levels)])
        s.close()
        return res

    try:
        return world.run(fn)
    finally:
        world.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("P,npts,levels,dist_min_n", [
    (2, 33, 4, 15),      # levels 31,15 distributed; 7,3 replicated
    (3, 65, 5, 15),      # uneven split, three distributed levels
    (4, 65, 6, 31),      # two distributed levels, four replicated
    (8, 129, 6, 31),     # 8 slabs as on the node: 127, 63, 31 distributed
    (2, 65, 3, 15),      # every level distributed (no replicated level)
])
def test_slab_ranks_equal_single_rank(P, npts, levels, dist_min_n):
    scale, maxiter = 6.0 / 7.0, 60
    it1, rn1, u1, e1 = _solve_single(npts, levels, scale, maxiter)
    res = _solve_ranks(P, npts, levels, scale, maxiter, dist_min_n)
    n = npts - 2
    planes = [r[4][0] for r in res]
    assert planes[0][0] == 0 and sum(p[1] for p in planes) == n
    for r in res:
        assert r[0] == it1
        assert np.allclose(r[1], rn1, rtol=1e-13, atol=0)
        assert r[3][0] == e1[0] and np.allclose(r[3][1:], e1[1:], rtol=1e-12, atol=0)
    u = np.concatenate([r[2] for r in res])
    assert u.size == n ** 3
    assert np.array_equal(u, u1), f"max diff {np.abs(u - u1).max()}"


@pytest.mark.timeout(300)
def test_overlapped_and_blocking_halo_agree():
    """overlap=1: boundary planes -> comm-stream exchange while the interior sweeps; overlap=0: exchange, then sweep"""
    a = _solve_ranks(4, 65, 5, 6.0 / 7.0, 40, 15, overlap=1)
    b = _solve_ranks(4, 65, 5, 6.0 / 7.0, 40, 15, overlap=0)
    for ra, rb in zip(a, b):
        # fields bit for bit; the norms to rounding (overlap=1 reduces the block partials of three launches per norm)
        assert ra[0] == rb[0] and np.allclose(ra[1], rb[1], rtol=1e-13, atol=0) and np.array_equal(ra[2], rb[2])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("P,npts,levels,dist_min_n", [(2, 33, 5, 15), (4, 65, 6, 15), (3, 65, 4, 31)])
def test_slab_ranks_mixed_precision(P, npts, levels, dist_min_n):
    """BASELINE config 5 on slabs: fp32 halos (4-byte planes) inside, fp64 residual/correction outside"""
    it1, rn1, u1, e1 = _solve_single(npts, levels, 6.0 / 7.0, 60, precision="mixed")
    res = _solve_ranks(P, npts, levels, 6.0 / 7.0, 60, dist_min_n, precision="mixed")
    for r in res:
        assert r[0] == it1
        assert np.allclose(r[1], rn1, rtol=1e-13, atol=0)
    assert np.array_equal(np.concatenate([r[2] for r in res]), u1)


@pytest.mark.timeout(300)
def test_slab_ranks_chebyshev(mgk):
    it1, rn1, u1, _ = _solve_single(65, 5, 1.0, 40, ksp_type="chebyshev", eigenvalues=(0.2, 2.0))
    res = _solve_ranks(4, 65, 5, 1.0, 40, 15, ksp_type="chebyshev", eigenvalues=(0.2, 2.0))
    assert all(r[0] == it1 for r in res)
    assert np.array_equal(np.concatenate([r[2] for r in res]), u1)


@pytest.mark.timeout(120)
def test_rccl_backend_single_rank_plumbing(mgk):
    """dlopen(librccl) + ncclCommInitRank with the 128-byte id passed by value + a 1-rank all-reduce.
    (Two or more ranks need one GPU each; the driver's 8-GPU run covers that.)"""
    from multigrid_petsc_amd.comm import rccl_comm, _lib
    c = rccl_comm(0, 1, 0)
    v = np.array([1.5, -2.25, 3.0])
    rc = _lib().mg_comm_allreduce_sum(c.handle, mgk.ctx, v.ctypes.data, 3)
    assert rc == 0, _lib().mg_comm_last_error()
    assert list(v) == [1.5, -2.25, 3.0]
    g = mgk.geom(3, 7)
    f = mgk.field(g)
    assert _lib().mg_comm_halo(c.handle, mgk.ctx, f, C.byref(g)) == 0     # no neighbours: no-op
    mgk.free(f)
    # point-to-point entry points, this rank as its own peer: the call shape of the halo exchange
    L = _lib()
    L.mg_comm_rccl_self_sendrecv.restype = C.c_int
    L.mg_comm_rccl_self_sendrecv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int]
    src = np.random.default_rng(5).uniform(-1, 1, 4096)
    for esz, arr in ((8, src), (4, src.astype(np.float32))):
        a = mgk.alloc(arr.nbytes)
        b = mgk.alloc(arr.nbytes)
        mgk._chk(mgk.L.mgk_h2d(mgk.ctx, a, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        mgk._chk(mgk.L.mgk_memset0(mgk.ctx, b, arr.nbytes, None))
        rc = L.mg_comm_rccl_self_sendrecv(c.handle, mgk.ctx, a, b, arr.size, esz)
        assert rc == 0, L.mg_comm_last_error()
        out = np.empty_like(arr)
        mgk._chk(mgk.L.mgk_d2h(mgk.ctx, out.ctypes.data_as(C.c_void_p), b, arr.nbytes))
        assert np.array_equal(out, arr)
        mgk.free(a)
        mgk.free(b)
    c.close()


@pytest.mark.timeout(240)
def test_rccl_two_ranks_on_one_gpu_if_allowed():
    """Try the real RCCL send/recv path with two processes sharing GPU 0.  RCCL normally rejects two ranks
    on one device ("duplicate GPU"); if it does, the test is skipped and says so."""
    script = os.path.join(ROOT, "tests", "rccl_pair.py")
    idfile = os.path.join("/tmp", f"mg_rccl_id_{os.getpid()}")
    if os.path.exists(idfile):
        os.remove(idfile)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", idfile], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=200)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o += "\n[timeout]"
        outs.append(o)
    # a refusal (ncclCommInitRank says no: two ranks on one device) is a skip; anything that goes wrong AFTER a communicator
    # came up -- a hang in the send/recv halo, a self-test mismatch -- is a failure of the transport
    if not any("COMM_UP" in o for o in outs):
        if all("RCCL_REFUSED" in o or "[timeout]" in o for o in outs) and any("RCCL_REFUSED" in o for o in outs):
            pytest.skip("RCCL does not run two ranks on one GPU here: " + " | ".join(o.strip().splitlines()[-1] for o in outs if o.strip()))
    for p, o in zip(procs, outs):
        assert "[timeout]" not in o, "hang after the communicator came up:\n" + o
        assert p.returncode == 0 and "SELFTEST_OK" in o and "PAIR_OK" in o, o


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,npts,levels,dmin", [(2, 65, 5, 15), (3, 65, 5, 15)])
def test_peer_transport_between_processes_on_one_gpu(tmp_path, world, npts, levels, dmin):
    """The peer transport (round 3: IPC-mapped mailboxes + flag words, one-wave flag kernels, plane copies by the runtime) between real
    PROCESSES that share GPU 0: hipIpc handles of fine-grained memory, cross-process flag words, the first-run gate, then the slab solve --
    iteration count and history of every rank equal the single rank's, the concatenated slabs equal its solution bit for bit, and the
    fixed-count loop with device all-reduces gives the same history.  (Between GPUs only the copies differ: copy engines over xGMI --
    unmeasured here.)"""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = str(sk.getsockname()[1])
    sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MG_PEER_TIMEOUT_S="30")
    script = os.path.join(ROOT, "tests", "peer_pair.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), port, str(npts), str(levels), str(dmin), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, _ = p.communicate()
            o += "\n[timeout]"
        outs.append(o)
    for p, o in zip(procs, outs):
        assert "[timeout]" not in o and p.returncode == 0 and "SELFTEST_OK" in o and "PAIR_OK" in o, o[-3000:]
    it1, rn1, u1, e1 = _solve_single(npts, levels, 6.0 / 7.0, 60)
    parts = [np.load(tmp_path / f"peer_rank{r}.npz") for r in range(world)]
    for p in parts:
        assert int(p["it"]) == it1
        assert np.abs(p["rn"] / rn1 - 1).max() <= 1e-12
        assert np.abs(p["rn3"] / rn1[:4] - 1).max() <= 1e-12
        assert np.array_equal(p["e"], parts[0]["e"])
    assert np.array_equal(np.concatenate([p["u"] for p in parts]), u1)


@pytest.mark.timeout(900)
def test_bench_two_processes_host_staged_transport(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), here with both
    ranks pinned to GPU 0 and the host-staged halo transport (RCCL refuses two ranks on one device): exercises the
    whole multi-process path -- rendezvous, slab solver per process, halos through the mg_comm hooks, max-over-ranks
    timing, rank-0 JSON -- and must reproduce the single-process numerics."""
    import json
    env = dict(os.environ, MG_BENCH_DEVICE="0", MG_BENCH_TRANSPORT="host", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "4", "--warmup", "1", "--npts", "129", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, bench, "--gpus", "1"] + common, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", bench, "--gpus", "2"] + common, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=800)
    assert two.returncode == 0, (two.stdout[-1500:], two.stderr[-3000:])
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and "host" in j2["config"]["decomposition"]
    assert j2["config"]["dof_updates_per_cycle"] == j1["config"]["dof_updates_per_cycle"]
    assert abs(j2["residual_reduction_per_cycle"] - j1["residual_reduction_per_cycle"]) <= 1e-12


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` exactly as the driver types it for N = 1 -- NO `-m torch.distributed.run` in front: bench.py starts the
    launcher itself as a child process (the parent never touches HIP), and ONE JSON line with n_gpus = 2 comes back through it, exit
    code 0.  Both ranks share GPU 0 here (host-staged transport: RCCL refuses two ranks on one device)."""
    import json
    env = dict(os.environ, MG_BENCH_DEVICE="0", MG_BENCH_TRANSPORT="host", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--npts", "129",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["transport"] == "host" and j["transport_fallback"] is False
    assert j["value"] > 0 and j["scaling"] == "strong"


@pytest.mark.timeout(900)
def test_bench_two_processes_peer_transport():
    """`MG_BENCH_TRANSPORT=peer python bench.py --gpus 2` (self-launched ranks, both on GPU 0): the peer transport passes the first-run
    gate and carries the run -- one JSON line, transport "peer", no fallback, the residual reduction of the single-process run"""
    import json
    env = dict(os.environ, MG_BENCH_DEVICE="0", MG_BENCH_TRANSPORT="peer", MG_BENCH_ALLOW_FALLBACK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", MG_PEER_TIMEOUT_S="30")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "4", "--warmup", "1", "--npts", "129", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, bench, "--gpus", "1"] + common, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, bench, "--gpus", "2"] + common, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert two.returncode == 0, (two.stdout[-1500:], two.stderr[-3000:])
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["transport"] == "peer" and j2["transport_fallback"] is False
    assert abs(j2["residual_reduction_per_cycle"] - j1["residual_reduction_per_cycle"]) <= 1e-12


@pytest.mark.timeout(300)
def test_fixed_count_cycling_defers_the_norms_on_slabs():
    """mg_solver_cycles (bench.py's loop): the per-cycle sums of squares stay on the device and are all-reduced once at the
    end; history and fields equal the convergence-driven loop of a single rank"""
    from multigrid_petsc_amd.solver import Solver
    from multigrid_petsc_amd.comm import LoopbackWorld
    it1, rn1, u1, _ = _solve_single(65, 5, 6.0 / 7.0, 40)
    world = LoopbackWorld(3)

    def fn(rank, comm):
        s = Solver(3, 65, 5, scale=6.0 / 7.0, maxiter=40, rank=rank, nranks=3, comm=comm, dist_min_n=15)
        s.set_rhs_problem()
        s.cycles(2)
        s.cycles(it1 - 2)
        s.sync()
        out = (s.rnorm.copy(), s.solution())
        s.close()
        return out

    try:
        res = world.run(fn)
    finally:
        world.close()
    for rn, _ in res:
        assert rn.shape == rn1.shape and np.abs(rn / rn1 - 1).max() <= 1e-13
    assert np.array_equal(np.concatenate([r[1] for r in res]), u1)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("P,npts,levels,dist_min_n,precision", [
    (2, 65, 5, 15, "fp64"), (3, 65, 5, 15, "fp64"), (4, 129, 6, 31, "fp64"), (8, 129, 6, 63, "fp64"), (2, 65, 3, 15, "fp64"),
    (2, 65, 5, 15, "mixed"), (4, 129, 6, 31, "mixed"),
    (8, 129, 6, 31, "fp64"),      # level 31: slabs of 4,4,...,3 planes -- every rank must take the same (one-sweep) path there
    (8, 65, 4, 15, "fp64"),       # slabs of 2 planes and 3 on the last rank
])
def test_two_sweep_passes_on_slabs(P, npts, levels, dist_min_n, precision):
    """fuse bit 5 on distributed levels: the pair kernel runs on every slab with the neighbours' two boundary planes of u
    (regular ghost plane + the far-plane field) and their b ghost plane; fields equal the single-rank, one-sweep-per-launch
    solve bit for bit"""
    it1, rn1, u1, _ = _solve_single(npts, levels, 6.0 / 7.0, 60, precision=precision, fuse=0)
    res = _solve_ranks(P, npts, levels, 6.0 / 7.0, 60, dist_min_n, precision=precision, pair_min_n=15, fuse=127)
    assert all(r[0] == it1 for r in res)
    for r in res:
        assert np.abs(r[1] / rn1 - 1).max() <= 1e-12
    assert np.array_equal(np.concatenate([r[2] for r in res]), u1)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("P", [1, 2, 3, 8])
def test_transport_selftest_on_loopback_ranks(P):
    """mg_comm_selftest -- the first-run gate bench.py applies to the RCCL communicator before it times anything -- on the
    loopback transport: rank-coded planes through halo / halo_n / allgather_planes (fp64 and fp32), both all-reduce forms"""
    from multigrid_petsc_amd.comm import LoopbackWorld, selftest
    from multigrid_petsc_amd.mgk import Mgk
    world = LoopbackWorld(P)

    def fn(rank, comm):
        m = Mgk(0)
        try:
            selftest(comm, m.ctx)
        finally:
            m.close()
        return True

    try:
        assert all(world.run(fn))
    finally:
        world.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("P,bad", [(2, 1), (8, 3)])
def test_transport_selftest_returns_on_every_rank_when_a_plane_is_wrong(P, bad):
    """the gate is collective-safe on FAILURE: with a transport that hands rank `bad` a wrong lo ghost plane every rank returns from
    mg_comm_selftest (none is left inside a collective -- this test would time out), `bad` with the first mismatch named, the rest 0"""
    from multigrid_petsc_amd.comm import LoopbackWorld, _lib
    from multigrid_petsc_amd.mgk import Mgk
    world = LoopbackWorld(P)
    world.inject_fault(bad)

    def fn(rank, comm):
        m = Mgk(0)
        try:
            rc = _lib().mg_comm_selftest(comm, m.ctx)
            return rc, _lib().mg_comm_last_error().decode()
        finally:
            m.close()

    try:
        res = world.run(fn)
    finally:
        world.close()
    for r, (rc, msg) in enumerate(res):
        if r == bad:
            assert rc != 0 and "halo plane" in msg and f"rank {bad}" in msg, (rc, msg)
        else:
            assert rc == 0, (r, rc, msg)


@pytest.mark.timeout(120)
def test_rccl_backend_refuses_a_second_stream(mgk):
    """one communicator, one stream: the RCCL hooks accept the context's comm stream (or NULL = that stream) only"""
    from multigrid_petsc_amd.comm import rccl_comm, _lib, _MgComm, selftest
    c = rccl_comm(0, 1, 0)
    selftest(c.handle, mgk.ctx)                      # 1 rank: every hook runs, nothing moves
    st = _MgComm.from_address(c.handle.value)
    v = (C.c_double * 2)(1.0, 2.0)
    cs, ms = mgk.L.mgk_stream_compute(mgk.ctx), mgk.L.mgk_stream_comm(mgk.ctx)
    assert st.allreduce_sum(c.handle, mgk.ctx, v, 2, ms) == 0 and list(v) == [1.0, 2.0]
    assert st.allreduce_sum(c.handle, mgk.ctx, v, 2, cs) != 0
    assert b"comm stream" in _lib().mg_comm_last_error()
    c.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("rank", [0, 3, 7])
def test_phantom_rank_runs_the_slab_cycle(rank):
    """the phantom communicator (timing aid: one rank of 8 alone on the GPU, neighbours' planes stood in for by copies) drives
    the slab code path of that rank -- the launches, streams and events of the 8-GPU cycle -- without error; the numbers are
    meaningless by construction and only checked to be finite"""
    from multigrid_petsc_amd.solver import Solver
    from multigrid_petsc_amd.comm import phantom_comm
    c = phantom_comm(rank, 8, lat_us=5.0, link_gbs=50.0)
    s = Solver(3, 129, 6, scale=6.0 / 7.0, maxiter=10, rank=rank, nranks=8, comm=c.handle, dist_min_n=31)
    s.set_rhs_problem()
    s.cycles(3)
    s.sync()
    assert np.all(np.isfinite(s.rnorm)) and s.iterations == 3
    s.close()
    c.close()


@pytest.mark.timeout(900)
def test_full_size_eight_slabs_equal_the_single_rank_cycle():
    """BASELINE config 4 at its own size: 1023^3 cut into the 8 z-slabs of the 8-GPU job (loopback ranks share this one GPU;
    every kernel, stream, event and exchange of the slab path runs, only the wire differs) for 3 V-cycles, against the
    single-rank run: residual history to rounding, and the field through the three GetError sums (max |e| identical --
    it is one element of the field -- the two sums to summation order)."""
    from multigrid_petsc_amd.solver import Solver
    from multigrid_petsc_amd.comm import LoopbackWorld
    s = Solver(3, 1025, 10, scale=6.0 / 7.0, maxiter=8)
    s.set_rhs_problem()
    s.cycles(3)
    s.sync()
    rn1, e1 = s.rnorm.copy(), s.error_norms()
    s.close()
    world = LoopbackWorld(8)

    def fn(rank, comm):
        r = Solver(3, 1025, 10, scale=6.0 / 7.0, maxiter=8, rank=rank, nranks=8, comm=comm)
        r.set_rhs_problem()
        r.cycles(3)
        r.sync()
        out = (r.rnorm.copy(), r.error_norms(), [r.level_planes(l) for l in range(10)])
        r.close()
        return out

    try:
        res = world.run(fn)
    finally:
        world.close()
    assert sum(r[2][0][1] for r in res) == 1023 and all(r[2][3][1] == 127 for r in res)      # levels 0-2 cut, 127^3 replicated
    for rn, e, _ in res:
        assert np.abs(rn / rn1 - 1).max() <= 1e-12
        assert e[0] == e1[0]
        assert abs(e[1] - e1[1]) <= 1e-12 * e1[1] and abs(e[2] - e1[2]) <= 1e-12 * e1[2]


@pytest.mark.timeout(900)
def test_bench_never_falls_back_silently(tmp_path):
    """bench.py with N = 2 and the default (RCCL) transport on ONE GPU: RCCL cannot give two ranks one device, so either the
    communicator or the first-run gate fails on every rank.  The run must then (a) end with exit code 3 when fallbacks are
    forbidden, (b) otherwise finish on the host-staged transport AND say so at the top level of its JSON line.  If RCCL does
    run two ranks on this device the line must say transport rccl with no fallback."""
    import json
    env = dict(os.environ, MG_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench = os.path.join(ROOT, "bench.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", bench, "--gpus", "2", "--steps", "3", "--warmup", "1", "--npts", "65", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2
    if j["transport_fallback"]:
        assert j["transport"] == "host" and "RCCL transport unusable" in p.stderr
        cmd[cmd.index("29541")] = "29543"
        q = subprocess.run(cmd, env=dict(env, MG_BENCH_ALLOW_FALLBACK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert q.returncode != 0 and not [l for l in q.stdout.strip().splitlines() if l.startswith("{")]
    else:
        assert j["transport"] == "rccl"
