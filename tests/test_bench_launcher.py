"""bench.py's self-launch path (`python bench.py --gpus N` without torch.distributed.run in front), CPU tier: the parent must
start the launcher as a CHILD with the driver's own arguments, relay its exit code, and must not import anything that could
initialise HIP before doing so."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_builds_the_drivers_command(monkeypatch):
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"], seen["kw"] = cmd, env, kw
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
        code = 0
    except SystemExit as e:
        code = e.code
    assert code == 7                                            # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "stdout" not in seen["kw"] and "stderr" not in seen["kw"]          # inherited: rank 0's JSON line passes through


def test_parent_touches_no_gpu_module_before_the_launch():
    """up to the self-launch, bench.py has imported nothing of the product (whose libraries initialise HIP) and no torch.cuda state"""
    code = ("import sys, subprocess, importlib.util\n"
            "subprocess.run = lambda *a, **k: subprocess.CompletedProcess(a, 0)\n"
            f"spec = importlib.util.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r})\n"
            "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
            "sys.argv = ['bench.py', '--gpus', '2']\n"
            "try:\n    m.main()\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "bad = [k for k in sys.modules if k.startswith('multigrid_petsc_amd') or k == 'oracle']\n"
            "assert not bad, bad\n"
            "import torch\nassert not torch.cuda.is_initialized()\nprint('CLEAN')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "CLEAN" in r.stdout, r.stderr[-2000:]


def test_launched_ranks_do_not_launch_again(monkeypatch):
    """a rank started by the launcher (WORLD_SIZE set) goes straight on; with WORLD_SIZE != --gpus the launcher's count wins"""
    bench = _load_bench()
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda n: called.append(n) or 0)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--npts", "17"])
    # the run itself needs a GPU: stop at the first product import (and keep the N-rank watchdog, which would end THIS process, unarmed)
    import builtins
    import threading

    class _NoTimer:
        daemon = True

        def __init__(self, *a, **k):
            pass

        def start(self):
            pass
    monkeypatch.setattr(threading, "Timer", _NoTimer)
    real_import = builtins.__import__

    def guard(name, *a, **k):
        if name.startswith("multigrid_petsc_amd") or name == "torch.distributed":
            raise RuntimeError("reached the solver")
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", guard)
    try:
        bench.main()
    except RuntimeError as e:
        assert "reached the solver" in str(e)
    assert called == []
