"""-cycle 1 with several grids in one level (the I-cycle's coupled operator, src/solver.c:255-487) through the reference's UNMODIFIED
driver over the drop-in: seconds per Richardson iteration (sweep + residual + norm) with the block operator recognised (stencil /
transfer kernels on [fine | coarse] vectors) and with recognition off (assembled AIJ on the generic CSR kernel).
Usage: python tools/bench_icycle.py [npts] [iters] [grids]   (needs build/refdriver/poisson from __graft_entry__.build())"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
grids = int(sys.argv[3]) if len(sys.argv) > 3 else 2
opts = (f"-npts {npts}\n-mesh 0\n-iter {iters}\n-grids {grids}\n-levels 1\n-cycle 1\n-map 2\n-v 3,3\n-moreNorm 0\n"
        "-pc_type jacobi\n-ksp_richardson_scale 0.3\n")
for tag, env in (("recognised", {}), ("generic CSR", {"MGPETSC_NO_RECOGNITION": "1"})):
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "poisson.in"), "w").write(opts)
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([os.path.join(ROOT, "build", "refdriver", "poisson")], cwd=d, env=e, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=900)
        out = p.stdout
        wall = [float(x) for x in re.findall(r"Solver walltime:\s+([0-9.eE+-]+)", out)]
        kind = re.search(r"device operator: ([^\n]*)", out)
        its = re.search(r"Number of iterations:\s+(\d+)", out)
        print(f"npts={npts} grids={grids} iters={iters} {tag}: rc={p.returncode} walltime={wall} iterations={its.group(1) if its else '?'} operator={kind.group(1) if kind else '?'}", flush=True)
        if p.returncode:
            print(out[-1500:])
