#!/bin/bash
# kernel trace of the reference's unmodified driver over the drop-in at 4097^2 (run on the GPU box): per-cycle kernel breakdown
# (gpurun_out/trace_refdriver_4097.txt, tools/trace_cycle.py) and the time from one closing norm to the next, first cycle included
# (gpurun_out/trace_refdriver_4097_spans.txt, tools/trace_cycle_spans.py)
set -e
d=$(mktemp -d); cd $d
printf -- "-npts 4097\n-mesh 0\n-iter 1000\n-grids 12\n-levels 12\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > poisson.in
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/trr -- $GRAFT_REPO_ROOT/build/refdriver/poisson > out.txt 2>&1
grep -E "Solver walltime|iterations" out.txt
python3 $GRAFT_REPO_ROOT/tools/trace_cycle.py $(ls $GRAFT_REPO_ROOT/gpurun_out/trr/*/*_results.db) 6 60 > $GRAFT_REPO_ROOT/gpurun_out/trace_refdriver_4097.txt 2>&1; python3 $GRAFT_REPO_ROOT/tools/trace_cycle_spans.py $(ls $GRAFT_REPO_ROOT/gpurun_out/trr/*/*_results.db) > $GRAFT_REPO_ROOT/gpurun_out/trace_refdriver_4097_spans.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trr
