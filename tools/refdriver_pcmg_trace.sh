#!/bin/bash
# kernel trace of -cycle 8 (PCMG) at 4097^2 through the reference's unmodified driver (run on the GPU box): per outer iteration
# (one k_finish_sum per monitored norm) -> gpurun_out/trace_refdriver_pcmg_4097.txt
set -e
d=$(mktemp -d); cd $d
printf -- "-npts 4097\n-mesh 0\n-iter 100\n-grids 12\n-levels 12\n-cycle 8\n-map 2\n-v 3,3\n-moreNorm 0\n-mg_levels_ksp_type richardson\n-mg_levels_pc_type jacobi\n-mg_levels_ksp_max_it 3\n-mg_levels_ksp_richardson_scale 0.8\n" > poisson.in
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/trp -- $GRAFT_REPO_ROOT/build/refdriver/poisson > out.txt 2>&1
grep -E "Solver walltime|iterations" out.txt | tail -2
python3 $GRAFT_REPO_ROOT/tools/trace_cycle.py $(ls $GRAFT_REPO_ROOT/gpurun_out/trp/*/*_results.db) 5 40 > $GRAFT_REPO_ROOT/gpurun_out/trace_refdriver_pcmg_4097.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trp
