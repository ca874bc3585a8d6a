#!/bin/bash
# kernel trace of rank 3 of 8 (phantom neighbours) at 1023^3 with the free and the expected link model (run on the GPU box): per-cycle kernel breakdown
# by tools/trace_slab_cycle.py -> gpurun_out/trace_slab_rank3_{free,expected}.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in "free 0 0" "expected 20 60"; do
    set -- $m
    rm -rf $R/gpurun_out/trs
    rocprofv3 --kernel-trace -d $R/gpurun_out/trs -- python3 $R/tools/trace_slab.py $2 $3 > /dev/null 2>&1
    python3 $R/tools/trace_slab_cycle.py $(ls $R/gpurun_out/trs/*/*_results.db) > $R/gpurun_out/trace_slab_rank3_$1.txt 2>&1
    rm -rf $R/gpurun_out/trs
    head -3 $R/gpurun_out/trace_slab_rank3_$1.txt
done
