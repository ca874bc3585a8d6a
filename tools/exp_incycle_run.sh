#!/bin/bash
# runs tools/exp_incycle_2d.py in the modes named below under rocprofv3 --kernel-trace --stats (one process per mode) and keeps the rows of
# k_jacobi3_2d from each kernel_stats.csv as gpurun_out/ex3_<mode>.txt: the read-after-write / store-policy experiment of DESIGN.md 4 (xv)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for md in chain_plain chain_st alone_st rot3; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ex_$md -- python3 tools/exp_incycle_2d.py $md rand > gpurun_out/ex.log 2>&1 || exit 1
f=$(find gpurun_out/ex_$md -name "*kernel_stats.csv" | head -1); grep jacobi3 $f > gpurun_out/ex3_$md.txt; rm -rf gpurun_out/ex_$md
done
