cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for md in chain_plain chain_st alone_st rot3; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ex_$md -- python3 tools/exp_incycle_2d.py $md rand > gpurun_out/ex.log 2>&1 || exit 1
f=$(find gpurun_out/ex_$md -name "*kernel_stats.csv" | head -1); grep jacobi3 $f > gpurun_out/ex3_$md.txt; rm -rf gpurun_out/ex_$md
done
