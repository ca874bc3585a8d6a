#!/bin/bash
# the profile set of a round (run on the GPU box through gpurun; outputs under gpurun_out/prof/): kernel stats of the headline bench run, the
# PMC traffic passes (3-D fine-level kernels at 1023^3, 2-D at 4095^2), kernel stats of configs 2 and 3, the plain bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stats() {   # name, command...
    name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$name -- "$@" > $O/${name}_stdout.txt 2> $O/${name}_stderr.txt
    f=$(find $O/raw_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
    rm -rf $O/raw_$name
}
pmc() {     # name, counter, script...
    name=$1; ctr=$2; shift 2
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/raw_$name -- "$@" > /dev/null 2> $O/${name}_stderr.txt
    python3 $R/tools/pmc_summary.py $O/raw_$name > $O/$name.csv
    rm -rf $O/raw_$name
}
cd $R
stats bench_1023 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs
echo "bench_1023 done"
pmc pmc_fetch_size_3d FETCH_SIZE python3 $R/tools/pmc_sweep.py
pmc pmc_write_size_3d WRITE_SIZE python3 $R/tools/pmc_sweep.py
echo "pmc 3d done"
pmc pmc_fetch_size_2d FETCH_SIZE python3 $R/tools/pmc_sweep_2d.py
pmc pmc_write_size_2d WRITE_SIZE python3 $R/tools/pmc_sweep_2d.py
echo "pmc 2d done"
stats config2_4097 python3 $R/tools/trace_config.py 2 4097
stats config3_513 python3 $R/tools/trace_config.py 3 513
echo "configs done"
python3 $R/tools/pmc_traffic.py $O/pmc_fetch_size_3d.csv $O/pmc_write_size_3d.csv 1070599167 > $O/traffic_3d.json
python3 $R/tools/pmc_traffic.py $O/pmc_fetch_size_2d.csv $O/pmc_write_size_2d.csv 16769025 > $O/traffic_2d.json
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20_stderr.txt
echo "bench rc=$?"
ls $O
