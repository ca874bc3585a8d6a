#!/bin/bash
# the reference's unmodified driver over the drop-in: the levels from 63^2 down recorded and run as ONE tail launch (MGPETSC_TAIL=1, default)
# against every level by its own launches (0): its own Solver walltime, best of 5, at 4097^2 / 1025^2 / 257^2, and the counters
for npts in 4097 1025 257; do
lv=0; n=$((npts-1)); while [ $n -ge 2 ]; do lv=$((lv+1)); n=$((n/2)); done
for tail in 1 0; do
d=$(mktemp -d); cd $d
printf -- "-npts $npts\n-mesh 0\n-iter 1000\n-grids $lv\n-levels $lv\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > poisson.in
echo "npts=$npts levels=$lv MGPETSC_TAIL=$tail"
for rep in 1 2 3 4 5; do
MGPETSC_TAIL=$tail MGPETSC_LAZY_STATS=1 /root/repo/build/refdriver/poisson > out.txt 2>&1
grep -E "Solver walltime" out.txt
done
grep -E "Number of iterations|error\[0\]" out.txt
grep -oE "[0-9]+ coarse sub-cycles.*" out.txt
done
done
