#!/bin/bash
# whole-process time of the reference driver over the drop-in at 4097^2 (set-up included: the reference assembles 83 M entries through MatSetValue)
d=$(mktemp -d); cd $d
printf -- "-npts 4097\n-mesh 0\n-iter 1000\n-grids 12\n-levels 12\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > poisson.in
t0=$(date +%s.%N); /root/repo/build/refdriver/poisson > out.txt 2>&1; t1=$(date +%s.%N)
grep -E "Solver walltime|iterations" out.txt | tail -2; python3 -c "print(\"whole process: %.2f s\" % ($t1 - $t0))"
