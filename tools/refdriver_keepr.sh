#!/bin/bash
# the reference's unmodified driver over the drop-in at 4097^2: the closing norm pass with r left deferred (MGPETSC_KEEP_R=1, default) against
# r stored by that pass (0): its own Solver walltime, best of 5, and the lazy-temporary counters
for keep in 1 0; do
d=$(mktemp -d); cd $d
printf -- "-npts 4097\n-mesh 0\n-iter 1000\n-grids 12\n-levels 12\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > poisson.in
echo "MGPETSC_KEEP_R=$keep"
for rep in 1 2 3 4 5; do
MGPETSC_KEEP_R=$keep MGPETSC_LAZY_STATS=1 /root/repo/build/refdriver/poisson > out.txt 2>&1
grep -E "Solver walltime" out.txt
done
grep -E "Number of iterations|error\[0\]|lazy temporaries" out.txt
done
