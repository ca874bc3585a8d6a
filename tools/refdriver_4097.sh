#!/bin/bash
# the reference's unmodified driver over the drop-in at 4097^2 (BASELINE.md / DESIGN.md 8b N2): prints its own Solver walltime (best of 3
# runs per mode; 9 cycles each) with the lazy temporaries of the drop-in on (1, default), without the deferred restriction (2), and off (0:
# every PETSc call executed at once)
for lazy in 1 2 0; do
d=$(mktemp -d); cd $d
printf -- "-npts 4097\n-mesh 0\n-iter 1000\n-grids 12\n-levels 12\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > poisson.in
echo "MGPETSC_LAZY=$lazy"
for rep in 1 2 3; do
MGPETSC_LAZY=$lazy MGPETSC_LAZY_STATS=1 /root/repo/build/refdriver/poisson > out.txt 2>&1
grep -E "Solver walltime" out.txt
done
grep -E "Number of iterations|error\[0\]|lazy temporaries" out.txt
done
