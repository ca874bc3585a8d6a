#!/bin/bash
# the reference's unmodified driver over the drop-in at 4097^2 (BASELINE.md / DESIGN.md 8b N2): prints its own Solver walltime, with the lazy
# temporaries of the drop-in on (default) and off (every PETSc call executed at once)
for lazy in 1 0; do
d=$(mktemp -d); cd $d
printf -- "-npts 4097\n-mesh 0\n-iter 1000\n-grids 12\n-levels 12\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > poisson.in
MGPETSC_LAZY=$lazy MGPETSC_LAZY_STATS=1 /root/repo/build/refdriver/poisson > out.txt 2>&1
echo "MGPETSC_LAZY=$lazy"; grep -E "Solver walltime|Number of iterations|error\[0\]|lazy temporaries" out.txt
done
