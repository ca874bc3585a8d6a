#!/bin/bash
# the reference's unmodified driver over the drop-in at 4097^2 (debugging aid / BASELINE.md note): prints its own Solver walltime
d=$(mktemp -d); cd $d
printf -- "-npts 4097\n-mesh 0\n-iter 1000\n-grids 12\n-levels 12\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > poisson.in
/root/repo/build/refdriver/poisson > out.txt 2>&1
grep -E "Solver walltime|Number of iterations|error\[0\]" out.txt
