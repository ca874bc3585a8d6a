#!/bin/bash
# -mesh 1 at 4097^2 (694 cycles to 1e-7): the reference's unmodified driver over the drop-in, with its lazy temporaries on and off, and the own driver
d=$(mktemp -d); cd $d; mkdir a a0 b
printf -- "-npts 4097\n-mesh 1\n-iter 1000\n-grids 12\n-levels 12\n-cycle 0\n-map 2\n-v 3,3\n-moreNorm 0\n-pc_type jacobi\n-ksp_richardson_scale 0.8\n" > a/poisson.in
cp a/poisson.in a0/; cp a/poisson.in b/
echo "---- reference driver over the drop-in (MGPETSC_LAZY=1, default)"
(cd a && MGPETSC_LAZY_STATS=1 /root/repo/build/refdriver/poisson > out.txt 2>&1; grep -E "Solver walltime|Number of iterations|error\[0\]|Relative residual|lazy temporaries" out.txt | cut -c 1-200)
echo "---- reference driver over the drop-in, every call executed at once (MGPETSC_LAZY=0)"
(cd a0 && MGPETSC_LAZY=0 /root/repo/build/refdriver/poisson > out.txt 2>&1; grep -E "Solver walltime|Number of iterations|error\[0\]|Relative residual" out.txt | cut -c 1-120)
echo "---- own driver"
(cd b && /root/repo/multigrid_petsc_amd/mgpoisson -dim 2 -write_fields 0 > out.txt 2>&1; grep -E "Solver walltime|Number of iterations|error\[0\]|Relative residual" out.txt | cut -c 1-120)
cmp -s a/uData.dat a0/uData.dat && echo "uData.dat of the two reference-driver runs identical" || echo "uData.dat of the two reference-driver runs DIFFER"
cmp -s a/rData.dat b/rData.dat && echo "rData.dat identical" || echo "rData.dat: equal up to the summation order of the norm"
