#!/bin/bash
# -cycle 8 (PCMG) at 4097^2 through the reference's unmodified driver: Richardson + Jacobi level smoothers (3 sweeps), PETSc's default coarse
# solver (preonly + LU: exact), lazy temporaries on / off; best of 3 runs of its own Solver walltime
for lazy in 1 0; do
d=$(mktemp -d); cd $d
printf -- "-npts 4097\n-mesh 0\n-iter 100\n-grids 12\n-levels 12\n-cycle 8\n-map 2\n-v 3,3\n-moreNorm 0\n-mg_levels_ksp_type richardson\n-mg_levels_pc_type jacobi\n-mg_levels_ksp_max_it 3\n-mg_levels_ksp_richardson_scale 0.8\n" > poisson.in
echo "MGPETSC_LAZY=$lazy"
for rep in 1 2 3; do
MGPETSC_LAZY=$lazy MGPETSC_LAZY_STATS=1 /root/repo/build/refdriver/poisson > out.txt 2>&1
grep -E "Solver walltime" out.txt
done
grep -E "Number of iterations|error\[0\]|lazy temporaries" out.txt | cut -c 1-260
done
