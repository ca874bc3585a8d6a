#!/bin/bash
# VGPR / SGPR / LDS / scratch of the gfx950 kernels in libmgk.so whose name matches $1 (regex on the mangled name)
# usage: tools/kernel_resources.sh k_jacobi3 [library]
set -e
# (one fat binary per translation unit: the objects csrc/mgk_kernels*.o are read, or the file given as $2)
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for LIB in ${2:-$(dirname "$0")/../multigrid_petsc_amd/csrc/mgk_kernels*.o}; do
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$LIB" 2>/dev/null
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
$B/llvm-readelf --notes $T/k.co | python3 -c "
import re, sys
pat = re.compile(sys.argv[1])
blk = {}
for line in sys.stdin:
    m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)', line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == 'agpr_count' and blk.get('name'):
        pass
    if k == 'name' and v.startswith('_Z') or (k == 'name' and 'k_' in v and not v.startswith('a')):
        blk = {'name': v}
    blk[k] = v
    if k == 'wavefront_size' and 'name' in blk and pat.search(blk['name']):
        print(f\"{blk['name'][:90]:90s} vgpr {blk.get('vgpr_count')} agpr {blk.get('agpr_count','-')} sgpr {blk.get('sgpr_count')} lds {blk.get('group_segment_fixed_size')} scratch {blk.get('private_segment_fixed_size')} spill v{blk.get('vgpr_spill_count')} s{blk.get('sgpr_spill_count')}\")
" "$1"
done
