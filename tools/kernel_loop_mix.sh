#!/bin/bash
# instruction mix of the LONGEST loop (backward branch with the most instructions between target and branch) of a gfx950 kernel of libmgk:
# what a marching step issues.  usage: tools/kernel_loop_mix.sh <regex on the mangled kernel name> [object]
set -e
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for LIB in ${2:-$(dirname "$0")/../multigrid_petsc_amd/csrc/mgk_kernels*.o}; do
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$LIB" 2>/dev/null
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
$B/llvm-objdump -d --no-show-raw-insn $T/k.co > $T/k.s
python3 - "$1" $T/k.s <<'PY'
import re, sys, collections
pat = re.compile(sys.argv[1])
name, body = None, []
kern = {}
for line in open(sys.argv[2]):
    m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
    if m:
        name = m.group(1); kern[name] = []
        continue
    m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)', line)
    if m and name:
        kern[name].append((int(m.group(3), 16), m.group(1), m.group(4)))
for name, ins in kern.items():
    if not pat.search(name) or not ins: continue
    addr = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (a, op, args) in enumerate(ins):
        if op.startswith('s_cbranch') or op == 's_branch':
            m = re.search(r'<\S+\+0x([0-9a-f]+)>', args)
            if not m: continue
            base = ins[0][0]
            tgt = base + int(m.group(1), 16)
            if tgt in addr and addr[tgt] < i and (best is None or i - addr[tgt] > best[1] - best[0]):
                best = (addr[tgt], i)
    if not best: print(name, 'no loop'); continue
    loop = ins[best[0]:best[1] + 1]
    cls = collections.Counter()
    for _, op, _ in loop:
        if op.startswith(('v_add_f64', 'v_mul_f64', 'v_fma_f64', 'v_pk_', 'v_add_f32', 'v_mul_f32', 'v_sub_f32')): c = 'fp arithmetic'
        elif op.startswith(('global_', 'buffer_', 'flat_')): c = 'vector memory'
        elif op.startswith('ds_'): c = 'LDS'
        elif op.startswith(('v_mov', 'v_accvgpr')): c = 'register moves'
        elif op.startswith(('v_cndmask', 'v_cmp')): c = 'selects / compares'
        elif op.startswith('v_'): c = 'other vector (address / integer / dpp)'
        elif op.startswith('s_waitcnt'): c = 's_waitcnt'
        elif op.startswith('s_barrier'): c = 's_barrier'
        elif op.startswith(('s_load', 's_buffer')): c = 'scalar memory'
        else: c = 'other scalar'
        cls[c] += 1
    print(f"{name[:80]}: loop of {len(loop)} instructions")
    for c, n in cls.most_common(): print(f"    {n:5d}  {c}")
    import os
    if os.environ.get("MIX_DETAIL"):
        ops = collections.Counter(op for _, op, _ in loop)
        print("    " + ", ".join(f"{o} {n}" for o, n in ops.most_common(40)))
PY
done
