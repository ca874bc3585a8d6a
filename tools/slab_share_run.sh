#!/bin/bash
# one rank's share of the 8-GPU cycle beside the single-GPU cycle of the SAME box: bench.py (headline only) -> ms per cycle -> tools/slab_share.py
# usage: tools/slab_share_run.sh OUT.json [extra slab_share.py arguments]
out=$1; shift
ms=$(python3 bench.py --steps 10 --warmup 2 --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
echo "single GPU: $ms ms per cycle"
python3 tools/slab_share.py --single-ms $ms --models free,expected --overlap 1 --slab-chunks=-1,0 --out $out "$@" 2>&1 | grep slab_share
python3 -c "
import json; j = json.load(open('$out'))
for k, v in j['models'].items(): print(k, 'worst rank %.3f ms' % v['worst_rank_ms'], 'predicted 1->8 speed-up %.2f' % v['predicted_speedup_1_to_N'])
"
