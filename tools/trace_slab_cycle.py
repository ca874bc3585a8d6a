#!/usr/bin/env python3
"""per-kernel breakdown of ONE cycle of a slab-rank trace (rocprofv3 --kernel-trace of tools/trace_slab.py), by stream: compute / copy kernels
grouped by name and grid, the modelled exchanges (k_delay) apart.  usage: trace_slab_cycle.py results.db"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, workgroup_x, stream_id from kernels order by start").fetchall()
fin = [i for i, r in enumerate(rows) if "k_finish_sum" in r[0] or "k_flat" in r[0] and False]
# cycles are delimited by the norm's partial-sum finish on the compute stream; take 4 cycles well inside the run
marks = [i for i, r in enumerate(rows) if "k_finish_sum" in r[0]]
a, b = marks[-6] + 1, marks[-2] + 1
nc = 4
cyc = rows[a:b]
span = (cyc[-1][2] - cyc[0][1]) / 1e3 / nc
agg = collections.defaultdict(lambda: [0, 0.0])
for r in cyc:
    k = re.sub(r"\(.*", "", r[0])[:64] + f"  [{r[3] // max(r[4], 1)} x {r[4]}]"
    agg[k][0] += 1
    agg[k][1] += (r[2] - r[1]) / 1e3
print(f"{nc} cycles: span {span:.1f} us/cycle; {sum(v[0] for v in agg.values()) / nc:.0f} launches/cycle")
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1] / nc:9.1f} us/cyc {v[0] / nc:6.1f} x  avg {v[1] / v[0]:8.1f} us  {k}")
    if "k_delay" not in k:
        tot += v[1] / nc
print(f"listed non-delay kernel time {tot:.1f} us/cycle")
