#!/usr/bin/env python3
"""per-kernel breakdown of ONE V-cycle from a rocprofv3 --kernel-trace database (rocpd sqlite): kernels between the last two
k_finish_sum launches of the run (one finish per cycle), grouped by name and grid.  usage: trace_cycle.py results.db [ncycles]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
fin = [i for i, r in enumerate(rows) if "k_finish_sum" in r[0]]
if len(fin) < nc + 1:
    raise SystemExit(f"only {len(fin)} cycles in the trace")
a, b = fin[-nc - 1] + 1, fin[-1] + 1
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[a:b]:
    k = re.sub(r"\(.*", "", r[0])[:70] + f"  [{r[3] // max(r[4], 1)} x {r[4]}]"
    agg[k][0] += 1
    agg[k][1] += (r[2] - r[1]) / 1e3
span = (rows[b - 1][2] - rows[a][1]) / 1e3
busy = sum(v[1] for v in agg.values())
print(f"{nc} cycles: span {span / nc:.1f} us/cycle, kernel time {busy / nc:.1f} us/cycle, idle {100 * (1 - busy / span):.1f} %, {sum(v[0] for v in agg.values()) / nc:.0f} launches/cycle")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{v[1] / nc:9.1f} us/cyc {v[0] / nc:6.1f} x  avg {v[1] / v[0]:8.1f} us  {k}")
