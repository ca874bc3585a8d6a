#!/usr/bin/env python3
"""idle time of the compute work in ONE cycle of a slab-share run (rocprofv3 --kernel-trace database): for every gap between two
consecutive non-delay kernels, what it waited for (the k_delay = modelled link time that ended last before the next kernel).
usage: trace_gaps.py results.db [min_gap_us]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = db.execute("select name, start, end, grid_x, workgroup_x, stream_id from kernels order by start").fetchall()
fin = [i for i, r in enumerate(rows) if "k_finish_sum" in r[0]]
a, b = fin[-3] + 1, fin[-2] + 1                       # one full cycle well inside the timed region
cyc = rows[a:b]
work = [r for r in cyc if "k_delay" not in r[0]]
delays = [r for r in cyc if "k_delay" in r[0]]
span = (cyc[-1][2] - cyc[0][1]) / 1e3
busy = sum(r[2] - r[1] for r in work) / 1e3
print(f"cycle span {span:.0f} us; compute/copy kernels {busy:.0f} us ({len(work)} launches); {len(delays)} modelled exchanges, {sum(r[2]-r[1] for r in delays)/1e3:.0f} us of link time")
tot = 0.0
prev_end = work[0][2]
for r in work[1:]:
    gap = (r[1] - prev_end) / 1e3
    if gap > ming:
        d = [x for x in delays if x[2] <= r[1] + 2000 and x[2] >= prev_end - 2000]
        why = f"after a modelled exchange of {(d[-1][2]-d[-1][1])/1e3:.0f} us" if d else "launch gap"
        nm = re.sub(r"\(.*", "", r[0])[:48]
        print(f"  gap {gap:7.1f} us before {nm:48s} [{r[3]//max(r[4],1)} x {r[4]}]  {why}")
        tot += gap
    prev_end = max(prev_end, r[2])
print(f"sum of gaps > {ming} us: {tot:.0f} us")
