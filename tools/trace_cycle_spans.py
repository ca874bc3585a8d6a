#!/usr/bin/env python3
"""time between successive k_finish_sum launches (one per cycle) in a rocprofv3 --kernel-trace database, and the largest gaps between
kernels inside the longest span: where a run's first cycle loses its time.  usage: trace_cycle_spans.py results.db"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
fin = [i for i, r in enumerate(rows) if "k_finish_sum" in r[0]]
print("kernels", len(rows), "finishes", len(fin))
for a, b in zip(fin[:-1], fin[1:]):
    busy = sum(rows[i][2] - rows[i][1] for i in range(a + 1, b + 1)) / 1e3
    print(f"  span {(rows[b][2] - rows[a][2]) / 1e3:9.1f} us  busy {busy:8.1f} us  launches {b - a}")
if len(fin) >= 2:
    a, b = max(zip(fin[:-1], fin[1:]), key=lambda ab: rows[ab[1]][2] - rows[ab[0]][2])        # the longest span
    gaps = sorted(((rows[i + 1][1] - rows[i][2]) / 1e3, rows[i][0][:50], rows[i + 1][0][:50]) for i in range(a, b))[-8:]
    for g in reversed(gaps):
        print(f"  gap {g[0]:8.1f} us between {g[1]} -> {g[2]}")
