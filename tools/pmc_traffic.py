#!/usr/bin/env python3
"""HBM-side bytes per unknown of every kernel from the two PMC summaries (tools/pmc_summary.py): FETCH_SIZE doubled (MI355X_MICROARCH.md,
HBM section: on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads), WRITE_SIZE as is; counters in KiB.
usage: pmc_traffic.py fetch.csv write.csv N_unknowns [name-filter] > traffic.json"""
import csv
import json
import sys

def read(path):
    out = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            out[row["Kernel_Name"]] = (int(row["Dispatches"]), float(row["Counter_Value_avg_KiB"]))
    return out

fetch, write, N = read(sys.argv[1]), read(sys.argv[2]), float(sys.argv[3])
flt = sys.argv[4] if len(sys.argv) > 4 else "k_"
res = {}
for name in sorted(fetch):
    if flt not in name or name not in write:
        continue
    rd, wr = 2.0 * fetch[name][1] * 1024.0 / N, write[name][1] * 1024.0 / N
    res[name] = {"dispatches": fetch[name][0], "read_B_per_unknown": round(rd, 3), "write_B_per_unknown": round(wr, 3), "total": round(rd + wr, 3)}
print(json.dumps({"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, with --kernel-trace only); KiB counters; FETCH_SIZE x 2 "
                          "(gfx950: 128-byte requests tallied at 64 bytes, MI355X_MICROARCH.md); per unknown of the FINE grid of the sweep script",
                  "unknowns": N, "kernels": res}, indent=1))
