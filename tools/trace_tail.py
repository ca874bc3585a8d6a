"""print the long kernels of the last V-cycle of a rocprofv3 kernel trace (tuning aid)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 > thr]
for r in big[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{r['Kernel_Name'][:55]:55s} {d:9.1f} us grid={r['Grid_Size_X']}x{r['Grid_Size_Y']} wg={r['Workgroup_Size_X']} vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']}")
