"""list individual dispatches of kernels matching a substring from a rocprofv3 kernel_trace.csv"""
import csv, glob, sys
pat = sys.argv[2]
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?")))
rows.sort()
t0 = rows[0][0] if rows else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
for s, e, g, wg in rows[:n]:
    print(f"start {(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  grid {g} wg {wg}")
print("total", sum(e - s for s, e, _, _ in rows) / 1e6, "ms over", len(rows))
