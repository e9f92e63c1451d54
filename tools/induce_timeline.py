"""one step's induce launches in time order with gaps (from a rocprofv3 kernel trace)"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
# last step: find the last set_entry_kernel
idx = [i for i, r in enumerate(rows) if "set_entry" in r[2]]
i0 = idx[-1]
end = next((i for i in range(i0, len(rows)) if "bwt_count" in rows[i][2] or "otable" in rows[i][2]), len(rows) - 1)
prev = rows[i0][0]
tot_gap = 0
for s, e, nme, g in rows[i0:end + 1]:
    gap = (s - prev) / 1e3
    tot_gap += max(gap, 0)
    print(f"gap {gap:7.1f}  dur {(e - s) / 1e3:8.1f}  grid {g:>9}  {nme}")
    prev = e
print("span ms", (rows[end][1] - rows[i0][0]) / 1e6, "gaps ms", tot_gap / 1e3)
