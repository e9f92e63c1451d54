"""every launch of the last step in a rocprofv3 kernel trace, in time order, with the idle gaps before them
   python tools/step_timeline.py <trace dir> [first-kernel-substring] [last-kernel-substring]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                     r.get("Workgroup_Size_X", "?")))
rows.sort()
first = sys.argv[2] if len(sys.argv) > 2 else "cls_first"
last = sys.argv[3] if len(sys.argv) > 3 else "otable"
if not rows:
    sys.exit(f"step_timeline: no kernel trace under {sys.argv[1]}")
ends = [i for i, r in enumerate(rows) if last in r[2]]
i1 = ends[-1] if ends else len(rows) - 1  # (--no-tables, byte workloads: no such kernel -- up to the trace's end)
starts = [i for i, r in enumerate(rows[:i1]) if first in r[2]]
if not starts:  # (the direct sort has no cls_first: the whole trace)
    print(f"step_timeline: no kernel matching '{first}' before the last '{last}': showing the whole trace", file=sys.stderr)
i0 = starts[-1] if starts else 0
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 20000 and first not in rows[i0 - 1][2]:  # the step's leading copies
    i0 -= 1
prev = rows[i0][0]
tot_gap = 0.0
for s, e, nme, g, wg in rows[i0:i1 + 1]:
    gap = (s - prev) / 1e3
    tot_gap += max(gap, 0)
    short = nme.split("(")[0].replace("void sx::", "").replace("sx::", "")[:70]
    print(f"gap {gap:7.1f}  dur {(e - s) / 1e3:8.1f}  grid {g:>9} x {wg:>4}  {short}")
    prev = max(prev, e)
print("span ms", (rows[i1][1] - rows[i0][0]) / 1e6, "gaps ms", tot_gap / 1e3, "launches", i1 - i0 + 1)
