"""idle time between consecutive kernels of the last build in a rocprofv3 kernel trace: tools/gaps.py <trace dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:]))
rows.sort()
# the last step: from the last cls_first kernel on
starts = [i for i, r in enumerate(rows) if "cls_first" in r[2]]
seg = rows[starts[-1]:]
end = next((i for i, r in enumerate(seg) if "otable" in r[2]), len(seg) - 1)
seg = seg[:end + 1]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
print(f"kernels {len(seg)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {(span-busy)/1e6:.3f} ms")
gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1)]
small = sum(g for g, _, _ in gaps if g < 5000)
print(f"gaps below 5 us: {sum(1 for g,_,_ in gaps if g < 5000)} totalling {small/1e6:.3f} ms")
for g, a, b in sorted(gaps, reverse=True)[:24]:
    print(f"{g/1e3:8.1f} us  after {a}  before {b}")
