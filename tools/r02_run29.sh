#!/bin/bash
# kernel trace of the genome-like build: where the refinement rounds spend their time
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
bash tools/r02_trace.sh genome_like --workload genome_like > /dev/null 2>&1
OUT="$ROOT/gpurun_out/trace_genome_like"
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
tot = collections.defaultdict(lambda: [0, 0])
for s, e, k, g in rows:
    k = k.split("(")[0][-60:]
    tot[k][0] += e - s; tot[k][1] += 1
for k, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{d/3e6:9.3f} ms/step {c/3:7.1f} launches  {k}")
print("--- refine_mid dispatches of the last step")
mid = [(s, e, g) for s, e, k, g in rows if "refine_mid" in k]
for s, e, g in mid[-8:]:
    print(f"dur {(e-s)/1e3:9.1f} us grid {g}")
PY
