"""sum every counter of tools/pmc_probe.sh per kernel (substring filter optional)"""
import csv, glob, sys, collections
root = sys.argv[1]
pats = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pats and not any(p in k for p in pats):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print("==", k[:90])
    for c in sorted(acc[k]):
        print(f"   {c:40s} {acc[k][c]:18.0f}   ({calls[k][c]} dispatches)")
