"""print the interesting fields of bench.py JSON lines"""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e, open(f).read()[-500:])
        continue
    bs = d["build_stats"]
    print(f, "| Msuf/s", d["value"], "| ms/step", d["ms_per_step"], "| sa ms", round(bs["ms_total"], 2), "| rounds", bs["induce_rounds"],
          "| passes", bs["sort_passes"], "| path", bs.get("lms_path"))
    print("   ", {k: v["ms_per_step"] for k, v in d["kernels"].items()})
