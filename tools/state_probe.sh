#!/bin/bash
# does a box that has turned slow recover after idling?  (bench lines: ms a step, the induce scatters' launch)
cd "${GRAFT_REPO_ROOT:-.}"
run() { python bench.py --no-e2e --no-cpu --no-other-configs --no-egress --no-ceiling --no-verify --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys,time
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', time.strftime('%H:%M:%S'), d['ms_per_step'], r['avg_ms'])"; }
for i in 1 2 3 4 5 6; do run b$i; done
sleep 45
for i in 7 8; do run after_idle_$i; done
rocm-smi --showpower --showclocks --showtemp 2>/dev/null | head -30
