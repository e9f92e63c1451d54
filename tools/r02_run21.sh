#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tables or fused or golden" 2>&1 | tail -2
for i in 1 2; do
timeout 200 python bench.py --no-e2e --no-cpu --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('otable','bwt_gather','radix_scatter','local_sort','induce_scatter')})"
done
timeout 200 python bench.py --workload n_runs --no-e2e --no-cpu --steps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n_runs', d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('otable',)})"
