#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "induce_round_forms or induced_passes or wide_alphabets or radix_sort" 2>&1 | tail -2
for w in "--workload text_like" "--workload bytes --no-direct-sort" "--workload uniform --sigma 21 --no-direct-sort"; do
timeout 300 python bench.py $w --no-e2e --no-cpu --steps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('otable','radix_hist','induce_scatter','induce_gather','induce_scan','radix_scatter')})"
done
