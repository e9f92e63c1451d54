#!/bin/bash
# same-box A/B of two versions of sx_localsort.hip (tools/ab/ls_a.hip, tools/ab/ls_b.hip)
cd "${GRAFT_REPO_ROOT:-.}"
for v in a b a b; do
  cp tools/ab/ls_$v.hip stralg_amd/csrc/sx_localsort.hip
  make -s -C stralg_amd/csrc -j8 2>&1 | grep -E "error" | head -3
  echo "== $v"
  timeout 200 python bench.py --no-e2e --no-cpu --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('local_sort','radix_scatter')})"
done
