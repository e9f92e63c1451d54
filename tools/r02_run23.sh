#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for v in "-DSX_LS_SPAN=4096 -DSX_LS_FIRST=10" "-DSX_LS_SPAN=4608 -DSX_LS_FIRST=11" "-DSX_LS_SPAN=5120 -DSX_LS_FIRST=12" "-DSX_LS_SPAN=3584 -DSX_LS_FIRST=9"; do
  touch stralg_amd/csrc/sx_localsort.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off $v" 2>&1 | grep -E "error" | head -3
  echo "== $v"
  timeout 200 python bench.py --no-e2e --no-cpu --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], d['build_stats']['sort_local'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('local_sort','radix_scatter')})"
done
