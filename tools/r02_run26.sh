#!/bin/bash
# radix tile shapes (same-box): threads x keys per thread
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "512 16" "1024 8" "512 16" "1024 8"; do
  set -- $cfg; th=$1; items=$2
  touch stralg_amd/csrc/sx_radix.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSX_RADIX_THREADS=$th -DSX_RADIX_ITEMS=$items" 2>&1 | grep -E "error" | head -3
  echo "== threads $th items $items"
  timeout 200 python bench.py --no-e2e --no-cpu --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('radix_scatter','radix_hist','scan')})"
done
