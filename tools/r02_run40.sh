#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for g in 1073741824 262144 65536 16384; do
  touch stralg_amd/csrc/sx_bwt.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSX_SMALL_GRID=${g}u" 2>&1 | grep -E "error" | head -3
  echo "== grid cap $g"
  timeout 200 python bench.py --no-e2e --no-cpu --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('otable',)})"
done
