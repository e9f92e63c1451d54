#!/bin/bash
# induce grid cap experiment
cd "${GRAFT_REPO_ROOT:-.}"
for cap in 4096 16384 65536 262144; do
  touch stralg_amd/csrc/sx_induce.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSX_INDUCE_GRID_CAP=$cap" 2>&1 | grep -E "error" | head -3
  echo "== cap $cap"
  timeout 200 python bench.py --no-e2e --no-cpu --no-verify --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k.startswith('induce')})"
done
