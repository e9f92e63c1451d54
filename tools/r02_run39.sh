#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for g in 4194304 65536 16384 4096; do
  touch stralg_amd/csrc/sx_bwt.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSX_WIDE_GRID=${g}u" 2>&1 | grep -E "error" | head -3
  echo "== grid cap $g"
  for a in "--workload uniform --sigma 21" "--workload uniform --sigma 12 --log2n 29"; do
  timeout 200 python bench.py $a --no-e2e --no-cpu --steps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('otable',)})"
  done
done
