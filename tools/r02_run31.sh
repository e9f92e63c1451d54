#!/bin/bash
# span of the LDS refinement kernel (same box)
cd "${GRAFT_REPO_ROOT:-.}"
for sp in 2048 1024 3072 2048; do
  touch stralg_amd/csrc/sx_lmssort.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSX_RM_SPAN=$sp" 2>&1 | grep -E "error" | head -3
  echo "== span $sp"
  timeout 200 python bench.py --workload genome_like --no-e2e --no-cpu --steps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in ('doubling','radix_scatter','radix_hist','scan')})"
done
