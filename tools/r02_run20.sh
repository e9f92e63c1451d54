#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for v in "" "-DSX_TAIL_ALWAYS_BATCH"; do
  touch stralg_amd/csrc/sx_induce.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off $v" 2>&1 | grep -E "error" | head -3
  echo "== $v"
  timeout 200 python bench.py --workload genome_like --no-e2e --no-cpu --steps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if k.startswith('induce')}, d['build_stats']['induce_rounds'])"
done
