#!/bin/bash
# same-box A/B of compile-time variants on several workloads: tools/ab_macros_wl.sh FILE.hip[,FILE2.hip] "WORKLOADS" CLASSES "" "-DX=1" ...
cd "${GRAFT_REPO_ROOT:-.}"
file=$1; wls=$2; cls=$3; shift 3
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
for v in "$@"; do
  for f in ${file//,/ }; do touch stralg_amd/csrc/$f; done
  make -s -C stralg_amd/csrc -j16 HIPFLAGS="$base $v" 2>&1 | grep -E "error" | head -3
  for wl in $wls; do
    echo "== [$v] $wl"
    timeout 300 python bench.py --no-e2e --no-cpu --no-other-configs --no-egress --no-ceiling --steps 3 --workload $wl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('verified'), {k:v['ms_per_step'] for k,v in d['kernels'].items() if k in '$cls'.split(',')})"
  done
done
for f in ${file//,/ }; do touch stralg_amd/csrc/$f; done
make -s -C stralg_amd/csrc -j16 2>&1 | grep -E "error" | head -3
