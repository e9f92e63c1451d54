#!/bin/bash
# rebuilds the library with different radix tile shapes and times the sort (GPU box)
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "256 16 4" "512 16 1" "512 12 2" "1024 16 1"; do
  set -- $cfg; th=$1; items=$2; mw=$3
  touch stralg_amd/csrc/sx_radix.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSX_RADIX_THREADS=$th -DSX_RADIX_ITEMS=$items -DSX_RADIX_MINWAVES=$mw" 2>&1 | grep -E "error" | head -3
  echo "== threads $th items $items minblocks $mw"
  timeout 120 python tools/sortbench.py 3e8 40 2>&1 | tail -1
done
