#!/bin/bash
# rebuilds the library with different radix tile sizes and times the sort (GPU box)
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "16 1" "16 4" "16 5" "12 5" "20 3" "24 3"; do
  set -- $cfg; items=$1; mw=$2
  touch stralg_amd/csrc/sx_radix.hip
  make -s -C stralg_amd/csrc -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSX_RADIX_ITEMS=$items -DSX_RADIX_MINWAVES=$mw" 2>&1 | grep -E "error" | head -3
  echo "== items $items minwaves $mw"
  python tools/sortbench.py 3e8 40 2>&1 | tail -1
done
