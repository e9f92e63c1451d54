#!/bin/bash
# diagnostic: the lean local sort's loads and stores alone (-DSX_LS_SKELETON: wrong results, timing only)
cd "${GRAFT_REPO_ROOT:-.}"
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
for v in "" "-DSX_LS_SKELETON"; do
  touch stralg_amd/csrc/sx_localsort.hip
  make -s -C stralg_amd/csrc -j16 HIPFLAGS="$base $v" 2>&1 | grep -E "error" | head -3
  echo "== [$v]"
  timeout 300 python tools/ls_skeleton.py 2>/dev/null | tail -1
done
touch stralg_amd/csrc/sx_localsort.hip
make -s -C stralg_amd/csrc -j16 2>&1 | grep -E "error" | head -3
