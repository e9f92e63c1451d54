#!/bin/bash
# same-box A/B of the FASTA pack's tiles a workgroup (SX_FASTA_SUB)
cd "${GRAFT_REPO_ROOT:-.}"
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
for v in "-DSX_FASTA_SUB=1" "-DSX_FASTA_SUB=2" "-DSX_FASTA_SUB=4" "-DSX_FASTA_SUB=8"; do
  touch stralg_amd/csrc/sx_fasta.hip
  make -s -C stralg_amd/csrc -j16 HIPFLAGS="$base $v" 2>&1 | grep -E "error" | head -3
  echo "== [$v]"
  timeout 300 python tools/bench_next.py 2>/dev/null | grep "FASTA pack, "
done
touch stralg_amd/csrc/sx_fasta.hip
make -s -C stralg_amd/csrc -j16 2>&1 | grep -E "error" | head -3
