#!/bin/bash
# diagnostic: the lean local sort's phase cycles and the workgroups it leaves to the other kernel (a library built with
# -DSX_LS_PROBE prints them)   tools/ls_probe.sh WORKLOAD [extra flags]
cd "${GRAFT_REPO_ROOT:-.}"
wl=${1:-dna}; shift
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
touch stralg_amd/csrc/sx_localsort.hip
make -s -C stralg_amd/csrc -j16 HIPFLAGS="$base -DSX_LS_PROBE $*" 2>&1 | grep -E "error" | head -3
timeout 300 python bench.py --no-e2e --no-cpu --no-other-configs --no-ceiling --steps 2 --warmup 1 --workload $wl 2>&1 >/dev/null | grep "local_sort" | tail -8
touch stralg_amd/csrc/sx_localsort.hip
make -s -C stralg_amd/csrc -j16 2>&1 | grep -E "error" | head -3
