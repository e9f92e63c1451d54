#!/bin/bash
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
bash tools/r02_trace.sh dna > /dev/null 2>&1
python3 tools/gaps.py "$ROOT/gpurun_out/trace_dna"
