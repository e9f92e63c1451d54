#!/bin/bash
# rocprofv3 kernel traces of bench.py workloads: tools/r02_trace.sh <name> <bench args...>
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
NAME=$1; shift
OUT="$ROOT/gpurun_out/trace_$NAME"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" "$@" --steps 2 --warmup 1 --no-cpu --no-e2e --no-verify > "$OUT/trace.log" 2>&1
python3 "$ROOT/tools/profile_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
head -45 "$OUT/summary.txt"
