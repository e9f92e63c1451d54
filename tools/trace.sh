#!/bin/bash
# kernel trace + stats only (quick look at per-kernel times)
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/trace_only"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
python3 "$ROOT/tools/profile_summary.py" "$OUT" 2>&1 | head -45
