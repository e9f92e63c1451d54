#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers (run on the GPU box via gpurun).
#   1. kernel trace + stats of the default bench command
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE separately, as the MI355X guide prescribes)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -50
python3 "$ROOT/tools/profile_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
