#!/bin/bash
# timings + rocprofv3 kernel stats of tools/bench_next.py (inverse + LCP, exact search, FASTA pack + remap)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/r02m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "next_rows or reference_named" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -3 $OUT/tests.log
timeout 600 python tools/bench_next.py > $OUT/bench_next.txt 2>&1; echo "rc=$?" >> $OUT/bench_next.txt
cat $OUT/bench_next.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/bench_next.py" > "$OUT/trace.log" 2>&1
python3 "$ROOT/tools/profile_summary.py" "$OUT" 2>&1 | grep -E "kernel |lcp|inverse|search|fasta|remap|radix" | head -30
