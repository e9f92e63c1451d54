#!/bin/bash
# one counter pass (instruction counts and wave cycles per kernel) over the bench command
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/pmc_valu"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/g0" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu > "$OUT/g0.log" 2>&1
python3 "$ROOT/tools/pmc_probe_summary.py" "$OUT" "$@"
