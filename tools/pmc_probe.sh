#!/bin/bash
# Hardware-counter passes over the bench command, one rocprofv3 run per counter group
# (PMC only: no tracing flags alongside).  Usage: tools/pmc_probe.sh [kernel-substring ...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/pmc_probe"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="${BENCH_ARGS:---steps 1 --warmup 0 --no-cpu}"
# (TCC_* / TCP_* groups made rocprofv3 abort and then hang until the gpurun limit on this pool: SQ counters only,
#  and every pass under its own timeout)
CSETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
)
i=0
for g in "${CSETS[@]}"; do
  timeout 150 rocprofv3 --pmc $g --output-format csv -d "$OUT/g$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/g$i.log" 2>&1
  i=$((i+1))
done
python3 "$ROOT/tools/pmc_probe_summary.py" "$OUT" "$@"
