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
CSETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"
 "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
)
i=0
for g in "${CSETS[@]}"; do
  timeout 150 rocprofv3 --pmc $g --output-format csv -d "$OUT/g$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/g$i.log" 2>&1
  i=$((i+1))
done
python3 "$ROOT/tools/pmc_probe_summary.py" "$OUT" "$@"
