#!/bin/bash
# wide O-table kernel: sigma 21 / 12 / 64 at 2^28..2^30
set -u
OUT=gpurun_out/r02y
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "table or bwt" 2>&1 | tail -2
for a in "--workload uniform --sigma 21" "--workload uniform --sigma 12 --log2n 29" "--workload uniform --sigma 64 --log2n 28"; do
  timeout 600 python bench.py $a --no-e2e --no-cpu --steps 3 > $OUT/bench_x.json 2> $OUT/bench_x.err; echo "$a rc=$?"
  python3 -c "
import json
d=json.loads(open('$OUT/bench_x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step'],v.get('GBps')) for k,v in d['kernels'].items() if k in ('otable','bwt_gather')})
"
done
