#!/bin/bash
set -u
OUT=gpurun_out/r02g
mkdir -p $OUT
export TMPDIR=/tmp
for cm in 16384 131072 524288 4194304; do
  timeout 300 python bench.py --workload bytes --no-direct-sort --chain-max $cm --no-e2e --no-cpu --no-verify --steps 2 > $OUT/bench_bi_$cm.json 2> $OUT/bench_bi_$cm.err
  python3 -c "
import json
d=json.loads(open('$OUT/bench_bi_$cm.json').read().strip().splitlines()[-1]); print($cm, d['ms_per_step'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>2})
"
done
