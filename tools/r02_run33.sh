#!/bin/bash
set -u
OUT=gpurun_out/r02u
mkdir -p $OUT
for a in "--workload pangenome --log2n 28" "--workload pangenome"; do
  timeout 600 python bench.py $a --no-e2e --no-cpu --steps 2 > $OUT/bench_pan.json 2> $OUT/bench_pan.err; echo "$a rc=$?"
  python3 -c "
import json
d=json.loads(open('$OUT/bench_pan.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], d['build_stats'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>1.0})
"
done
