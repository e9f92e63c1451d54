#!/bin/bash
set -u
OUT=gpurun_out/r02x
mkdir -p $OUT
for a in "--workload text_like" "--workload pangenome" "--workload periodic --log2n 28"; do
  timeout 600 python bench.py $a --no-e2e --no-cpu --steps 2 > $OUT/bench_x.json 2> $OUT/bench_x.err; echo "$a rc=$?"
  python3 -c "
import json
d=json.loads(open('$OUT/bench_x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], d['build_stats']['doubling_rounds'], d['build_stats']['refine_tiers'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>3.0})
"
done
