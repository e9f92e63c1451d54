#!/bin/bash
set -u
OUT=gpurun_out/r02ac
mkdir -p $OUT
for a in "--n 4294967294 --no-tables --no-verify" "--n 3000000000 --no-tables --no-verify" "--workload genome_like"; do
  timeout 600 python bench.py $a --no-e2e --no-cpu --steps 2 > $OUT/bench_x.json 2> $OUT/bench_x.err; echo "$a rc=$?"
  python3 -c "
import json
d=json.loads(open('$OUT/bench_x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], d['build_stats']['sort_local'], d['build_stats']['sort_passes'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step'] > 2})
"
done
