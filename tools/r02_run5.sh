#!/bin/bash
set -u
OUT=gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
for db in 8 9 10; do timeout 120 python tools/sortbench.py 3.13e8 40 $db 2>&1 | tail -1 >> $OUT/sortbench.txt; done
timeout 120 python tools/sortbench.py 3.13e8 18 9 2>&1 | tail -1 >> $OUT/sortbench.txt
timeout 120 python tools/sortbench.py 3.13e8 20 10 2>&1 | tail -1 >> $OUT/sortbench.txt
timeout 120 python tools/sortbench.py 3.13e8 16 8 2>&1 | tail -1 >> $OUT/sortbench.txt
cat $OUT/sortbench.txt
timeout 300 python bench.py --workload genome_like --no-e2e --no-cpu > $OUT/bench_genome_like.json 2> $OUT/bench_genome_like.err; echo "rc=$?"
timeout 300 python bench.py --workload n_runs --no-e2e --no-cpu > $OUT/bench_n_runs.json 2> $OUT/bench_n_runs.err; echo "rc=$?"
python3 -c "
import json
for w in ('genome_like','n_runs'):
    d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:v['ms_per_step'] for k,v in d['kernels'].items() if v['ms_per_step']>2})
"
