#!/bin/bash
set -u
OUT=gpurun_out/r02zz
mkdir -p $OUT
timeout 600 python bench.py --workload pangenome --no-e2e --no-cpu > $OUT/bench_pangenome.json 2> $OUT/err.txt; echo rc=$?
timeout 600 python bench.py --workload uniform --sigma 21 --no-e2e --no-cpu > $OUT/bench_sigma21.json 2>> $OUT/err.txt; echo rc=$?
timeout 600 python bench.py --workload uniform --sigma 21 --no-direct-sort --no-e2e --no-cpu > $OUT/bench_sigma21_induced.json 2>> $OUT/err.txt; echo rc=$?
python3 -c "
import json
for w in ('pangenome','sigma21','sigma21_induced'):
    d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>3.0})
"
