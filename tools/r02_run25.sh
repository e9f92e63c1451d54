#!/bin/bash
set -u
OUT=gpurun_out/r02p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -3 $OUT/tests.log
for w in dna genome_like n_runs; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu --steps 5 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?"
done
python3 -c "
import json
for w in ('dna','genome_like','n_runs'):
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>0.6})
    except Exception as e: print(w, 'ERR', e)
"
