#!/bin/bash
set -u
OUT=gpurun_out/r02n
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
for w in genome_like text_like; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?"
done
timeout 300 python bench.py --workload periodic --log2n 28 --no-e2e --no-cpu > $OUT/bench_periodic28.json 2> $OUT/bench_periodic28.err; echo "rc=$?"
python3 -c "
import json
for w in ('genome_like','text_like','periodic28'):
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>2.0})
    except Exception as e: print(w, 'ERR', e)
"
