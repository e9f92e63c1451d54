#!/bin/bash
set -u
OUT=gpurun_out/r02k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "hybrid or next_rows or static_key" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -3 $OUT/tests.log
for w in dna n_runs; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?"
done
python3 -c "
import json
for w in ('dna','n_runs'):
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], d['build_stats'].get('sort_local'), {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>0.7})
    except Exception as e: print(w, 'ERR', e)
"
