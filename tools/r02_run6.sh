#!/bin/bash
set -u
OUT=gpurun_out/r02f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "hybrid or radix_sort or random_against or static_key" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
for mode in 1 2; do
  timeout 300 python bench.py --sort-mode $mode --no-e2e --no-cpu > $OUT/bench_mode$mode.json 2> $OUT/bench_mode$mode.err; echo "rc=$?"
done
timeout 300 python bench.py --workload bytes --sort-mode 2 --no-e2e --no-cpu > $OUT/bench_bytes_mode2.json 2> $OUT/bench_bytes_mode2.err; echo "rc=$?"
timeout 300 python bench.py --workload genome_like --no-e2e --no-cpu > $OUT/bench_genome_like.json 2> $OUT/bench_genome_like.err; echo "rc=$?"
python3 -c "
import json
for w in ('mode1','mode2','bytes_mode2','genome_like'):
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], d['build_stats'].get('sort_local'), {k:v['ms_per_step'] for k,v in d['kernels'].items() if v['ms_per_step']>0.7})
    except Exception as e: print(w, 'ERR', e)
"
