#!/bin/bash
set -u
OUT=gpurun_out/r02o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "runs or induce_round_forms or structured or golden or random_against or fuzz or long_repeats or full_size" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
for w in dna genome_like n_runs; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?"
done
python3 -c "
import json
for w in ('dna','genome_like','n_runs'):
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>1.0})
    except Exception as e: print(w, 'ERR', e)
"
