#!/bin/bash
# GPU suite + every bench workload after the 1024 x 8 radix tile became the default
set -u
OUT=gpurun_out/r02q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -3 $OUT/tests.log
WL="dna genome_like n_runs text_like bytes periodic"
for w in $WL; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu --steps 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"
done
python3 -c "
import json
for w in '$WL'.split():
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>0.6})
    except Exception as e: print(w, 'ERR', e)
"
