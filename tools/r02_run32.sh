#!/bin/bash
set -u
OUT=gpurun_out/r02t
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repeat or duplication or both_lms" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -3 $OUT/tests.log
timeout 900 python tools/fuzz_gpu.py 300 79 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log
for w in genome_like; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu --steps 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"
done
python3 -c "
import json
for w in 'genome_like'.split():
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>0.6})
    except Exception as e: print(w, 'ERR', e)
"
bash tools/r02_run29.sh 2>&1 | grep -E "refine|copyBuffer|tail"
