#!/bin/bash
set -u
OUT=gpurun_out/r02l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "induce_round_forms or induced_passes or wide_alphabets or fuzz or long_runs or structured" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 300 python bench.py --workload bytes --no-direct-sort --no-e2e --no-cpu > $OUT/bench_bytes_induced.json 2> $OUT/bench_bytes_induced.err; echo "rc=$?"
timeout 300 python bench.py --workload text_like --no-e2e --no-cpu > $OUT/bench_text_like.json 2> $OUT/bench_text_like.err; echo "rc=$?"
timeout 300 python bench.py --workload uniform --sigma 21 --no-direct-sort --no-e2e --no-cpu > $OUT/bench_s21_induced.json 2> $OUT/bench_s21_induced.err; echo "rc=$?"
python3 -c "
import json
for w in ('bytes_induced','text_like','s21_induced'):
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>1.5})
    except Exception as e: print(w, 'ERR', e)
"
