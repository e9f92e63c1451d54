#!/bin/bash
set -u
OUT=gpurun_out/r02w
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "near_identical or general or both_lms or golden" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -3 $OUT/tests.log
timeout 600 python tools/fuzz_gpu.py 600 201 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log | cut -c1-400
for a in "--workload pangenome" "--workload pangenome --sort-mode 1" "--workload text_like" "--workload periodic --log2n 28"; do
  timeout 600 python bench.py $a --no-e2e --no-cpu --steps 2 > $OUT/bench_x.json 2> $OUT/bench_x.err; echo "$a rc=$?"
  python3 -c "
import json
d=json.loads(open('$OUT/bench_x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], d['build_stats']['doubling_rounds'], d['build_stats']['refine_tiers'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>3.0})
"
done
