#!/bin/bash
# tie refinement in LDS: the new parity test, a fuzz run, genome_like / dna / text_like timings
set -u
OUT=gpurun_out/r02r
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repeat or hybrid or both_lms or fuzz" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -3 $OUT/tests.log
timeout 900 python tools/fuzz_gpu.py 400 77 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz.log
WL="genome_like dna n_runs"
for w in $WL; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu --steps 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w rc=$?"
done
timeout 300 python bench.py --workload genome_like --sort-mode 1 --no-e2e --no-cpu --steps 3 > $OUT/bench_genome_like_mode1.json 2> $OUT/bench_gl1.err; echo "mode1 rc=$?"
python3 -c "
import json
for w in '$WL genome_like_mode1'.split():
    try:
        d=json.loads(open('$OUT/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if v['ms_per_step']>0.6})
    except Exception as e: print(w, 'ERR', e)
"
