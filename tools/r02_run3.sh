#!/bin/bash
set -u
OUT=gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
for w in fasta genome_like n_runs; do
  timeout 600 python bench.py --workload $w --no-e2e --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?" >> $OUT/bench_$w.err
done
timeout 600 python bench.py --no-e2e --no-cpu > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/tests.log
