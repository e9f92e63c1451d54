#!/bin/bash
# round 2, first GPU pass: tests, the default bench line, the 2-rank FASTA path on one GPU, the other workloads
set -u
OUT=gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err
STRALG_BENCH_SHARE_GPU=1 STRALG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --log2n 28 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; echo "rc=$?" >> $OUT/bench_2rank.err
timeout 600 python bench.py --workload fasta --no-e2e --no-cpu > $OUT/bench_fasta.json 2> $OUT/bench_fasta.err; echo "rc=$?" >> $OUT/bench_fasta.err
for w in genome_like n_runs text_like; do
  timeout 600 python bench.py --workload $w --no-e2e --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?" >> $OUT/bench_$w.err
done
timeout 600 python bench.py --workload periodic --log2n 28 --no-e2e --no-cpu > $OUT/bench_periodic28.json 2> $OUT/bench_periodic28.err; echo "rc=$?" >> $OUT/bench_periodic28.err
timeout 600 python bench.py --workload bytes --no-e2e --no-cpu > $OUT/bench_bytes.json 2> $OUT/bench_bytes.err; echo "rc=$?" >> $OUT/bench_bytes.err
tail -3 $OUT/tests.log
