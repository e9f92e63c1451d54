#!/bin/bash
# round 2 (second session), first GPU pass: default bench line, rocprofv3 kernel trace + PMC passes of the same
# command, then the other workloads
set -u
OUT=gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu --no-e2e --no-verify" timeout 600 bash tools/profile.sh > $OUT/profile.log 2>&1
cp gpurun_out/prof/summary.txt $OUT/profile_summary.txt
for w in fasta genome_like n_runs text_like bytes; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?" >> $OUT/bench_$w.err
done
timeout 300 python bench.py --workload bytes --no-direct-sort --no-e2e --no-cpu > $OUT/bench_bytes_induced.json 2> $OUT/bench_bytes_induced.err
timeout 300 python bench.py --workload periodic --log2n 28 --no-e2e --no-cpu > $OUT/bench_periodic28.json 2> $OUT/bench_periodic28.err; echo "rc=$?" >> $OUT/bench_periodic28.err
head -c 600 $OUT/bench_default.json
