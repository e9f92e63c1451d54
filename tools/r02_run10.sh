#!/bin/bash
# default bench line + rocprofv3 kernel trace + PMC passes of the same command (mid-session state: hybrid sort, tail kernel in registers)
set -u
OUT=gpurun_out/r02j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu --no-e2e --no-verify" timeout 600 bash tools/profile.sh > $OUT/profile.log 2>&1
cp gpurun_out/prof/summary.txt $OUT/profile_summary.txt
find gpurun_out/prof/trace -name "*kernel_stats.csv" -newer $OUT/bench_default.json -exec cp {} $OUT/kernel_stats.csv \;
for w in fasta genome_like n_runs bytes; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?" >> $OUT/bench_$w.err
done
timeout 300 python bench.py --log2n 28 --no-e2e --no-cpu > $OUT/bench_dna28.json 2> $OUT/bench_dna28.err
head -c 400 $OUT/bench_default.json
