#!/bin/bash
set -u
OUT=gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err
timeout 600 python bench.py --workload text_like --no-e2e --no-cpu > $OUT/bench_text_like.json 2> $OUT/bench_text_like.err; echo "rc=$?" >> $OUT/bench_text_like.err
timeout 600 python bench.py --workload text_like --log2n 28 --no-e2e --no-cpu > $OUT/bench_text_like28.json 2> $OUT/bench_text_like28.err; echo "rc=$?" >> $OUT/bench_text_like28.err
cat /sys/kernel/mm/transparent_hugepage/enabled > $OUT/thp.txt 2>&1
nproc >> $OUT/thp.txt; free -g >> $OUT/thp.txt
tail -3 $OUT/tests.log
