#!/bin/bash
set -u
OUT=gpurun_out/r02v
mkdir -p $OUT
for seed in 401 402 403; do
  timeout 500 python tools/fuzz_gpu.py 1500 $seed > $OUT/fuzz_$seed.log 2>&1; echo "fuzz $seed rc=$?"; tail -1 $OUT/fuzz_$seed.log | cut -c1-300
done
timeout 900 python tools/soak.py 10 > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -12 $OUT/soak.log
