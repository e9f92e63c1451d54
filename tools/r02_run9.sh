#!/bin/bash
set -u
OUT=gpurun_out/r02i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1150 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -25 $OUT/tests.log
