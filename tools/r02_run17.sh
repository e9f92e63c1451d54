#!/bin/bash
# final-state measurements of round 2: default bench line, rocprofv3 kernel trace + PMC passes, the other workloads, (f) rows
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=gpurun_out/r02z
mkdir -p $OUT
export TMPDIR=/tmp
T0=$SECONDS; timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$? wall=$((SECONDS-T0)) s" >> $OUT/bench_default.err
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu --no-e2e --no-verify" timeout 600 bash tools/profile.sh > $OUT/profile.log 2>&1
cp gpurun_out/prof/summary.txt $OUT/profile_summary.txt
find gpurun_out/prof/trace -name "*kernel_stats.csv" -newer $OUT/bench_default.json -exec cp {} $OUT/kernel_stats.csv \;
for w in fasta genome_like n_runs bytes text_like; do
  timeout 300 python bench.py --workload $w --no-e2e --no-cpu > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "rc=$?" >> $OUT/bench_$w.err
done
timeout 300 python bench.py --log2n 28 --no-e2e --no-cpu > $OUT/bench_dna28.json 2> $OUT/bench_dna28.err
timeout 300 python bench.py --workload bytes --no-direct-sort --no-e2e --no-cpu > $OUT/bench_bytes_induced.json 2> $OUT/bench_bytes_induced.err
timeout 300 python bench.py --workload periodic --log2n 28 --no-e2e --no-cpu > $OUT/bench_periodic28.json 2> $OUT/bench_periodic28.err
STRALG_BENCH_SHARE_GPU=1 STRALG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --log2n 28 > $OUT/bench_2rank_shared_gpu.json 2> $OUT/bench_2rank.err; echo "rc=$?" >> $OUT/bench_2rank.err
timeout 600 python tools/bench_next.py > $OUT/bench_next.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/next/trace" -- python3 "$ROOT/tools/bench_next.py" > "$ROOT/$OUT/next_trace.log" 2>&1
cd "$ROOT"
python3 tools/profile_summary.py $OUT/next > $OUT/next_summary.txt 2>&1
head -c 300 $OUT/bench_default.json
