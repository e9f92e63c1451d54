#!/bin/bash
# Round 5's evidence, collected on a GPU box (gpurun calls: parts a, b, c, then d once the collected traffic is in the tree); the summaries are copied into profiles/
# afterwards by tools/r05_collect.py.   bash tools/r05_final.sh a|b|c|d
set -o pipefail
part=$1
out=gpurun_out/r05_final
mkdir -p $out
common="--no-cpu --no-e2e --no-other-configs"
case $part in
  a) # the driver-like line (with the CPU baselines and the host-buffer legs), its kernel trace and PMC passes
    bash tools/gpu_step.sh r05_final/a "bench:--steps 20 --warmup 5" "prof:bench.py --steps 2 --warmup 1 $common" \
      "py:tools/e2e_probe.py 30";;
  b) # PMC passes of the other configurations bench.py reports (other_configs and the FASTA step)
    bash tools/gpu_step.sh r05_final/b "prof:bench.py --log2n 28 --steps 2 --warmup 1 $common" \
      "prof:bench.py --workload bytes --steps 2 --warmup 1 $common" \
      "prof:bench.py --workload bytes --no-direct-sort --steps 2 --warmup 1 $common" \
      "prof:bench.py --workload genome_like --steps 2 --warmup 1 $common" \
      "prof:bench.py --workload fasta --steps 2 --warmup 1 $common --no-egress --no-ro";;
  c) # lines of the other workloads, the two-rank legs, the next rows
    bash tools/gpu_step.sh r05_final/c "bench:--workload bytes --steps 5 --warmup 2 $common" \
      "bench:--workload bytes --no-direct-sort --steps 5 --warmup 2 $common" \
      "bench:--workload uniform --sigma 21 --steps 3 --warmup 1 $common" \
      "bench:--workload uniform --sigma 21 --no-direct-sort --steps 3 --warmup 1 $common" \
      "bench:--workload uniform --sigma 6 --steps 5 --warmup 2 $common" \
      "bench:--workload genome_like --steps 5 --warmup 2 $common" \
      "bench:--workload n_runs --steps 5 --warmup 2 $common" \
      "bench:--workload text_like --steps 3 --warmup 1 $common" \
      "bench:--workload pangenome --steps 3 --warmup 1 $common" \
      "bench:--workload periodic --steps 3 --warmup 1 $common" \
      "bench:--workload fasta --steps 5 --warmup 2 --no-cpu --no-e2e" \
      "bench:--log2n 22 --steps 50 --warmup 5 $common" \
      "bench2n:--log2n 26 --steps 3 --warmup 1" "bench2:--log2n 26 --steps 3 --warmup 1" \
      "py:tools/bench_next.py" "prof:tools/bench_next.py";;
  d) # after tools/r05_collect.py has stamped profiles/pmc_traffic.json: the lines that quote it, once more (traffic_stale false)
    common2="--no-cpu --no-e2e --no-other-configs"
    mkdir -p $out/a $out/c
    timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $out/a/bench_1.json 2> $out/a/bench_1.err &&
    timeout -k 10 600 python bench.py --workload bytes --steps 5 --warmup 2 $common2 > $out/c/bench_1.json 2> $out/c/bench_1.err &&
    timeout -k 10 600 python bench.py --workload genome_like --steps 5 --warmup 2 $common2 > $out/c/bench_6.json 2> $out/c/bench_6.err &&
    timeout -k 10 600 python bench.py --workload fasta --steps 5 --warmup 2 --no-cpu --no-e2e > $out/c/bench_11.json 2> $out/c/bench_11.err
    echo "rc=$?";;
esac
