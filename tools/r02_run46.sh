#!/bin/bash
set -u
OUT=gpurun_out/r02ab
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hybrid or sort" 2>&1 | tail -2
timeout 600 python tools/fuzz_gpu.py 800 301 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log | cut -c1-300
for a in "--log2n 30" "--log2n 30 --sort-mode 3" "--log2n 31 --no-tables" "--log2n 31 --no-tables --sort-mode 1" "--workload genome_like --sort-mode 3"; do
  timeout 600 python bench.py $a --no-e2e --no-cpu --steps 3 > $OUT/bench_x.json 2> $OUT/bench_x.err; echo "$a rc=$?"
  python3 -c "
import json
d=json.loads(open('$OUT/bench_x.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['verified'], d['build_stats']['sort_local'], d['build_stats']['sort_passes'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in d['kernels'].items() if k in ('radix_scatter','local_sort','names','radix_hist')})
"
done
timeout 800 python tools/max_n.py 2>&1 | tail -2 | cut -c1-400
