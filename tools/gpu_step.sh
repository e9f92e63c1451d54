#!/bin/bash
# One GPU-box session: the named steps, in order, each logging under gpurun_out/<tag>/ (tools/gpu_step.sh TAG step...).
#   tests[:K_EXPR]   pytest -m gpu (optionally -k K_EXPR)
#   bench[:ARGS]     python bench.py ARGS > bench[_<n>].json
#   bench2           two ranks sharing the one GPU (gloo carries the scalars): the N > 1 path of bench.py
#   bench2n          the same with RCCL tried first (refused for two ranks on one GPU: the fall-back to gloo, on hardware)
#   prof             rocprofv3 --kernel-trace --stats of bench.py (summary via tools/profile_summary.py)
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
k=0
for step in "$@"; do
  k=$((k+1))
  name=${step%%:*}; arg=""
  [[ "$step" == *:* ]] && arg=${step#*:}
  echo "== $step" | tee -a "$out/steps.log"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "$arg" > "$out/tests_$k.log" 2>&1
      else timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$out/tests_$k.log" 2>&1; fi
      rc=$?; tail -5 "$out/tests_$k.log";;
    bench)
      timeout -k 10 900 python bench.py $arg > "$out/bench_$k.json" 2> "$out/bench_$k.err"
      rc=$?; tail -c 600 "$out/bench_$k.err"; head -c 400 "$out/bench_$k.json"; echo;;
    bench2)
      STRALG_BENCH_BACKEND=gloo STRALG_BENCH_SHARE_GPU=1 timeout -k 10 900 python bench.py --gpus 2 $arg > "$out/bench2_$k.json" 2> "$out/bench2_$k.err"
      rc=$?; tail -c 600 "$out/bench2_$k.err"; head -c 300 "$out/bench2_$k.json"; echo;;
    bench2n)
      # the same with the default backend: RCCL is tried first; two ranks on ONE GPU are refused by RCCL ("duplicate GPU"),
      # which is the fall-back branch of farm.init_collectives on real hardware (the line then says collective_backend gloo)
      STRALG_BENCH_SHARE_GPU=1 timeout -k 10 900 python bench.py --gpus 2 $arg > "$out/bench2n_$k.json" 2> "$out/bench2n_$k.err"
      rc=$?; tail -c 600 "$out/bench2n_$k.err"; head -c 300 "$out/bench2n_$k.json"; echo;;
    trace)
      # rocprofv3 kernel trace of bench.py ARGS (python3 directly after --), per-kernel stats and the last step's timeline
      rm -rf "$out/trace_$k"; mkdir -p "$out/trace_$k"
      ( cd /tmp && TMPDIR=/tmp timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/trace_$k/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" $arg > "$GRAFT_REPO_ROOT/$out/trace_$k.log" 2>&1 )
      rc=$?
      python3 tools/profile_summary.py "$out/trace_$k" > "$out/trace_${k}_summary.txt" 2>&1
      python3 tools/step_timeline.py "$out/trace_$k" > "$out/trace_${k}_timeline.txt" 2>&1
      tail -3 "$out/trace_${k}_timeline.txt"
      find "$out/trace_$k" -name "*.csv" -size +8M -delete;;
    prof)
      # rocprofv3 of `python3 ARG...` (bench.py or a tool): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in passes of
      # their own (PMC passes never share a run with a trace: the MI355X guide's HBM section), one summary
      pd="$out/prof_$k"; rm -rf "$pd"; mkdir -p "$pd"
      ( cd /tmp && export TMPDIR=/tmp
        timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$pd/trace" -- python3 $GRAFT_REPO_ROOT/$arg > "$GRAFT_REPO_ROOT/$pd/trace.log" 2>&1 &&
        timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/$pd/pmc_fetch" -- python3 $GRAFT_REPO_ROOT/$arg > "$GRAFT_REPO_ROOT/$pd/pmc_fetch.log" 2>&1 &&
        timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/$pd/pmc_write" -- python3 $GRAFT_REPO_ROOT/$arg > "$GRAFT_REPO_ROOT/$pd/pmc_write.log" 2>&1 )
      rc=$?
      python3 tools/profile_summary.py "$pd" > "$pd/summary.txt" 2>&1
      tail -4 "$pd/trace.log"; head -12 "$pd/summary.txt"
      find "$pd" -name "*.csv" -size +6M -delete;;
    py)
      timeout -k 10 900 python $arg > "$out/py_$k.log" 2>&1
      rc=$?; tail -20 "$out/py_$k.log";;
    *) echo "unknown step $name"; rc=64;;
  esac
  echo "   rc=$rc" | tee -a "$out/steps.log"
  [ $rc -ne 0 ] && exit $rc
done
exit 0
