cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
for m in 1 4 1000; do
  for l in 30 28; do
  python bench.py --no-e2e --no-cpu --no-other-configs --no-egress --no-ceiling --steps 20 --warmup 3 --log2n $l --roofline-every $m 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('every $m log2n $l:', d['ms_per_step'], r['kernel'], r['avg_ms'], r['launches'], r['timed_steps'], r['frac'])"
  done
done
done
