"""times the suffix-array build on adversarial inputs (GPU box)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import stralg_amd, oracle
ctx = stralg_amd.Context(0)
rng = np.random.default_rng(1)
def run(name, x, sigma, check=True):
    ctx.sa_build(x[:1000], sigma)
    t0 = time.perf_counter(); sa = ctx.sa_build(x, sigma); dt = time.perf_counter() - t0
    st = ctx.last_stats()
    ok = oracle.check_sa(x, sa) if check else None
    print(f"{name:28s} n={x.size:10d} {dt*1e3:10.1f} ms  path={st['lms_path']} rounds={st['induce_rounds']} dbl={st['doubling_rounds']} ok={ok}", flush=True)
n = 1 << 22
run("random dna", oracle.synth(n, 5, 1), 5)
run("all equal 1M", np.full(1 << 20, 1, np.uint8), 2)
run("dna with one 1M N-run", np.concatenate([oracle.synth(n // 2, 5, 2), np.full(1 << 20, 5, np.uint8), oracle.synth(n // 2, 5, 3)]), 6)
run("dna with 1000 runs of 1000", np.concatenate([np.concatenate([oracle.synth(3000, 5, 10 + i), np.full(1000, 5, np.uint8)]) for i in range(1000)]), 6)
run("period 97 x 40000", np.tile(rng.integers(1, 5, size=97, dtype=np.uint8), 40000), 5)
run("fibonacci 3.5M", (lambda: (lambda f: np.array(f, np.uint8))(__import__('functools').reduce(lambda ab, _: (ab[1], ab[1] + ab[0]), range(31), ([1], [1, 2]))[1]))(), 3)
run("two copies of 2M random", np.tile(oracle.synth(n // 2, 5, 4), 2), 5)
run("english-like (skewed)", rng.choice(np.arange(1, 28, dtype=np.uint8), size=n, p=np.array([8.2,1.5,2.8,4.3,12.7,2.2,2.0,6.1,7.0,0.15,0.77,4.0,2.4,6.7,7.5,1.9,0.1,6.0,6.3,9.1,2.8,0.98,2.4,0.15,2.0,0.07,18.0])/np.sum([8.2,1.5,2.8,4.3,12.7,2.2,2.0,6.1,7.0,0.15,0.77,4.0,2.4,6.7,7.5,1.9,0.1,6.0,6.3,9.1,2.8,0.98,2.4,0.15,2.0,0.07,18.0])).astype(np.uint8), 28)
