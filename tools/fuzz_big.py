"""differential fuzz at sizes where the hybrid sort, the dense keys and the two-pass scatters are taken (15 M ... 40 M symbols,
four-letter texts mostly), against the oracle"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import stralg_amd, oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = stralg_amd.Context(0)
t0 = time.time(); paths = {}
for k in range(cases):
    sigma = int(rng.choice([5, 5, 5, 5, 6, 3, 21, 256]))
    n = int(rng.choice([15_000_000, 17_000_001, 24_000_000, 33_554_433, 40_000_000]))
    x = rng.integers(1, sigma, size=n, dtype=np.uint8) if sigma > 2 else np.ones(n, np.uint8)
    kind = int(rng.integers(0, 6))
    if kind == 1:    # planted repeats
        for _ in range(int(rng.integers(1, 30))):
            L = int(rng.integers(20, 200000)); a, b = rng.integers(0, n - L, size=2); x[b:b + L] = x[a:a + L]
    elif kind == 2:  # runs
        for _ in range(int(rng.integers(1, 30))):
            L = int(rng.integers(1, 50000)); a = int(rng.integers(0, n - L)); x[a:a + L] = x[a]
    elif kind == 3:  # a family of diverged repeats
        L = int(rng.integers(50, 400)); el = x[:L].copy()
        for pos in rng.choice(n // L - 1, size=int(rng.integers(100, 20000)), replace=False) * L:
            c = el.copy(); mm = rng.random(L) < 0.02
            c[mm] = rng.integers(1, sigma, size=int(mm.sum()), dtype=np.uint8); x[pos:pos + L] = c
    elif kind == 4:  # copies of the first eighth
        p = n // 8
        for c in range(1, int(rng.integers(2, 8))): x[c * p:(c + 1) * p] = x[:p]
    elif kind == 5 and sigma >= 5:  # genome-like: skewed symbol counts, a tenth of the text one family of diverged repeats, microsatellites
        x = rng.choice(np.arange(1, 5, dtype=np.uint8), size=n, p=[0.3, 0.2, 0.2, 0.3]) if sigma == 5 else x
        L = int(rng.integers(100, 400)); el = rng.integers(1, min(sigma, 5), size=L, dtype=np.uint8)
        for pos in rng.integers(0, n - L - 1, size=n // (10 * L)):
            c = el.copy(); mm = rng.random(L) < float(rng.choice([0.0, 0.02, 0.08]))
            c[mm] = rng.integers(1, min(sigma, 5), size=int(mm.sum()), dtype=np.uint8); x[pos:pos + L] = c
        for pos in rng.integers(0, n - 400, size=n // 20000):
            unit = rng.integers(1, min(sigma, 5), size=int(rng.integers(1, 5)), dtype=np.uint8)
            R = int(rng.integers(20, 200)); x[pos:pos + R] = np.resize(unit, R)
    ctx.set_sort_mode(int(rng.choice([0, 0, 0, 1, 2, 3])))
    ctx.force_general_path(bool(rng.integers(0, 6) == 0))
    ctx.set_no_direct_sort(bool(rng.integers(0, 3) == 0))
    ctx.set_text_keys(bool(rng.integers(0, 4)))
    ctx.set_local_sort_lean(bool(rng.integers(0, 5)))  # (round 5: crowded bins by the lean kernel's waves, or the other kernel for all)
    want = oracle.sa_is(x, sigma)
    got = ctx.sa_build(x, sigma)
    st = ctx.last_stats()
    pk = (st["lms_path"], st["sort_local"], st["key_bits"], st["refine_tiers"])
    paths[pk] = paths.get(pk, 0) + 1
    assert (got == want).all(), ("SA", k, sigma, n, kind, st)
    if k % 10 == 9: print(f"{k + 1} cases, {time.time() - t0:.0f} s", flush=True)
ctx.set_sort_mode(0); ctx.force_general_path(False); ctx.set_no_direct_sort(False); ctx.set_text_keys(True); ctx.set_local_sort_lean(True)
print(f"{cases} cases ok in {time.time() - t0:.0f} s; (lms_path, sort_local, key_bits, refine_tiers): {paths}")
