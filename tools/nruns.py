"""a genome-sized text with long N runs (reference assemblies have gaps of up to 30 Mbp): build time on the GPU box"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import stralg_amd, oracle
ctx = stralg_amd.Context(0)
n = 1 << 30
x = oracle.synth(n, 5, 7)
x[x == 4] = 5  # symbols 1, 2, 3, 5 = A C G T; 4 = N
for start, length in ((100_000_000, 30_000_000), (400_000_000, 3_000_000), (700_000_000, 18_000_000), (900_000_000, 50_000)):
    x[start:start + length] = 4
ctx.sa_build(x[:100000], 6)
for rep in range(2):
    if rep == 1:
        ctx.profile_reset(); ctx.profile_only(None); ctx.profile_enable(True)
    t0 = time.perf_counter(); sa = ctx.sa_build(x, 6); dt = time.perf_counter() - t0
ctx.profile_enable(False)
print({k: (v["launches"], round(v["ms"], 2)) for k, v in ctx.profile_read().items() if v["launches"]})
print(f"1 GiB DNA with N runs of 30M, 18M, 3M, 50k: {dt*1e3:.1f} ms incl. transfers, device {ctx.last_stats()['ms_total']:.1f} ms, stats {ctx.last_stats()}")
print("verified" if oracle.check_sa(x, sa) else "WRONG")
