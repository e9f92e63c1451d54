"""a DNA text with genome-like structure instead of uniform noise: biased base composition, a family of
diverged interspersed repeats, exact duplications of several lengths, microsatellites and poly-A runs.
Reports which LMS path the build takes and how long it needs; checks the result on the device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stralg_amd
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
rng = np.random.default_rng(5)
x = rng.choice(np.array([1, 2, 3, 4], dtype=np.uint8), size=n, p=[0.3, 0.2, 0.2, 0.3])
alu = rng.integers(1, 5, size=300, dtype=np.uint8)
for pos in rng.integers(0, n - 400, size=n // 3000):          # ~10 % of the text: diverged copies of one element
    copy = alu.copy()
    mut = rng.random(300) < 0.08
    copy[mut] = rng.integers(1, 5, size=int(mut.sum()), dtype=np.uint8)
    x[pos:pos + 300] = copy
for L, count in ((100, 2000), (1000, 300), (6000, 40), (50000, 2)):  # exact duplications
    for _ in range(count):
        a, b = rng.integers(0, n - L - 1, size=2)
        x[b:b + L] = x[a:a + L]
for pos in rng.integers(0, n - 400, size=n // 20000):          # microsatellites and poly-A
    unit = rng.integers(1, 5, size=int(rng.integers(1, 5)), dtype=np.uint8)
    L = int(rng.integers(20, 200))
    x[pos:pos + L] = np.resize(unit, L)
ctx = stralg_amd.Context(0)
text = torch.from_numpy(x).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.sa_build_dev(text, n, 5, sa)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = ctx.last_stats()
ctx.profile_reset(); ctx.profile_enable(True)
ctx.sa_build_dev(text, n, 5, sa)
torch.cuda.synchronize(); ctx.profile_enable(False)
print({k: round(v["ms"], 2) for k, v in ctx.profile_read().items() if v["launches"]})
print(f"genome-like 2^{log2n}: {dt*1e3:.1f} ms = {n/dt/1e6:.0f} Msuffixes/s  path={st['lms_path']} refinement rounds={st['doubling_rounds']} "
      f"key symbols={st['key_slots']} told apart by the first sort={st['n_names']}/{st['n_lms']}")
# check on the device: permutation + neighbouring suffixes in order (rank of the next suffix decides ties of the first symbol)
N = n + 1
pos = sa.long() & 0xFFFFFFFF
rank = torch.full((N + 1,), -1, dtype=torch.int64, device="cuda")
rank[pos] = torch.arange(N, dtype=torch.int64, device="cuda")
assert bool((rank[:N] >= 0).all())
T = torch.zeros(N + 1, dtype=torch.uint8, device="cuda"); T[:n] = text
a, b = pos[1:N - 1], pos[2:N]
ok = (T[a] < T[b]) | ((T[a] == T[b]) & (rank[a + 1] < rank[b + 1]))
print("suffix array verified" if bool(ok.all()) and int(pos[0]) == n else "WRONG")
