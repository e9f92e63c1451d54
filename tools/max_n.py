"""the largest text the reference's uint32_t length allows (n = 2^32 - 2): suffix array on one GPU, checked on the device"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 32) - 2
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = stralg_amd.Context(0)
if len(sys.argv) > 3:
    ctx.set_sort_mode(int(sys.argv[3]))  # (3: four HBM passes on the top 32 key bits, then sub-buckets in LDS)
text = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx.synth_dev(text, n, sigma, 7)
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
for call in range(2):  # (the first call allocates the workspace)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.sa_build_dev(text, n, sigma, sa)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"call {call}: {dt*1e3:.1f} ms", flush=True)
print(f"n = {n}: {dt*1e3:.1f} ms = {n/dt/1e6:.0f} Msuffixes/s, stats {ctx.last_stats()}", flush=True)
ctx.trim()
N = n + 1
assert int(sa[0]) & 0xFFFFFFFF == n
pos = sa.long() & 0xFFFFFFFF
del sa
rank = torch.full((N + 1,), -1, dtype=torch.int64, device="cuda")
for s0 in range(0, N, 1 << 30):  # (one scatter of 2^32 elements exceeds torch's own launch limits)
    e0 = min(N, s0 + (1 << 30))
    rank[pos[s0:e0]] = torch.arange(s0, e0, dtype=torch.int64, device="cuda")
assert bool((rank[:N] >= 0).all()), "not a permutation"
T = torch.zeros(N + 1, dtype=torch.uint8, device="cuda")
for s0 in range(0, n, 1 << 30):
    T[s0:min(n, s0 + (1 << 30))] = text[s0:min(n, s0 + (1 << 30))]
del text
step = 1 << 28
for s0 in range(1, N - 1, step):
    e0 = min(N - 1, s0 + step)
    a, b = pos[s0:e0], pos[s0 + 1:e0 + 1]
    ca, cb = T[a], T[b]
    ok = (ca < cb) | ((ca == cb) & (rank[a + 1] < rank[b + 1]))
    assert bool(ok.all()), f"suffixes out of order in slots [{s0}, {e0})"
print("suffix array verified: permutation, strictly increasing suffixes")
