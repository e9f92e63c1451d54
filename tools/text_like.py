"""a natural-language-like text: Zipf-distributed words from a fixed vocabulary, separated by blanks (27 symbols,
heavy short repeats).  Reports path and time; checks the suffix array on the device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stralg_amd
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log2n
rng = np.random.default_rng(9)
vocab = [rng.integers(2, 28, size=int(rng.integers(2, 11)), dtype=np.uint8) for _ in range(20000)]
ranks = rng.zipf(1.3, size=n // 4) % len(vocab)
parts, total = [], 0
for r in ranks:
    w = vocab[int(r)]
    parts.append(w); parts.append(np.array([1], dtype=np.uint8))
    total += len(w) + 1
    if total >= n:
        break
x = np.concatenate(parts)[:n]
sigma = 28
ctx = stralg_amd.Context(0)
text = torch.from_numpy(x).cuda()
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.sa_build_dev(text, n, sigma, sa)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = ctx.last_stats()
print(f"text-like 2^{log2n}: {dt*1e3:.1f} ms = {n/dt/1e6:.0f} Msuffixes/s  path={st['lms_path']} refinement rounds={st['doubling_rounds']} "
      f"induce rounds={st['induce_rounds']} key symbols={st['key_slots']}")
N = n + 1
pos = sa.long() & 0xFFFFFFFF
rank = torch.full((N + 1,), -1, dtype=torch.int64, device="cuda")
rank[pos] = torch.arange(N, dtype=torch.int64, device="cuda")
assert bool((rank[:N] >= 0).all())
T = torch.zeros(N + 1, dtype=torch.uint8, device="cuda"); T[:n] = text
a, b = pos[1:N - 1], pos[2:N]
ok = (T[a] < T[b]) | ((T[a] == T[b]) & (rank[a + 1] < rank[b + 1]))
print("suffix array verified" if bool(ok.all()) and int(pos[0]) == n else "WRONG")
