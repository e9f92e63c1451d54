"""the inverse of a random permutation of 2^k + 1 entries (sx_sa_inverse_dev: two passes from 2^23 entries on), checked"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stralg_amd
ctx = stralg_amd.Context(0)
for log2n in (22, 24, 28, 30):
    N = (1 << log2n) + 1
    sa = torch.randperm(N, device="cuda").to(torch.int32)
    inv = torch.empty(N, dtype=torch.int32, device="cuda")
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.sa_inverse_dev(sa, N, inv); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = bool((inv[sa.long()] == torch.arange(N, dtype=torch.int32, device="cuda")).all())
    print(f"2^{log2n}: {dt*1e3:.2f} ms = {8 * N / dt / 1e9:.0f} GB/s of the 8 B a position it has to move; ok={ok}", flush=True)
    del sa, inv
