"""repeatability soak: the same records built many times must give bit-identical outputs (the chained kernels'
look-back and ticket order vary from run to run; the results must not)"""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = stralg_amd.Context(0)
def digest(t):
    # order-sensitive checksum on the device: sum of x[i] * (i + 1) mod 2^64 in two halves, plus a plain sum
    x = t.to(torch.int64) & 0xFFFFFFFF
    idx = torch.arange(1, x.numel() + 1, dtype=torch.int64, device=x.device)
    return int((x * idx).sum()), int(x.sum())
from stralg_amd import workloads
cases = [("dna", 30, 5, False), ("dna", 28, 5, False), ("bytes", 28, 256, False), ("uniform", 27, 21, False), ("dna", 24, 5, False),
         ("genome_like", 28, 5, False), ("n_runs", 28, 6, False), ("uniform", 26, 12, True), ("bytes", 26, 256, True), ("text_like", 26, 28, False), ("pangenome", 27, 5, False)]
for gen, log2n, sigma, no_direct in cases:
    n = 1 << log2n
    ctx.set_no_direct_sort(no_direct)
    text, sigma = workloads.make_text(ctx, gen, n, sigma, 1234, torch.device("cuda", 0))
    torch.cuda.synchronize()
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
    seen = set()
    t0 = time.time()
    for r in range(reps):
        sa.zero_(); bw.zero_()
        ctx.sa_bwt_build_dev(text, n, sigma, sa, bw)
        seen.add((digest(sa), digest(bw)))
    print(f"{gen} 2^{log2n} sigma={sigma}{' (induced passes)' if no_direct else ''}: {reps} builds, {len(seen)} distinct result(s), "
          f"{time.time()-t0:.1f} s", flush=True)
    assert len(seen) == 1
    del text, sa, bw
ctx.set_no_direct_sort(False)
print("soak ok")
