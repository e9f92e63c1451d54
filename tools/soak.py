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
for log2n, sigma in ((30, 5), (28, 5), (28, 256), (27, 21), (24, 5)):
    n = 1 << log2n
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.synth_dev(text, n, sigma, 1234)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
    seen = set()
    t0 = time.time()
    for r in range(reps):
        sa.zero_(); bw.zero_()
        ctx.sa_bwt_build_dev(text, n, sigma, sa, bw)
        seen.add((digest(sa), digest(bw)))
    print(f"2^{log2n} sigma={sigma}: {reps} builds, {len(seen)} distinct result(s), {time.time()-t0:.1f} s", flush=True)
    assert len(seen) == 1
    del text, sa, bw
print("soak ok")
