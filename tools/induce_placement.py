"""do the induce scatters' two speeds (4.95 or 5.6 ms a 1 GiB step) follow the allocations too?  Six contexts one after the
other in one process, each with fresh slabs and a fresh suffix-array buffer: the per-class table of one profiled step"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
dev = torch.device("cuda:0")
n = 1 << 30; N = n + 1; sigma = 5
ctx0 = stralg_amd.Context(0)
text = torch.empty(n, dtype=torch.uint8, device=dev)
ctx0.synth_dev(text, n, sigma, 42)
ctx0.close()
for trial in range(6):
    pad = torch.empty((trial * 611) << 20, dtype=torch.uint8, device=dev) if trial else None
    ctx = stralg_amd.Context(0)
    sa = torch.empty(N, dtype=torch.int32, device=dev)
    bwt = torch.empty(N, dtype=torch.uint8, device=dev)
    for _ in range(2):
        ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
    torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_only(None); ctx.profile_enable(True)
    ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    tab = ctx.profile_read()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"context {trial}: step {ms:.2f} ms  " + "  ".join(f"{k} {v['ms']:.2f}" for k, v in tab.items() if v["ms"] > 0.3), flush=True)
    ctx.close()
    del sa, bwt, pad
    torch.cuda.empty_cache()
