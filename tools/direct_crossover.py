"""direct sort of all suffixes against LMS sort + induction, by alphabet size (1 GiB of uniform symbols, suffix array only)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << log2n
ctx = stralg_amd.Context(0)
text = torch.empty(n, dtype=torch.uint8, device="cuda")
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
for sigma in (17, 21, 25, 29, 33, 41, 49, 65, 97, 129, 193, 256):
    ctx.synth_dev(text, n, sigma, 42)
    res = []
    for nd in (False, True):
        ctx.set_no_direct_sort(nd)
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.sa_build_dev(text, n, sigma, sa)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res.append((best * 1e3, ctx.last_stats()["lms_path"]))
    print(f"sigma {sigma:3d}: default {res[0][0]:6.1f} ms (path {res[0][1]})   induction {res[1][0]:6.1f} ms (path {res[1][1]})", flush=True)
ctx.set_no_direct_sort(False)
