import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import stralg_amd
ctx = stralg_amd.Context(0)
for sigma, sizes in ((5, (24, 25)), (8, (24, 25, 26)), (16, (25, 26, 27)), (12, (26, 27))):
    for log2n in sizes:
        n = 1 << log2n
        text = torch.empty(n, dtype=torch.uint8, device="cuda")
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
        ctx.synth_dev(text, n, sigma, 42)
        res = []
        for limit in (0, -1, 1 << 30):
            ctx.set_small_direct_max(limit)
            best = 1e9
            for it in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ctx.sa_bwt_build_dev(text, n, sigma, sa, bw)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            res.append((round(best * 1e3, 3), ctx.last_stats()["lms_path"]))
        print(f"sigma {sigma} 2^{log2n}: SA-IS {res[0]}  default {res[1]}  direct {res[2]}", flush=True)
