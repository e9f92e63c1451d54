"""short records of few symbols: the direct sort of all suffixes (SX_FLAG_SMALL_DIRECT_MAX) against classification + LMS sort +
induced passes, by length (uniform symbols, suffix array + BWT)   python tools/small_direct.py [sigma ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
sigmas = [int(a) for a in sys.argv[1:]] or [5]
ctx = stralg_amd.Context(0)
for sigma in sigmas:
    for log2n in (10, 12, 14, 16, 18, 20, 21, 22, 23, 24, 25, 26):
        n = 1 << log2n
        text = torch.empty(n, dtype=torch.uint8, device="cuda")
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
        ctx.synth_dev(text, n, sigma, 42)
        res = []
        for limit in (0, 1 << 30):
            ctx.set_small_direct_max(limit)
            best = 1e9
            for it in range(7):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ctx.sa_bwt_build_dev(text, n, sigma, sa, bw)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            res.append((best * 1e3, ctx.last_stats()["lms_path"], int(sa.to(torch.int64).sum())))
        assert res[0][2] == res[1][2]
        print(f"sigma {sigma:3d} 2^{log2n}: induction {res[0][0]:7.3f} ms (path {res[0][1]})   direct {res[1][0]:7.3f} ms (path {res[1][1]})", flush=True)
ctx.set_small_direct_max(-1)
