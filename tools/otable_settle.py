"""does the O-table kernel's time over a fresh 20 GiB buffer settle with time?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
ctx = stralg_amd.Context(0)
dev = torch.device("cuda:0")
n = 1 << 30; N = n + 1; sigma = 5
bwt = torch.randint(1, 5, (N,), dtype=torch.uint8, device=dev)
c_tab = torch.zeros(sigma, dtype=torch.int32, device=dev)
rows = (N + 1) * sigma
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for trial in range(3):
    t_alloc = time.perf_counter()
    big = torch.empty(rows + (64 << 20), dtype=torch.int32, device=dev)
    out = []
    for k in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c_tab, big[:rows])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out.append("%.0f:%.2f" % ((t0 - t_alloc) * 1e3, dt * 1e3))
        if k >= 10: time.sleep(0.1)
    print(f"trial {trial} (ms since allocation : ms):", " ".join(out), flush=True)
    del big
    torch.cuda.empty_cache()
