"""for six successive 20 GiB allocations: the O-table kernel's time over it, and the time of 256 M random 4-byte reads (and of as many
random 4-byte writes) spread over its first 16 GiB -- does the placement that slows the many-window writer also slow plain random
accesses (translations), or only that writer?"""
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
g = torch.Generator(device=dev); g.manual_seed(3)
idx = torch.randint(0, 1 << 32, (1 << 28,), dtype=torch.int64, device=dev, generator=g)  # 4 G words = 16 GiB
src = torch.arange(1 << 28, dtype=torch.int32, device=dev)
out = torch.empty(1 << 28, dtype=torch.int32, device=dev)

def best(f, reps=3):
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0)
    return b * 1e3

for trial in range(6):
    pad = torch.empty(int(trial * 777) << 20, dtype=torch.uint8, device=dev) if trial else None
    big = torch.empty(rows + (64 << 20), dtype=torch.int32, device=dev)
    t_o = best(lambda: ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c_tab, big[:rows]))
    w = big[:1 << 32]
    t_r = best(lambda: torch.index_select(w, 0, idx, out=out))
    t_w = best(lambda: w.index_copy_(0, idx, src))
    print(f"allocation {trial}: O table {t_o:.2f} ms   256 M random reads {t_r:.2f} ms   256 M random writes {t_w:.2f} ms", flush=True)
    del big, pad, w
    torch.cuda.empty_cache()
