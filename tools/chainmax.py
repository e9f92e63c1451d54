"""sweep of the round size below which an induce round takes the single chained launch (SX_FLAG_CHAIN_MAX_ENTRIES)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << log2n
ctx = stralg_amd.Context(0)
text = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx.synth_dev(text, n, 5, 42)
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
for tiles in (64, 256, 1024, 4096, 16384):
    ctx.set_chain_max_entries(tiles * 2048)
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.sa_bwt_build_dev(text, n, 5, sa, bw)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"chain_max {tiles:6d} tiles: SA + BWT {best*1e3:.2f} ms")
