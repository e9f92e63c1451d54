"""where the O table lands and how fast it is written: the same kernel (1 GiB of DNA: 20 GiB of rows) over buffers from
successive allocations and at offsets inside one, beside a plain fill of the same buffer (sx_membw_probe) -- round 5's finding:
the O-table kernel's time follows the allocation (4.0 ... 4.65 ms, host-timed), not the offset, the BWT's place or the time since
the allocation, while the plain fill does not care (profiles/r05_buffer_placement.txt)"""
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

def run(o, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c_tab, o)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3

for trial in range(6):
    pad = torch.empty(int(trial * 777) << 20, dtype=torch.uint8, device=dev) if trial else None
    big = torch.empty(rows + (64 << 20), dtype=torch.int32, device=dev)
    base = big.data_ptr()
    line = [f"allocation {trial}: base {base:#x}"]
    for off_words in (0, 1 << 10, (1 << 19) + 5 * 256, 1 << 24):
        line.append(f"+{off_words * 4 >> 10} KiB: {run(big[off_words:off_words + rows]):.2f} ms")
    u8 = big.view(torch.uint8)
    half = 8 << 30
    r = ctx.membw_probe(u8[:half], u8[half:2 * half], half, 3)
    line.append("plain fill of its first 16 GiB: %.0f GB/s, copy %.0f" % (r["fill"], r["copy"]))
    print("  ".join(line), flush=True)
    del big, pad
    torch.cuda.empty_cache()
