"""one radix pass with (a) uniformly random digits, (b) a constant digit: separates the cost of the
scattered write granularity from the rest of the scatter kernel"""
import sys, time
import torch
sys.path.insert(0, ".")
import stralg_amd
n = 300_000_000
ctx = stralg_amd.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
base = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device="cuda", generator=g)
vals = torch.arange(n, dtype=torch.int32, device="cuda")
for name, keys in (("random digit", base), ("constant digit", (base >> 8) << 8), ("16 distinct digits", ((base >> 8) << 8) | (base & 15))):
    ka, va = keys.clone(), vals.clone(); kb, vb = torch.empty_like(ka), torch.empty_like(va)
    ctx.profile_reset(); ctx.profile_enable(True)
    for it in range(3):
        ctx.prim_sort_pairs_dev(ka, va, kb, vb, n, 0, 8)
    ctx.profile_enable(False); p = ctx.profile_read()
    print(f"{name:20s}: hist {p['radix_hist']['ms']/3:.2f} ms  scan {p['scan']['ms']/3:.2f} ms  scatter {p['radix_scatter']['ms']/3:.2f} ms "
          f"= {n*24/(p['radix_scatter']['ms']/3*1e-3)/1e12:.2f} TB/s")
