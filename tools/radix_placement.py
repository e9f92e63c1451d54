"""the plain radix passes (313 M pairs of 8-byte key + 4-byte value, two 8-bit passes: the 1 GiB DNA step's) over buffers from
separate allocations and over four pieces of one allocation at chosen distances: does where the two sides of the ping-pong lie
relative to each other decide the 2.8 / 3.3 ms?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
ctx = stralg_amd.Context(0)
dev = torch.device("cuda:0")
m = 313_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
keys0 = torch.randint(0, 1 << 40, (m,), dtype=torch.int64, device=dev, generator=g)
vals0 = torch.arange(m, dtype=torch.int32, device=dev)

def run(ka, va, kb, vb, reps=3):
    best = 1e9
    for _ in range(reps):
        ka.copy_(keys0); va.copy_(vals0)
        torch.cuda.synchronize()
        ctx.profile_reset(); ctx.profile_only("radix_scatter"); ctx.profile_enable(True)
        ctx.prim_sort_pairs_dev(ka, va, kb, vb, m, 16, 32)
        torch.cuda.synchronize()
        ctx.profile_enable(False)
        best = min(best, ctx.profile_read()["radix_scatter"]["ms"])
    return best

print("separate allocations:", flush=True)
for trial in range(5):
    pad = torch.empty((trial * 389) << 20, dtype=torch.uint8, device=dev) if trial else None
    ka = torch.empty(m, dtype=torch.int64, device=dev); va = torch.empty(m, dtype=torch.int32, device=dev)
    kb = torch.empty(m, dtype=torch.int64, device=dev); vb = torch.empty(m, dtype=torch.int32, device=dev)
    print(f"  trial {trial}: two passes {run(ka, va, kb, vb):.2f} ms", flush=True)
    del ka, va, kb, vb, pad
    torch.cuda.empty_cache()
print("one allocation, the four arrays at distances (ka at 0; va, kb, vb behind it with a gap each):", flush=True)
kbytes, vbytes = 8 * m, 4 * m
for trial in range(2):
    big = torch.empty(2 * kbytes + 2 * vbytes + (2 << 30), dtype=torch.uint8, device=dev)
    line = []
    for gap in (0, 4 << 10, 64 << 10, 1 << 20, (2 << 20) + (4 << 10), 33 << 20, (256 << 20) + (1 << 20)):
        def al(x): return (x + 255) & ~255
        o1 = al(kbytes + gap); o2 = al(o1 + vbytes + gap); o3 = al(o2 + kbytes + gap)
        ka = big[0:kbytes].view(torch.int64); va = big[o1:o1 + vbytes].view(torch.int32)
        kb = big[o2:o2 + kbytes].view(torch.int64); vb = big[o3:o3 + vbytes].view(torch.int32)
        line.append(f"gap {gap >> 10} KiB: {run(ka, va, kb, vb):.2f}")
    print(f"  allocation {trial}: " + "  ".join(line), flush=True)
    del big, ka, va, kb, vb
    torch.cuda.empty_cache()
