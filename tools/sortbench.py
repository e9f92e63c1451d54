"""micro-benchmark of sx_prim_sort_pairs_dev: uniformly random keys, kbits wide"""
import sys, time
import torch
sys.path.insert(0, ".")
import stralg_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000_000
kbits = int(sys.argv[2]) if len(sys.argv) > 2 else 40
db = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = stralg_amd.Context(0)
ctx.set_radix_digit_bits(db)
g = torch.Generator(device="cuda"); g.manual_seed(1)
keys = torch.randint(0, 1 << kbits, (n,), dtype=torch.int64, device="cuda", generator=g)
vals = torch.arange(n, dtype=torch.int32, device="cuda")
ka, va = keys.clone(), vals.clone()
kb, vb = torch.empty_like(ka), torch.empty_like(va)
for it in range(3):
    ka.copy_(keys); va.copy_(vals)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    in_b = ctx.prim_sort_pairs_dev(ka, va, kb, vb, n, 0, kbits)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
res = kb if in_b else ka
ok = bool((res[1:] >= res[:-1]).all())
passes = (kbits + db - 1) // db
print(f"n={n} kbits={kbits} digit_bits={db} passes={passes} time={dt*1e3:.2f} ms  per pass {dt*1e3/passes:.2f} ms  "
      f"{n*24*passes/dt/1e9:.0f} GB/s(alg, scatter only)  sorted={ok}")
