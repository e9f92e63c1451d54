"""device memory bandwidth reference points (torch fill / copy) for the roofline discussion"""
import time, torch
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
n = 5 * (1 << 30)  # 5 Gi int32 = 20 GiB
a = torch.empty(n, dtype=torch.int32, device="cuda")
dt = t(lambda: a.zero_()); print(f"fill 20 GiB: {dt*1e3:.2f} ms = {n*4/dt/1e12:.2f} TB/s written")
b = torch.empty(n // 2, dtype=torch.int32, device="cuda"); c = torch.empty_like(b)
dt = t(lambda: c.copy_(b)); print(f"copy 10 GiB: {dt*1e3:.2f} ms = {n*2*2/dt/1e12:.2f} TB/s read+written")
dt = t(lambda: b.sum()); print(f"sum 10 GiB: {dt*1e3:.2f} ms = {n*2/dt/1e12:.2f} TB/s read")
