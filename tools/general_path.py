"""time the general LMS path (pieces + names + prefix doubling) against the prefix-key path on the same record"""
import sys, time
import torch
sys.path.insert(0, ".")
import stralg_amd
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << log2n
ctx = stralg_amd.Context(0)
text = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx.synth_dev(text, n, sigma, 42)
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
for forced in (0, 1):
    ctx.force_general_path(bool(forced))
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.sa_build_dev(text, n, sigma, sa)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = ctx.last_stats()
    print(f"log2n={log2n} sigma={sigma} forced_general={forced}: {dt*1e3:.1f} ms  path={st['lms_path']} doubling_rounds={st['doubling_rounds']} passes={st['sort_passes']}")
