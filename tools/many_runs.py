"""thousands of runs of 2 ... 8 thousand symbols with differing lengths (just under two classification tiles, so the passes
stay unattended) in 1 GiB of DNA: what the tail kernel's closed form, its step limit and the host's carrying-on cost
   python tools/many_runs.py [log2n] [runs] [shortest longest]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stralg_amd
from stralg_amd import verify
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2000, 8000)
n = 1 << log2n
ctx = stralg_amd.Context(0)
text = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx.synth_dev(text, n, 5, 42)
rng = np.random.default_rng(3)
starts = np.sort(rng.choice(n // 8192 - 2, size=runs, replace=False)) * 8192
for a, l, c in zip(starts.tolist(), rng.integers(lo, hi, size=runs).tolist(), rng.integers(1, 5, size=runs).tolist()):
    text[a:a + l] = c
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.sa_bwt_build_dev(text, n, 5, sa, bw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = ctx.last_stats()
print(f"{runs} runs of {lo} .. {hi} symbols in 2^{log2n}: {dt * 1e3:.2f} ms; induce_redo {st['induce_redo']} long_runs {st['long_runs']} "
      f"induce_rounds {st['induce_rounds']} lms_path {st['lms_path']}")
ctx.trim()
print(verify.verify_build_on_device(text, n, 5, sa, bw, None, None))
