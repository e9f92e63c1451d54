"""diagnostic: per-class event times of one 1 GiB DNA build, also when the build fails (a library built with -DSX_LS_SKELETON
sorts nothing in its LDS step: the induced passes then report an error; the classes before them have been timed)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stralg_amd
ctx = stralg_amd.Context(0)
n = 1 << 30
text = torch.empty(n, dtype=torch.uint8, device="cuda")
ctx.synth_dev(text, n, 5, 42)
sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
for it in range(3):
    ctx.profile_reset(); ctx.profile_only(None); ctx.profile_enable(True)
    try:
        ctx.sa_build_dev(text, n, 5, sa)
    except Exception as e:  # noqa: BLE001
        err = str(e)[:80]
    else:
        err = None
    torch.cuda.synchronize(); ctx.profile_enable(False)
    t = ctx.profile_read()
print(err, {k: round(v["ms"], 3) for k, v in t.items() if k in ("local_sort", "keys", "radix_scatter")})
