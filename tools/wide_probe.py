"""diagnostic: phase cycles of the wide-alphabet induce scatter (a library built with -DSX_WIDE_PROBE prints them)
   make -C stralg_amd/csrc OBJDIR=$PWD/stralg_amd/csrc/build_probe OUT=$PWD/tools/ref/libstralg_amd_probe.so HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -DSX_WIDE_PROBE"
   python tools/wide_probe.py [log2n] [sigma]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import stralg_amd  # noqa: E402
from stralg_amd import workloads  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = stralg_amd.Context(0, lib_path=os.path.join(ROOT, "tools", "ref", "libstralg_amd_probe.so"))
ctx.set_no_direct_sort(True)
dev = torch.device("cuda", 0)
n = 1 << log2n
text, sig = workloads.make_text(ctx, "bytes" if sigma == 256 else "uniform", n, sigma, 42, dev)
sa = torch.empty(n + 1, dtype=torch.int32, device=dev)
for _ in range(2):
    ctx.sa_build_dev(text, n, sig, sa)
torch.cuda.synchronize()
print(ctx.last_stats())
