"""end-to-end rate through the host-buffer entry point (what build_complete_table pays): H2D text,
device build, D2H suffix array + C + O.  Not bench.py's `value` (that one starts from HBM)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import stralg_amd
ctx = stralg_amd.Context(0)
for log2n in (24, 26, 28):
    n = 1 << log2n
    x = stralg_amd.synth(n, 5, 42)
    ctx.build_tables(x[: 1 << 16], 5)
    t0 = time.perf_counter(); sa, c, o = ctx.build_tables(x, 5); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); sa2 = ctx.sa_build(x, 5); dt2 = time.perf_counter() - t1
    print(f"2^{log2n}: sx_build_tables {dt*1e3:.1f} ms = {(n+1)/dt/1e6:.0f} Msuffixes/s (moves {(n + 4*(n+1) + 20*(n+2))/1e9:.2f} GB over PCIe); "
          f"sx_sa_build {dt2*1e3:.1f} ms = {(n+1)/dt2/1e6:.0f} Msuffixes/s", flush=True)
    del sa, o, sa2
