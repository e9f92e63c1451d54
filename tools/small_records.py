"""how long one record takes as it gets short: sx_sa_bwt_build_dev + sx_bwt_tables_from_bwt_dev on resident data, and
build_complete_table (host buffers) -- the regime of assemblies with thousands of contigs"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stralg_amd
from stralg_amd.synth import synth
ctx = stralg_amd.Context(0)
lib = ctx.lib
lib.build_complete_table.argtypes = [C.c_char_p, C.c_bool]; lib.build_complete_table.restype = C.c_void_p
lib.completely_free_bwt_table.argtypes = [C.c_void_p]
for log2n in (10, 13, 16, 18, 20, 22, 24, 26):
    n = 1 << log2n; N = n + 1
    text = torch.empty(n, dtype=torch.uint8, device="cuda"); ctx.synth_dev(text, n, 5, 7)
    sa = torch.empty(N, dtype=torch.int32, device="cuda"); bw = torch.empty(N, dtype=torch.uint8, device="cuda")
    c = torch.zeros(5, dtype=torch.int32, device="cuda"); o = torch.empty((N + 1) * 5, dtype=torch.int32, device="cuda")
    reps = 20 if log2n <= 20 else 5
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            ctx.sa_bwt_build_dev(text, n, 5, sa, bw); ctx.bwt_tables_from_bwt_dev(bw, N, 5, c, o)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    letters = np.frombuffer(b"\0ACGT", dtype=np.uint8)[synth(n, 5, 7)].tobytes()
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(reps):
            t = lib.build_complete_table(letters, False); lib.completely_free_bwt_table(t)
        dh = (time.perf_counter() - t0) / reps
    print(f"n = 2^{log2n}: resident {dt*1e3:8.3f} ms = {N/dt/1e6:9.1f} Msuffixes/s; build_complete_table {dh*1e3:8.3f} ms = {N/dh/1e6:8.1f} Msuffixes/s", flush=True)
