"""the C batch farm on collections of short records: stralg_amd_build_tables_batch (host buffers in, malloc'd tables out)
with one worker a device and with the default (up to four)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stralg_amd
from stralg_amd.synth import synth
ctx = stralg_amd.Context(0)
lib = ctx.lib
lib.stralg_amd_build_tables_batch.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_bool, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
lib.stralg_amd_build_tables_batch.restype = C.c_int
lib.completely_free_bwt_table.argtypes = [C.c_void_p]
letters = np.frombuffer(b"\0ACGT", dtype=np.uint8)
devs = (C.c_int * 1)(0)
for log2n, count in ((13, 512), (16, 512), (18, 256), (20, 128), (22, 48), (24, 12)):
    n = 1 << log2n
    raws = [letters[synth(n, 5, 40 + (i % 4))].tobytes() for i in range(min(count, 4))]
    arr = (C.c_char_p * count)(*[raws[i % len(raws)] for i in range(count)])
    line = f"{count} records of 2^{log2n}:"
    for workers in ("1", "2", "4", "8"):
        os.environ["STRALG_AMD_FARM_WORKERS"] = workers
        best = 1e9
        for _ in range(2):
            out = (C.c_void_p * count)()
            t0 = time.perf_counter()
            rc = lib.stralg_amd_build_tables_batch(arr, count, False, devs, 1, out)
            dt = time.perf_counter() - t0
            assert rc == 0
            for t in out: lib.completely_free_bwt_table(t)
            best = min(best, dt)
        line += f"  {workers} worker(s): {best / count * 1e3:7.3f} ms/record = {count * (n + 1) / best / 1e6:8.1f} Msuffixes/s"
    print(line, flush=True)
