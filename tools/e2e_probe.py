"""build_complete_table with the phase timing on ($STRALG_AMD_TIMING): forward only, then with the reverse table"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stralg_amd
from stralg_amd.synth import synth
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = stralg_amd.Context(0)
lib = ctx.lib
lib.build_complete_table.argtypes = [C.c_char_p, C.c_bool]
lib.build_complete_table.restype = C.c_void_p
lib.completely_free_bwt_table.argtypes = [C.c_void_p]
x = synth(1 << log2n, 5, 42)
letters = np.frombuffer(b"\0ACGT", dtype=np.uint8)[x].tobytes()
del x
os.environ["STRALG_AMD_TIMING"] = "1"
for rev in (False, False, True, True):
    t0 = time.perf_counter()
    t = lib.build_complete_table(letters, rev)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    lib.completely_free_bwt_table(t)
    print(f"reverse={rev}: {dt * 1e3:.1f} ms (free {1e3 * (time.perf_counter() - t1):.1f} ms)", flush=True)
