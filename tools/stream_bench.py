"""stralg_amd_write_complete_bwt_info_stream to /dev/null: build on the device + stream SA and O table through pinned chunks"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stralg_amd
from stralg_amd.synth import synth
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
ctx = stralg_amd.Context(0)  # loads the library (torch first)
lib = ctx.lib
libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]; libc.fclose.argtypes = [C.c_void_p]
lib.stralg_amd_write_complete_bwt_info_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
raw = np.frombuffer(b"ACGT", dtype=np.uint8)[synth(n, 5, 3) - 1].tobytes()
for it in range(2):
    f = libc.fopen(b"/dev/null", b"wb")
    t0 = time.perf_counter()
    rc = lib.stralg_amd_write_complete_bwt_info_stream(f, raw, False)
    dt = time.perf_counter() - t0
    libc.fclose(f)
    assert rc == 0
size = 4 + n + 4 * (n + 1) + 388 + 20 + 20 * (n + 2) + 1
print(f"2^{log2n} DNA: index of {size/2**30:.2f} GiB built and streamed to /dev/null in {dt*1e3:.0f} ms = {size/dt/1e9:.1f} GB/s")
