"""The kernel sources on the CPU harness under AddressSanitizer (the only sanitizer run there is: the GPU pool has
none).  __shared__ arrays are plain statics in the harness, so an index past an LDS array, a global buffer or a
stack array in a kernel is reported; this is the check that would have caught the undersized LDS rows of the wide
O-table kernel.  The work is done in a child process because the sanitizer runtime has to be preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
from stralg_amd.api import Context
import oracle
from stralg_amd.synth import synth
ctx = Context(0, lib_path=ROOT + "/tests/emu/libstralg_amd_emu_asan.so")
for sigma, n in ((5, 5000), (5, 2049), (3, 700), (9, 2500), (21, 3000), (32, 1500), (33, 600), (64, 1200), (65, 500),
                 (128, 600), (256, 3000)):
    x = synth(n, sigma, 3)
    if n > 1100:
        x[100:140] = x[1000:1040]
    want = oracle.sa_is(x, sigma)
    for no_direct, general in ((False, False), (True, False), (False, True)):
        ctx.set_no_direct_sort(no_direct); ctx.force_general_path(general)
        sa, bw = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint8)
        ctx.sa_bwt_build_dev(x, n, sigma, sa, bw)
        assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), (sigma, n)
    ctx.set_no_direct_sort(False); ctx.force_general_path(False)
    if sigma <= 128:
        c, o = ctx.bwt_tables(x, want, sigma)
        assert (o.ravel() == oracle.o_table(x, want, sigma).ravel()).all(), (sigma, n)
        inv, lcp = ctx.inverse_lcp(x, want)
        assert (lcp == oracle.lcp(x, want)).all()
assert ctx.fasta_records(b">a b\nACGT\nAC\n>c\nTT\n") == [(b"ab", b"ACGTAC"), (b"c", b"TT")]
out = np.zeros(9, np.uint8)
assert ctx.remap_dev(np.frombuffer(b"GATTACA!", dtype=np.uint8).copy(), 8, out)[0] == 6
print("asan run clean")
"""


def test_kernels_under_address_sanitizer():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "stralg_amd", "csrc"), "emu-asan"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0")
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=env, capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0 and "asan run clean" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
