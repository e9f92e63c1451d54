"""The kernel sources on the CPU harness under AddressSanitizer (the only sanitizer run there is: the GPU pool has
none).  __shared__ arrays are plain statics in the harness, so an index past an LDS array, a global buffer or a
stack array in a kernel is reported; this is the check that would have caught the undersized LDS rows of the wide
O-table kernel.  tests/test_emu_pipeline.py is run once more in a child process (the sanitizer runtime has to be
preloaded) against the sanitized build."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def test_kernels_under_address_sanitizer():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "stralg_amd", "csrc"), "emu-asan"])
    env = dict(os.environ, LD_PRELOAD=asan, STRALG_EMU_ASAN="1",
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0")
    # every kernel-level test of the harness once more, with the sanitizer watching
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_emu_pipeline.py"), "-x", "-q",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    assert out.returncode == 0 and "AddressSanitizer" not in out.stdout + out.stderr, (out.stdout[-3000:], out.stderr[-3000:])
