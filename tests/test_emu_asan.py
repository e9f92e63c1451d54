"""The kernel sources on the CPU harness under AddressSanitizer (the only sanitizer run there is: the GPU pool has
none).  __shared__ arrays are plain statics in the harness, so an index past an LDS array, a global buffer or a
stack array in a kernel is reported; this is the check that would have caught the undersized LDS rows of the wide
O-table kernel.  tests/test_emu_pipeline.py is run once more in a child process (the sanitizer runtime has to be
preloaded) against the sanitized build; tests/conftest.py starts that child at the beginning of the session, beside
the other tests, and this test collects it."""
import pytest

import conftest


def test_kernels_under_address_sanitizer():
    rc, out = conftest.wait_asan_child()
    if rc is None:
        pytest.skip(out)
    assert rc == 0 and "AddressSanitizer" not in out, out[-6000:]
