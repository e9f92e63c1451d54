import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libstralg_amd_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


# ---- the AddressSanitizer run of the kernel harness (tests/test_emu_asan.py) -----------------------------------------
# It repeats tests/test_emu_pipeline.py in a child process under the sanitizer, which takes four minutes: the child is
# started as soon as the collection shows that the test is part of this run, works beside the other CPU tests on
# another core, and the test itself only waits for it and reads its output.
_asan = {"proc": None, "log": None, "skip": None}


def start_asan_child():
    if _asan["proc"] is not None or _asan["skip"] is not None:
        return
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        _asan["skip"] = "no libasan in this toolchain"
        return
    import tempfile
    log = tempfile.NamedTemporaryFile(prefix="stralg_asan_", suffix=".log", delete=False)
    env = dict(os.environ, STRALG_EMU_ASAN="1", STRALG_ASAN_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0")
    # build first (no preload: the compiler is not to be sanitized), then every kernel-level test of the harness once
    # more with the sanitizer watching
    cmd = ("make -s -C '%s' emu-asan && LD_PRELOAD=\"$STRALG_ASAN_PRELOAD\" '%s' -m pytest '%s' -x -q -p no:cacheprovider"
           % (os.path.join(ROOT, "stralg_amd", "csrc"), sys.executable, os.path.join(ROOT, "tests", "test_emu_pipeline.py")))
    _asan["log"] = log.name
    _asan["proc"] = subprocess.Popen(["bash", "-c", cmd], env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)


def wait_asan_child(timeout=2400):
    """(return code, output) of the sanitizer run, or (None, reason) when it cannot run here"""
    start_asan_child()
    if _asan["skip"] is not None:
        return None, _asan["skip"]
    rc = _asan["proc"].wait(timeout=timeout)
    with open(_asan["log"]) as f:
        out = f.read()
    os.unlink(_asan["log"])
    _asan["proc"] = None
    return rc, out


def pytest_collection_modifyitems(config, items):
    # the test that collects the sanitizer child goes last: the child works while the others run
    items.sort(key=lambda it: "test_kernels_under_address_sanitizer" in it.nodeid)


def pytest_collection_finish(session):
    if any("test_kernels_under_address_sanitizer" in it.nodeid for it in session.items):
        start_asan_child()


def pytest_sessionfinish(session, exitstatus):
    if _asan["proc"] is not None and _asan["proc"].poll() is None:  # (the run stopped early: -x)
        _asan["proc"].kill()


def golden_cases():
    """name -> dict(sym, sigma, sa[, raw, c, o, ro]) from tests/golden/golden.npz."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
    cases = {}
    for key in z.files:
        name, field = key.rsplit("/", 1)
        cases.setdefault(name, {})[field] = z[key]
    for c in cases.values():
        c["sigma"] = int(c["sigma"][0])
    return cases


@pytest.fixture(scope="session")
def golden():
    return golden_cases()


def fasta_cases():
    """name -> dict(file bytes, err, packed image bytes, records) from tests/golden/golden_fasta.npz."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_fasta.npz"))
    cases = {}
    for key in z.files:
        if key.startswith("serial/"):
            continue
        name, field = key.rsplit("/", 1)
        cases.setdefault(name, {})[field] = z[key]
    return {k: dict(file=v["file"].tobytes(), err=int(v["err"][0]), packed=v["packed"].tobytes(), records=int(v["records"][0]))
            for k, v in cases.items()}


def serial_cases():
    """name -> dict(raw, with_reverse, forward_only): the reference's write_complete_bwt_info byte streams"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_fasta.npz"))
    cases = {}
    for key in z.files:
        if key.startswith("serial/"):
            _, name, field = key.split("/")
            cases.setdefault(name, {})[field] = z[key].tobytes()
    return cases


def check_serialisation(lib, cases, tmp_path):
    """the library's build_complete_table + write_complete_bwt_info, its streaming writer, and read-back"""
    import ctypes as C
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    lib.build_complete_table.argtypes = [C.c_char_p, C.c_bool]
    lib.build_complete_table.restype = C.c_void_p
    lib.write_complete_bwt_info_fname.argtypes = [C.c_char_p, C.c_void_p]
    lib.write_complete_bwt_info_fname.restype = None
    lib.read_complete_bwt_info_fname.argtypes = [C.c_char_p]
    lib.read_complete_bwt_info_fname.restype = C.c_void_p
    lib.completely_free_bwt_table.argtypes = [C.c_void_p]
    lib.completely_free_bwt_table.restype = None
    lib.stralg_amd_write_complete_bwt_info_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    for name, c in cases.items():
        for rev, want in ((True, c["with_reverse"]), (False, c["forward_only"])):
            a, b, d = (str(tmp_path / f"{name}-{int(rev)}-{k}").encode() for k in "abd")
            t = lib.build_complete_table(c["raw"], rev)
            lib.write_complete_bwt_info_fname(a, t)
            lib.completely_free_bwt_table(t)
            assert open(a, "rb").read() == want, (name, rev, "write_complete_bwt_info")
            f = libc.fopen(b, b"wb")
            assert lib.stralg_amd_write_complete_bwt_info_stream(f, c["raw"], rev) == 0
            libc.fclose(f)
            assert open(b, "rb").read() == want, (name, rev, "streaming writer")
            t = lib.read_complete_bwt_info_fname(a)  # and back: read, write again
            lib.write_complete_bwt_info_fname(d, t)
            lib.completely_free_bwt_table(t)
            assert open(d, "rb").read() == want, (name, rev, "read + write")


def genome_cases():
    """tests/golden/golden_genomes.npz: the production caller's genomes (tools/readmappers/data/genomes/hg38-*.fa).
    file name -> dict(file, err, packed, records, recs=[dict(name, sym, sigma, c, sa_sha256, sa_rows, ...)])"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_genomes.npz"))
    cases = {}
    for key in z.files:
        parts = key.split("/")
        c = cases.setdefault(parts[0], {"recs": {}})
        if len(parts) == 2:
            c[parts[1]] = z[key]
        else:
            c["recs"].setdefault(parts[1], {})[parts[2]] = z[key]
    out = {}
    for fname, c in cases.items():
        recs = []
        for k in sorted(c["recs"], key=lambda r: int(r[3:])):
            r = dict(c["recs"][k])
            r["name"] = r["name"].tobytes()
            r["sigma"] = int(r["sigma"][0])
            for f in list(r):
                if f.endswith("_sha256"):
                    r[f] = r[f].tobytes().hex()
                elif f.endswith("_len"):
                    r[f] = int(r[f][0])
            recs.append(r)
        out[fname] = dict(file=c["file"].tobytes(), err=int(c["err"][0]), packed=c["packed"].tobytes(),
                          records=int(c["records"][0]), recs=recs)
    return out


def check_against_sha(arr, rec, field, what):
    """an array against its fixture: SHA-256 of the little-endian uint32 bytes, and every 997th row for a useful message"""
    import hashlib
    a = np.ascontiguousarray(arr, dtype=np.uint32)
    assert (a[::997] == rec[field + "_rows"]).all(), (what, field, "sampled rows differ")
    assert hashlib.sha256(a.tobytes()).hexdigest() == rec[field + "_sha256"], (what, field)


def check_genomes(lib, cases, tmp_path, names=None):
    """load_fasta_records -> stralg_amd_fasta_tables_batch (include_reverse, two lanes on device 0) ->
    write_complete_bwt_info on the genome files, everything against the reference's results"""
    import ctypes as C
    import hashlib

    class SA(C.Structure):
        _fields_ = [("string", C.POINTER(C.c_uint8)), ("length", C.c_uint32), ("array", C.POINTER(C.c_uint32)),
                    ("inverse", C.c_void_p), ("lcp", C.c_void_p)]

    class BT(C.Structure):
        _fields_ = [("remap_table", C.c_void_p), ("sa", C.POINTER(SA)), ("c_table", C.POINTER(C.c_uint32)),
                    ("o_table", C.POINTER(C.c_uint32)), ("o_indices", C.c_void_p),
                    ("ro_table", C.POINTER(C.c_uint32)), ("ro_indices", C.c_void_p)]

    lib.load_fasta_records.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.load_fasta_records.restype = C.c_void_p
    lib.free_fasta_records.argtypes = [C.c_void_p]
    lib.number_of_fasta_records.argtypes = [C.c_void_p]
    lib.number_of_fasta_records.restype = C.c_uint32
    lib.stralg_amd_fasta_tables_batch.argtypes = [C.c_void_p, C.c_bool, C.POINTER(C.c_int), C.c_int, C.POINTER(C.POINTER(BT))]
    lib.stralg_amd_fasta_tables_batch.restype = C.c_int
    lib.write_complete_bwt_info_fname.argtypes = [C.c_char_p, C.POINTER(BT)]
    lib.write_complete_bwt_info_fname.restype = None
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    for fname, g in cases.items():
        if names is not None and fname not in names:
            continue
        path = tmp_path / fname
        path.write_bytes(g["file"])
        err = C.c_int(-1)
        h = lib.load_fasta_records(str(path).encode(), C.byref(err))
        assert h and err.value == 0 and lib.number_of_fasta_records(h) == g["records"]
        out = (C.POINTER(BT) * g["records"])()
        devs = (C.c_int * 2)(0, 0)
        assert lib.stralg_amd_fasta_tables_batch(h, True, devs, 2, out) == g["records"]
        for t, want in zip(out, g["recs"]):
            N, sigma = t.contents.sa.contents.length, want["sigma"]
            assert N == want["sym"].size + 1
            assert (np.ctypeslib.as_array(t.contents.sa.contents.string, shape=(N - 1,)) == want["sym"]).all()
            check_against_sha(np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(N,)), want, "sa", fname)
            assert (np.ctypeslib.as_array(t.contents.c_table, shape=(sigma,)) == want["c"]).all()
            check_against_sha(np.ctypeslib.as_array(t.contents.o_table, shape=(N + 1, sigma)), want, "o", fname)
            check_against_sha(np.ctypeslib.as_array(t.contents.ro_table, shape=(N + 1, sigma)), want, "ro", fname)
            idx = tmp_path / (fname + ".bwt")
            lib.write_complete_bwt_info_fname(str(idx).encode(), t)
            blob = idx.read_bytes()
            assert len(blob) == want["serial_with_reverse_len"]
            assert hashlib.sha256(blob).hexdigest() == want["serial_with_reverse_sha256"], fname
            lib.completely_free_bwt_table(t)
        lib.free_fasta_records(h)


@pytest.fixture(scope="session")
def golden_genomes():
    return genome_cases()


@pytest.fixture(scope="session")
def golden_fasta():
    return fasta_cases()


def check_fasta(records_of, cases):
    """records_of(file bytes) -> [(name, seq)] or raises an error carrying SX_E_MALFORMED (-4)"""
    from oracle import pyoracle
    for name, c in cases.items():
        want = pyoracle.fasta_records_of(c["packed"], c["records"])
        try:
            got = records_of(c["file"])
            assert c["err"] == 0, (name, "should be malformed")
            assert got == want, name
        except Exception as e:  # noqa: BLE001 -- the binding's own error type
            if isinstance(e, AssertionError):
                raise
            assert c["err"] == 2 and "-4" in str(e), (name, e)


def fasta_soup_cases(rng, cases, sizes):
    """files of '>' / newline / white space / letters / bytes above 0x7f in any order, of well-formed lines with stray '>', of
    random bytes, and with a NUL inside (the image ends there: io.c:15-18); most end in a sequence line"""
    soup = np.array(list(b">\n \t\r\x0b\x0cACGTNacgt>>\n\n\n") + [0x80, 0xFF, 0x0E, 0x08, 0x1F, 0x21, 0x3D, 0x3F], dtype=np.uint8)
    for k in range(cases):
        n = int(rng.choice(sizes)) + int(rng.integers(0, 3))
        kind = int(rng.integers(0, 4))
        if kind == 1:
            data = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n)
            data[59::61] = 10
            if n:
                data[rng.integers(0, n, size=max(1, n // 500))] = ord(">")
        elif kind == 2:
            data = rng.integers(1, 256, size=n, dtype=np.uint8)
        else:
            data = rng.choice(soup, size=n)
            if kind == 3 and n:
                data[int(rng.integers(0, n))] = 0
        if kind != 3 and n > 8 and rng.integers(0, 4):
            data[-6:] = np.frombuffer(b"\nACGT\n", np.uint8)
        if n and rng.integers(0, 2):
            data[0] = ord(">")
        yield k, kind, np.ascontiguousarray(data, np.uint8)


def check_fasta_soup(ctx, rng, cases, sizes):
    from oracle import pyoracle
    for k, kind, data in fasta_soup_cases(rng, cases, sizes):
        w_bad, _, w_recs = pyoracle.fasta_pack(data)
        try:
            got = ctx.fasta_records(data.tobytes())
        except Exception as e:  # noqa: BLE001 -- the binding's own error type
            assert w_bad and "-4" in str(e), (k, data.size, kind, e)
            continue
        assert not w_bad, (k, data.size, kind, "should be malformed")
        assert got == w_recs, (k, data.size, kind)


@pytest.fixture(scope="session")
def emu_ctx():
    """Context on the CPU execution harness build of the kernel sources (tests/emu).
    Test infrastructure: it exercises kernel logic without a GPU and is never what
    the product loads."""
    # STRALG_EMU_ASAN=1 (set by tests/test_emu_asan.py for its child run): the AddressSanitizer build of the harness
    asan = os.environ.get("STRALG_EMU_ASAN") == "1"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "stralg_amd", "csrc"), "emu-asan" if asan else "emu"])
    from stralg_amd.api import Context
    ctx = Context(0, lib_path=EMU_LIB.replace("_emu.so", "_emu_asan.so") if asan else EMU_LIB)
    # (short records of few symbols are sorted directly by default -- SX_FLAG_SMALL_DIRECT_MAX --: the tests are short records
    #  and mean the SA-IS kernels unless they say otherwise; test_short_records_direct_sort switches it back on)
    ctx.set_small_direct_max(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from stralg_amd.api import Context
    ctx = Context(0)
    ctx.set_small_direct_max(0)  # (as in emu_ctx: the tests mean the SA-IS kernels; test_short_records_direct_sort switches it back on)
    yield ctx
    ctx.close()


# the session contexts above switch the direct sort of short records off, so the suites that run through
# them exercise the SA-IS kernels and not what a caller gets by default.  Tests that take one of these fixtures run twice:
# once as before, once with the library's default routing (short records of few symbols sorted directly).
def _routed(ctx, request):
    ctx.set_small_direct_max(0 if request.param == "sa_is_kernels" else -1)
    try:
        yield ctx
    finally:
        ctx.set_small_direct_max(0)


@pytest.fixture(params=["sa_is_kernels", "default_routing"])
def gpu_routed(gpu_ctx, request):
    yield from _routed(gpu_ctx, request)


@pytest.fixture(params=["sa_is_kernels", "default_routing"])
def emu_routed(emu_ctx, request):
    yield from _routed(emu_ctx, request)
