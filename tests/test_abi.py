"""The C-ABI library loads and exports every symbol include/*.h declares.  No
compute calls (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef[^;]*;", "", text)  # function-pointer types are not symbols
    return sorted(set(re.findall(r"\b(\w+)\s*\([^;{]*\)\s*;", text)))


@pytest.fixture(scope="module")
def product_lib():
    path = os.path.join(ROOT, "stralg_amd", "libstralg_amd.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "stralg_amd", "csrc")])
    return C.CDLL(path)


def test_exports_shim_symbols(product_lib):
    names = _declared("stralg_amd.h")
    assert "sx_sa_build" in names and "sx_bwt_tables" in names
    for n in names:
        assert hasattr(product_lib, n), n


def test_exports_reference_named_symbols(product_lib):
    names = _declared("stralg_compat.h")
    for must in ("sa_is_construction", "sa_is_mem_construction", "skew_sa_construction", "init_bwt_table",
                 "alloc_bwt_table", "build_complete_table", "allocate_sa_", "free_suffix_array", "remap"):
        assert must in names
    for n in names:
        assert hasattr(product_lib, n), n


def test_python_binding_matches_header(product_lib):
    from stralg_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared("stralg_amd.h")
    lib = _lib.load()
    assert lib.sx_kernel_class_name(0) == b"classify"
    assert [lib.sx_kernel_class_name(i).decode() for i in range(len(_lib.KC_NAMES))] == _lib.KC_NAMES


def test_struct_layouts_match_reference():
    # 40 / 388 / 56 bytes on LP64; checked by compiling against the header
    src = r'''
    #include "stralg_compat.h"
    #include <stdio.h>
    #include <stddef.h>
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu\n", sizeof(struct suffix_array), sizeof(struct remap_table),
               sizeof(struct bwt_table), offsetof(struct suffix_array, array),
               offsetof(struct remap_table, rev_table), offsetof(struct bwt_table, ro_indices));
        return 0;
    }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o",
                               os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).split()
    assert [int(x) for x in out] == [40, 388, 56, 16, 260, 48]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from stralg_amd.api import Context, StralgAmdError
    with pytest.raises(StralgAmdError):
        Context(0)


def test_c_harness_links_against_the_library(product_lib, tmp_path):
    """tools/sa_construction_harness.c (the restated performance harness) compiles as plain C against
    include/stralg_compat.h and links with nothing but libstralg_amd.so; the gpu suite runs it."""
    exe = tmp_path / "harness"
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-D_GNU_SOURCE", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "sa_construction_harness.c"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "stralg_amd"), "-lstralg_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "stralg_amd")])
    assert exe.exists()


# ---- the boundary against a real libstralg (needs the reference checkout: the build container) -----------------

REF = "/root/reference"

CALLER = r'''
/* a caller written against the reference's own, unmodified headers: functions that stay in libstralg next to
 * functions that now come from libstralg_amd.so (host-side ones only: there is no GPU where this runs) */
#include <stralg.h>
#include <suffix_array_internal.h>
#include <fasta.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : ".";
    char path[512];
    uint8_t text[] = "mississippi";
    struct remap_table *table = alloc_remap_table(text);         /* libstralg_amd.so */
    uint8_t remapped[16], back[16];
    remap(remapped, text, table);                                /* libstralg_amd.so */
    rev_remap(back, remapped, table);                            /* libstralg (remap.c:133-141) */
    if (strcmp((char *)back, (char *)text) != 0) return 2;
    if (table->alphabet_size != 5) return 3;
    snprintf(path, sizeof path, "%s/table.bin", dir);
    write_remap_table_fname(path, table);                        /* libstralg_amd.so */
    struct remap_table *again = read_remap_table_fname(path);    /* libstralg_amd.so */
    if (!identical_remap_tables(table, again)) return 4;         /* libstralg (remap.c:224-237) */
    snprintf(path, sizeof path, "%s/string.bin", dir);
    write_string_fname(path, text);                              /* libstralg_amd.so */
    uint8_t *s = read_string_fname(path);                        /* libstralg_amd.so */
    uint8_t *r = str_rev(s);                                     /* libstralg (string_utils.c:40-43) */
    if (strcmp((char *)r, "ippississim") != 0) return 5;
    /* a hand-made suffix array through the serialisation pair (no construction: that needs the GPU) */
    struct suffix_array *sa = allocate_sa_(remapped);            /* libstralg_amd.so */
    const uint32_t want[12] = {11, 10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2};
    memcpy(sa->array, want, sizeof want);
    snprintf(path, sizeof path, "%s/sa.bin", dir);
    write_suffix_array_fname(path, sa);                          /* libstralg_amd.so */
    struct suffix_array *sa2 = read_suffix_array_fname(path, remapped); /* libstralg_amd.so */
    if (!identical_suffix_arrays(sa, sa2)) return 6;             /* libstralg (suffix_array.c) */
    if (lower_bound_search(sa, (uint8_t *)"\2") != 5) return 7;  /* libstralg (suffix_array.c:90-112): first suffix >= "m" */
    printf("ok %u %u\n", table->alphabet_size, sa2->length);
    free_suffix_array(sa);                                       /* libstralg_amd.so */
    free_suffix_array(sa2);
    free_remap_table(table);
    free_remap_table(again);
    free(s);
    free(r);
    /* never called, only bound: the construction entry points must come from the GPU library */
    if (argc > 5) {
        (void)sa_is_construction(remapped, 5);
        (void)skew_sa_construction(text);
        (void)build_complete_table(text, true);
        (void)load_fasta_records("x", NULL);
    }
    return 0;
}
'''


def _defined(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return {l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in "TWDBR"}


def _undefined(lib):
    out = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1].split("@")[0] for l in out.splitlines() if l.split()}


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "stralg")), reason="needs the reference checkout")
def test_links_next_to_a_real_libstralg(product_lib, tmp_path):
    """INTEGRATION.md section 2, executed: the reference with the guards of tools/stralg_guard.py, compiled
    with -DSTRALG_WITH_MI355X and linked against libstralg_amd.so.  Every reference-named symbol is defined
    exactly once across the two libraries, libstralg's own references to the moved functions resolve from the
    GPU library, and a caller built against the reference's unmodified headers binds each call to the
    intended library (LD_DEBUG=bindings)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stralg_guard
    amd_dir = os.path.join(ROOT, "stralg_amd")
    amd = os.path.join(amd_dir, "libstralg_amd.so")
    report = stralg_guard.guard_tree(REF, str(tmp_path / "src"))
    guarded_names = {name for _, _, _, name in report}
    ours = {s for s in _defined(amd) if not s.startswith(("sx_", "stralg_amd_"))}
    assert ours == guarded_names, (sorted(ours - guarded_names), sorted(guarded_names - ours))
    # the library exports nothing but its C ABI
    assert not [s for s in _defined(amd) if s.startswith("_Z")]

    inc = ["-I", str(tmp_path / "src" / "stralg"), "-I", str(tmp_path / "src" / "bioinf")]
    link = ["-L", amd_dir, "-lstralg_amd", "-Wl,-rpath," + amd_dir]
    stralg_so, bioinf_so = str(tmp_path / "libstralg.so"), str(tmp_path / "libstralg_bioinf.so")
    csrc = sorted(str(p) for p in (tmp_path / "src" / "stralg").glob("*.c"))
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-D_GNU_SOURCE", "-w", "-DSTRALG_WITH_MI355X", "-shared", "-fPIC"] +
                          inc + csrc + ["-o", stralg_so] + link)
    bsrc = sorted(str(p) for p in (tmp_path / "src" / "bioinf").glob("*.c"))
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-D_GNU_SOURCE", "-w", "-DSTRALG_WITH_MI355X", "-shared", "-fPIC"] +
                          inc + bsrc + ["-o", bioinf_so, "-L", str(tmp_path), "-lstralg", "-Wl,-rpath," + str(tmp_path)] + link)
    # exactly once: nothing the GPU library defines is still defined by the guarded reference libraries
    theirs = _defined(stralg_so) | _defined(bioinf_so)
    assert not (theirs & ours), sorted(theirs & ours)
    # ... and what they no longer define but still use resolves from the GPU library
    needed = (_undefined(stralg_so) | _undefined(bioinf_so)) & guarded_names
    assert needed and needed <= ours
    assert "allocate_sa_" in needed  # suffix_array.c's qsort construction still allocates through it
    # without the macro the same sources still build the complete reference library (the guards change nothing else)
    whole = str(tmp_path / "libstralg_whole.so")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-D_GNU_SOURCE", "-w", "-shared", "-fPIC"] + inc + csrc + ["-o", whole])
    assert guarded_names - {n for f, _, _, n in report if f.startswith("bioinf/")} <= _defined(whole)

    (tmp_path / "caller.c").write_text(CALLER)
    exe = str(tmp_path / "caller")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-D_GNU_SOURCE", "-I", os.path.join(REF, "stralg"), "-I",
                           os.path.join(REF, "bioinf"), str(tmp_path / "caller.c"), "-o", exe, "-L", str(tmp_path),
                           "-lstralg_bioinf", "-lstralg", "-Wl,-rpath," + str(tmp_path)] + link)
    env = dict(os.environ, LD_BIND_NOW="1", LD_DEBUG="bindings", LD_DEBUG_OUTPUT=str(tmp_path / "ld"))
    run = subprocess.run([exe, str(tmp_path)], env=env, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("ok 5 12"), (run.returncode, run.stdout, run.stderr[-500:])
    log = "".join(p.read_text(errors="replace") for p in tmp_path.glob("ld.*"))
    bound = {}
    for m in re.finditer(r"binding file (\S+) \[\d+\] to (\S+) \[\d+\]: normal symbol `(\w+)'", log):
        if os.path.basename(m.group(1)) == "caller":
            bound[m.group(3)] = os.path.basename(m.group(2))
    for name in ("alloc_remap_table", "remap", "write_remap_table_fname", "read_remap_table_fname", "write_string_fname",
                 "read_string_fname", "allocate_sa_", "write_suffix_array_fname", "read_suffix_array_fname",
                 "free_suffix_array", "free_remap_table", "sa_is_construction", "skew_sa_construction",
                 "build_complete_table", "load_fasta_records"):
        assert bound.get(name) == "libstralg_amd.so", (name, bound.get(name))
    for name in ("rev_remap", "identical_remap_tables", "str_rev", "identical_suffix_arrays", "lower_bound_search"):
        assert bound.get(name) == "libstralg.so", (name, bound.get(name))
    # libstralg's own calls into the moved functions bind to the GPU library as well
    inner = {m.group(3): os.path.basename(m.group(2))
             for m in re.finditer(r"binding file (\S+) \[\d+\] to (\S+) \[\d+\]: normal symbol `(\w+)'", log)
             if os.path.basename(m.group(1)) == "libstralg.so" and m.group(3) in guarded_names}
    assert inner and set(inner.values()) == {"libstralg_amd.so"}, inner


def test_long_strlen_never_reads_behind_the_terminator(product_lib):
    """build_complete_table's strlen of a long record: one walk with aligned 32-byte loads, never a byte behind the
    terminator's own block.  A record that ends exactly at the end of its mapping, with an unreadable page behind (mmap +
    mprotect), strings with the terminator in every position class and every alignment, the letters that come back; and
     a second thread that keeps taking the memory right behind the string away while the scan runs."""
    import ctypes as C
    import mmap
    import threading
    import numpy as np
    lib = product_lib
    lib.stralg_amd_strlen_and_letters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    lib.stralg_amd_strlen_and_letters.restype = C.c_size_t
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE
    chunk = 4 << 20
    rng = np.random.default_rng(2)
    for n in (0, 5, chunk - 1, chunk, chunk + 1, chunk + 31, chunk + 32, chunk + 33, 3 * chunk + 12345, 9 * chunk):
        total = ((n + 1 + page - 1) // page + 1) * page  # the string, then one guard page
        m = mmap.mmap(-1, total, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)  # (what malloc hands out for a long record)
        buf = (C.c_uint8 * total).from_buffer(m)
        base = C.addressof(buf)
        start = total - page - (n + 1)  # the terminator is the last byte in front of the guard page
        arr = np.frombuffer(m, dtype=np.uint8)
        letters = rng.choice(np.array([65, 67, 71, 84, 78, 200], dtype=np.uint8), size=4, replace=False)
        arr[start:start + n] = rng.choice(letters, size=n) if n else []
        arr[start + n] = 0
        assert libc.mprotect(base + total - page, page, 0) == 0  # PROT_NONE: a read there is a fault
        present = (C.c_uint8 * 256)()
        have = C.c_int(-1)
        got = lib.stralg_amd_strlen_and_letters(base + start, present, C.byref(have))
        assert got == n, (n, got)
        if n >= chunk:
            assert have.value == 1
            assert sorted(c for c in range(256) if present[c]) == sorted(set(arr[start:start + n].tolist())), n
        else:
            assert have.value == 0
        libc.mprotect(base + total - page, page, 3)
        del arr, buf
        m.close()
    # more than 16 distinct letters (the block compare gives way to the byte walk), terminator at an odd offset from an odd start
    n = chunk + 1000
    raw = np.zeros(n + 64, dtype=np.uint8)
    raw[7:7 + n] = rng.integers(1, 256, size=n, dtype=np.uint8)
    raw[7 + n] = 0
    present = (C.c_uint8 * 256)()
    have = C.c_int(-1)
    assert lib.stralg_amd_strlen_and_letters(raw.ctypes.data + 7, present, C.byref(have)) == n and have.value == 1
    assert sorted(c for c in range(256) if present[c]) == sorted(set(raw[7:7 + n].tolist()))
    # a record inside a FILE mapping: pages of such a mapping behind the file's end are listed as readable and are a
    # SIGBUS to touch; the walk stops at the terminator like strlen (Python's default mmap(-1, ...) is a shared mapping)
    n = 2 * chunk + 77
    m = mmap.mmap(-1, n + 1)
    arr = np.frombuffer(m, dtype=np.uint8)
    arr[:n] = 65
    arr[n] = 0
    buf = (C.c_uint8 * (n + 1)).from_buffer(m)
    have = C.c_int(-1)
    assert lib.stralg_amd_strlen_and_letters(C.addressof(buf), None, C.byref(have)) == n
    del arr, buf
    m.close()
    # the memory right behind the string is somebody else's: a second thread flips it between PROT_NONE and readable while
    # 16 "host threads" are configured (round 4's read-ahead scan took a SIGSEGV here)
    n = 64 << 20
    behind = 8 << 20
    m = mmap.mmap(-1, n + page + behind, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    buf = (C.c_uint8 * (n + page + behind)).from_buffer(m)
    base = C.addressof(buf)
    arr = np.frombuffer(m, dtype=np.uint8)
    arr[:] = 67
    start = page - 1  # the string ends with the last byte of its pages: [start, n + page)
    arr[n + page - 1] = 0
    stop = threading.Event()

    def flip():
        while not stop.is_set():
            libc.mprotect(base + n + page, behind, 0)
            libc.mprotect(base + n + page, behind, 3)

    th = threading.Thread(target=flip)
    old = os.environ.get("STRALG_AMD_HOST_THREADS")
    os.environ["STRALG_AMD_HOST_THREADS"] = "16"
    th.start()
    try:
        for _ in range(20):
            assert lib.stralg_amd_strlen_and_letters(base + start, None, C.byref(have)) == n + page - 1 - start
    finally:
        stop.set()
        th.join()
        if old is None:
            del os.environ["STRALG_AMD_HOST_THREADS"]
        else:
            os.environ["STRALG_AMD_HOST_THREADS"] = old
    libc.mprotect(base + n + page, behind, 3)
    del arr, buf
    m.close()
