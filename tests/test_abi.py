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
