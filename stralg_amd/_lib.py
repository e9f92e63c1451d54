"""ctypes binding of libstralg_amd.so (the C-ABI declared in include/stralg_amd.h).

The product library is stralg_amd/libstralg_amd.so, built in-tree by
stralg_amd/csrc/Makefile for gfx950.  There is no CPU fallback: if the library
is missing, or no GPU is visible when a context is created, the call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "libstralg_amd.so")

KC_NAMES = ["classify", "samples", "keys", "radix_hist", "radix_scatter", "scan", "names", "doubling",
            "induce_gather", "induce_scan", "induce_scatter", "induce_chain", "bwt_gather", "otable", "misc",
            "fasta", "remap", "lcp", "search", "local_sort"]


class KernelStat(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("ms", C.c_double), ("alg_bytes", C.c_uint64)]


class BuildStats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_lms", C.c_uint64), ("n_samples", C.c_uint64),
                ("n_names", C.c_uint64), ("key_bits", C.c_uint32), ("key_slots", C.c_uint32),
                ("doubling_rounds", C.c_uint32), ("induce_rounds", C.c_uint32),
                ("sort_passes", C.c_uint32), ("lms_path", C.c_uint32), ("sort_local", C.c_uint32), ("refine_tiers", C.c_uint32),
                ("ms_total", C.c_double), ("induce_redo", C.c_uint32), ("long_runs", C.c_uint32),
                ("recursion_levels", C.c_uint32), ("sample_tied_permille", C.c_uint32), ("long_subbuckets", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def load(path=None):
    """Load the shared library and declare every entry point of include/stralg_amd.h."""
    path = path or PRODUCT_LIB
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever
    # is loaded first serves the whole process, and a second copy cannot open the GPU,
    # so torch -- the process's owner of device memory and streams -- must come first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `make -C stralg_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, u8p, u32p, u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    sig = {
        "sx_device_count": (C.c_int, []),
        "sx_device_numa_node": (C.c_int, [C.c_int]),
        "sx_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "sx_ctx_destroy": (None, [vp]),
        "sx_ctx_live_count": (C.c_int, []),
        "sx_last_error": (C.c_char_p, [vp]),
        "sx_ctx_trim": (None, [vp]),
        "sx_ctx_set_flag": (C.c_int, [vp, C.c_int, C.c_int]),
        "sx_sa_build": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint32, u32p]),
        "sx_sa_build_dev": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint32, u32p]),
        "sx_sa_bwt_build_dev": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint32, u32p, u8p]),
        "sx_bwt_tables_from_bwt_dev": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint32, u32p, u32p]),
        "sx_build_tables": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint32, u32p, u32p, u32p]),
        "sx_bwt_tables": (C.c_int, [vp, u8p, u32p, C.c_uint64, C.c_uint32, u32p, u32p]),
        "sx_bwt_tables_dev": (C.c_int, [vp, u8p, u32p, C.c_uint64, C.c_uint32, u32p, u32p, u8p]),
        "sx_sa_inverse_dev": (C.c_int, [vp, u32p, C.c_uint64, u32p]),
        "sx_sa_lcp_dev": (C.c_int, [vp, u8p, u32p, C.c_uint64, u32p, u32p]),
        "sx_sa_inverse_lcp": (C.c_int, [vp, u8p, u32p, C.c_uint64, u32p, u32p]),
        "sx_bwt_exact_search_dev": (C.c_int, [vp, u32p, u32p, C.c_uint64, C.c_uint32, u8p, u32p, C.c_uint32, u32p, u32p]),
        "sx_build_tables_stream": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
        "sx_fasta_pack_dev": (C.c_int, [vp, u8p, C.c_uint64, u8p, C.POINTER(C.c_uint64), u32p, C.c_uint64,
                                        C.POINTER(C.c_uint32)]),
        "sx_fasta_pack": (C.c_int, [vp, u8p, C.c_uint64, u8p, C.POINTER(C.c_uint64), u32p, C.c_uint64,
                                    C.POINTER(C.c_uint32)]),
        "sx_remap_dev": (C.c_int, [vp, u8p, C.c_uint64, u8p, C.POINTER(C.c_int16), C.POINTER(C.c_uint32)]),
        "sx_reverse_dev": (C.c_int, [vp, u8p, C.c_uint64, u8p]),
        "sx_profile_enable": (C.c_int, [vp, C.c_int]),
        "sx_profile_only": (C.c_int, [vp, C.c_int]),
        "sx_profile_reset": (C.c_int, [vp]),
        "sx_profile_read": (C.c_int, [vp, C.POINTER(KernelStat)]),
        "sx_kernel_class_name": (C.c_char_p, [C.c_int]),
        "sx_last_stats": (C.c_int, [vp, C.POINTER(BuildStats)]),
        "sx_synth_dev": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint32, C.c_uint64]),
        "sx_membw_probe": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
        "sx_prim_sort_pairs_dev": (C.c_int, [vp, u64p, u32p, u64p, u32p, C.c_uint64, C.c_int, C.c_int,
                                             C.POINTER(C.c_int)]),
        "sx_prim_exclusive_sum_dev": (C.c_int, [vp, u32p, u32p, C.c_uint64, u32p]),
        "sx_prim_classify_dev": (C.c_int, [vp, u8p, C.c_uint64, u8p, u32p, u32p, u32p]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise RuntimeError(f"{path} lacks symbols declared in include/stralg_amd.h: {missing}")
    return lib


EXPORTS = ["sx_device_count", "sx_device_numa_node", "sx_ctx_create", "sx_ctx_destroy", "sx_ctx_live_count", "sx_last_error", "sx_ctx_trim", "sx_ctx_set_flag",
           "sx_sa_build", "sx_sa_build_dev", "sx_sa_bwt_build_dev", "sx_bwt_tables", "sx_bwt_tables_dev",
           "sx_bwt_tables_from_bwt_dev", "sx_build_tables", "sx_sa_inverse_dev", "sx_sa_lcp_dev", "sx_sa_inverse_lcp",
           "sx_bwt_exact_search_dev", "sx_build_tables_stream", "sx_fasta_pack_dev", "sx_fasta_pack", "sx_remap_dev", "sx_reverse_dev", "sx_profile_enable", "sx_profile_only",
           "sx_profile_reset", "sx_profile_read", "sx_kernel_class_name", "sx_last_stats",
           "sx_synth_dev", "sx_membw_probe", "sx_prim_sort_pairs_dev", "sx_prim_exclusive_sum_dev", "sx_prim_classify_dev"]


def kernel_sources_sha16():
    """SHA-256 (first 16 hex digits) over the kernel sources the product library is built from (stralg_amd/csrc/*.hip,
    *.hpp, in name order): what a rocprofv3 PMC pass is stamped with (tools/pmc_to_json.py) so that bench.py can tell when
    the kernels have changed since the committed traffic figures were measured (there is no .git on the GPU box)."""
    import glob
    import hashlib
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(here, "*.hip")) + glob.glob(os.path.join(here, "*.hpp"))):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
