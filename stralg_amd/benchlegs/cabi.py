"""ctypes views of the reference's structs (include/stralg_compat.h: stralg/suffix_array.h:10-20, remap.h:9-19,
bwt.h:36-44) and the prototypes of the reference-named entry points: what a C caller of libstralg sees."""
import ctypes as C


class SA(C.Structure):
    _fields_ = [("string", C.POINTER(C.c_uint8)), ("length", C.c_uint32), ("array", C.POINTER(C.c_uint32)),
                ("inverse", C.c_void_p), ("lcp", C.c_void_p)]


class RT(C.Structure):
    _fields_ = [("alphabet_size", C.c_uint32), ("table", C.c_byte * 256), ("rev_table", C.c_byte * 128)]


class BT(C.Structure):
    _fields_ = [("remap_table", C.POINTER(RT)), ("sa", C.POINTER(SA)), ("c_table", C.POINTER(C.c_uint32)),
                ("o_table", C.POINTER(C.c_uint32)), ("o_indices", C.POINTER(C.POINTER(C.c_uint32))),
                ("ro_table", C.POINTER(C.c_uint32)), ("ro_indices", C.POINTER(C.POINTER(C.c_uint32)))]


def declare(lib):
    """argtypes / restypes of the reference-named functions used by the tests and bench legs; returns lib"""
    lib.build_complete_table.argtypes = [C.c_void_p, C.c_bool]
    lib.build_complete_table.restype = C.POINTER(BT)
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    for fn in (lib.sa_is_construction, lib.sa_is_mem_construction):
        fn.argtypes = [C.c_void_p, C.c_uint32]
        fn.restype = C.POINTER(SA)
    lib.skew_sa_construction.argtypes = [C.c_void_p]
    lib.skew_sa_construction.restype = C.POINTER(SA)
    lib.free_suffix_array.argtypes = [C.POINTER(SA)]
    lib.free_suffix_array.restype = None
    lib.write_complete_bwt_info.argtypes = [C.c_void_p, C.POINTER(BT)]
    lib.write_complete_bwt_info.restype = None
    lib.stralg_amd_release.argtypes = []
    lib.stralg_amd_release.restype = None
    return lib
