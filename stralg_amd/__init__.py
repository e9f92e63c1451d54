"""stralg_amd -- MI355X-native suffix-array / BWT-table construction behind stralg's C API.

Python mirror of the reference interface for this path (same names, argument
meaning and error behaviour as stralg/suffix_array.h and stralg/bwt.h); the
work is done by libstralg_amd.so (HIP kernels for gfx950) through the C-ABI of
include/stralg_amd.h.  There is no CPU fallback.
"""
from .api import (  # noqa: F401
    Context, SuffixArray, BwtTable, RemapTable, StralgAmdError,
    sa_is_construction, sa_is_mem_construction, skew_sa_construction,
    remap_string, alloc_remap_table, remap, init_bwt_table, build_complete_table,
    default_context,
)
from .synth import synth  # noqa: F401
