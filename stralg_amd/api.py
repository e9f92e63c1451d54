"""Host-side mirror of stralg's interface for the suffix-array / BWT-table path.

Names, argument meaning and results follow the reference declarations cited on
each function; arrays come back as numpy arrays instead of malloc'd pointers.
All work happens in libstralg_amd.so (include/stralg_amd.h); a missing library
or GPU raises -- there is no CPU fallback.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib


class StralgAmdError(RuntimeError):
    pass


def _ptr(x):
    """Raw address of a numpy array, a torch tensor, an int address or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    raise TypeError(f"cannot take the address of {type(x)!r}")


class Context:
    """One device context (sx_ctx): a HIP stream plus cached workspace on one GPU."""

    def __init__(self, device=0, lib_path=None):
        self.lib = _lib.load(lib_path)
        self.device = device
        h = C.c_void_p()
        rc = self.lib.sx_ctx_create(device, C.byref(h))
        if rc != 0 or not h:
            raise StralgAmdError(
                f"sx_ctx_create(device={device}) failed with code {rc}: no usable GPU "
                "(stralg_amd has no CPU fallback)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.sx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.sx_last_error(self.h)
            raise StralgAmdError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")

    # ---- host-buffer entry points ------------------------------------------------
    def sa_build(self, text, alphabet_size):
        """sx_sa_build: text = uint8 symbols in [1, alphabet_size) without terminator."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        out = np.empty(text.size + 1, dtype=np.uint32)
        self._check(self.lib.sx_sa_build(self.h, _ptr(text), text.size, alphabet_size, _ptr(out)), "sx_sa_build")
        return out

    def bwt_tables(self, text, sa, sigma, want_o=True):
        """sx_bwt_tables: returns (c_table[sigma], o_table[(N+1), sigma] or None)."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        sa = np.ascontiguousarray(sa, dtype=np.uint32)
        N = sa.size
        if text.size != N - 1:
            raise ValueError("sa must have len(text) + 1 entries")
        c = np.zeros(sigma, dtype=np.uint32)
        o = np.empty((N + 1, sigma), dtype=np.uint32) if want_o else None
        self._check(self.lib.sx_bwt_tables(self.h, _ptr(text), _ptr(sa), N, sigma, _ptr(c), _ptr(o)), "sx_bwt_tables")
        return c, o

    def build_tables(self, text, sigma, want_sa=True, want_o=True):
        """sx_build_tables: (sa or None, c_table, o_table or None) with one text upload."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        N = text.size + 1
        sa = np.empty(N, dtype=np.uint32) if want_sa else None
        c = np.zeros(sigma, dtype=np.uint32)
        o = np.empty((N + 1, sigma), dtype=np.uint32) if want_o else None
        self._check(self.lib.sx_build_tables(self.h, _ptr(text), text.size, sigma, _ptr(sa), _ptr(c), _ptr(o)),
                    "sx_build_tables")
        return sa, c, o

    # ---- device-buffer entry points (torch tensors or raw addresses) --------------
    def sa_build_dev(self, d_text, n, alphabet_size, d_sa_out):
        self._check(self.lib.sx_sa_build_dev(self.h, _ptr(d_text), n, alphabet_size, _ptr(d_sa_out)), "sx_sa_build_dev")

    def sa_bwt_build_dev(self, d_text, n, alphabet_size, d_sa_out, d_bwt_out):
        self._check(self.lib.sx_sa_bwt_build_dev(self.h, _ptr(d_text), n, alphabet_size, _ptr(d_sa_out),
                                                 _ptr(d_bwt_out)), "sx_sa_bwt_build_dev")

    def bwt_tables_from_bwt_dev(self, d_bwt, N, sigma, d_c_out, d_o_out=None):
        self._check(self.lib.sx_bwt_tables_from_bwt_dev(self.h, _ptr(d_bwt), N, sigma, _ptr(d_c_out), _ptr(d_o_out)),
                    "sx_bwt_tables_from_bwt_dev")

    def bwt_tables_dev(self, d_text, d_sa, N, sigma, d_c_out, d_o_out=None, d_bwt_out=None):
        self._check(self.lib.sx_bwt_tables_dev(self.h, _ptr(d_text), _ptr(d_sa), N, sigma, _ptr(d_c_out),
                                               _ptr(d_o_out), _ptr(d_bwt_out)), "sx_bwt_tables_dev")

    def synth_dev(self, d_out, n, sigma, seed):
        self._check(self.lib.sx_synth_dev(self.h, _ptr(d_out), n, sigma, seed), "sx_synth_dev")

    # ---- consumers of a resident suffix array / table ---------------------------------
    def membw_probe(self, d_a, d_b, nbytes, reps=5):
        """the box's streaming rates in GB/s over two device buffers of nbytes each: read, fill, copy, four-way split"""
        out = (C.c_double * 4)()
        self._check(self.lib.sx_membw_probe(self.h, _ptr(d_a), _ptr(d_b), nbytes, reps, out), "sx_membw_probe")
        return dict(zip(("read", "fill", "copy", "split4"), (float(v) for v in out)))

    def inverse_lcp(self, text, sa, want_lcp=True):
        """sx_sa_inverse_lcp: (inverse, lcp or None) as compute_inverse / compute_lcp (suffix_array.c:53-85)."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        sa = np.ascontiguousarray(sa, dtype=np.uint32)
        inv = np.empty(sa.size, dtype=np.uint32)
        lcp = np.empty(sa.size, dtype=np.uint32) if want_lcp else None
        self._check(self.lib.sx_sa_inverse_lcp(self.h, _ptr(text), _ptr(sa), sa.size, _ptr(inv), _ptr(lcp)),
                    "sx_sa_inverse_lcp")
        return inv, lcp

    def sa_inverse_dev(self, d_sa, N, d_inv):
        self._check(self.lib.sx_sa_inverse_dev(self.h, _ptr(d_sa), N, _ptr(d_inv)), "sx_sa_inverse_dev")

    def sa_lcp_dev(self, d_text, d_sa, N, d_inv, d_lcp):
        self._check(self.lib.sx_sa_lcp_dev(self.h, _ptr(d_text), _ptr(d_sa), N, _ptr(d_inv), _ptr(d_lcp)), "sx_sa_lcp_dev")

    def bwt_exact_search_dev(self, d_c, d_o, N, sigma, d_patterns, d_offsets, count, d_l, d_r):
        self._check(self.lib.sx_bwt_exact_search_dev(self.h, _ptr(d_c), _ptr(d_o), N, sigma, _ptr(d_patterns),
                                                     _ptr(d_offsets), count, _ptr(d_l), _ptr(d_r)),
                    "sx_bwt_exact_search_dev")

    # ---- FASTA ingest and remap ------------------------------
    def fasta_pack_dev(self, d_file, file_len, d_packed, d_term=None, term_cap=0):
        """bioinf/fasta.c load_fasta_records' packing on the device: returns (packed_len, n_records); raises
        StralgAmdError (code SX_E_MALFORMED = -4) where the reference reports MALFORMED_FILE."""
        plen, nrec = C.c_uint64(0), C.c_uint32(0)
        self._check(self.lib.sx_fasta_pack_dev(self.h, _ptr(d_file), file_len, _ptr(d_packed), C.byref(plen), _ptr(d_term),
                                               term_cap, C.byref(nrec)), "sx_fasta_pack_dev")
        return int(plen.value), int(nrec.value)

    def remap_dev(self, d_in, n, d_out):
        """stralg/remap.c on the device: d_out[0..n) dense codes, d_out[n] = 0; returns (alphabet_size, table[256])."""
        table = np.zeros(256, dtype=np.int16)
        sigma = C.c_uint32(0)
        self._check(self.lib.sx_remap_dev(self.h, _ptr(d_in), n, _ptr(d_out), table.ctypes.data_as(C.POINTER(C.c_int16)),
                                          C.byref(sigma)), "sx_remap_dev")
        return int(sigma.value), table

    def reverse_dev(self, d_in, n, d_out):
        """sx_reverse_dev: d_out[0..n) = d_in reversed, d_out[n] = 0 (the string build_complete_table sorts for RO, bwt.c:147-151)"""
        self._check(self.lib.sx_reverse_dev(self.h, _ptr(d_in), n, _ptr(d_out)), "sx_reverse_dev")

    def fasta_records(self, data):
        """sx_fasta_pack (host buffers): [(name, sequence), ...] in file order from the bytes of a FASTA file"""
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        packed = np.zeros(buf.size + 1, dtype=np.uint8)
        term = np.zeros(buf.size + 2, dtype=np.uint32)
        plen, nrec = C.c_uint64(0), C.c_uint32(0)
        self._check(self.lib.sx_fasta_pack(self.h, _ptr(buf) if buf.size else None, buf.size, _ptr(packed), C.byref(plen),
                                           _ptr(term), term.size, C.byref(nrec)), "sx_fasta_pack")
        out = []
        for r in range(nrec.value):
            n0 = 0 if r == 0 else int(term[2 * r - 1]) + 1
            s0 = int(term[2 * r]) + 1
            out.append((packed[n0:int(term[2 * r])].tobytes(), packed[s0:int(term[2 * r + 1])].tobytes()))
        return out

    # ---- primitives (kernel-level tests) ------------------------------------------
    def prim_sort_pairs_dev(self, ka, va, kb, vb, n, begin_bit, end_bit):
        flag = C.c_int(0)
        self._check(self.lib.sx_prim_sort_pairs_dev(self.h, _ptr(ka), _ptr(va), _ptr(kb), _ptr(vb), n, begin_bit,
                                                    end_bit, C.byref(flag)), "sx_prim_sort_pairs_dev")
        return bool(flag.value)

    def prim_exclusive_sum_dev(self, d_in, d_out, n, d_total=None):
        self._check(self.lib.sx_prim_exclusive_sum_dev(self.h, _ptr(d_in), _ptr(d_out), n, _ptr(d_total)),
                    "sx_prim_exclusive_sum_dev")

    def prim_classify_dev(self, d_text, n, d_flags, d_hist_all, d_hist_l, d_hist_lms):
        self._check(self.lib.sx_prim_classify_dev(self.h, _ptr(d_text), n, _ptr(d_flags), _ptr(d_hist_all),
                                                  _ptr(d_hist_l), _ptr(d_hist_lms)), "sx_prim_classify_dev")

    # ---- measurement -----------------------------------------------------------------
    def profile_only(self, kclass_name=None):
        """events only around launches of one kernel class (None: all classes)"""
        k = -1 if kclass_name is None else _lib.KC_NAMES.index(kclass_name)
        self._check(self.lib.sx_profile_only(self.h, k), "sx_profile_only")

    def profile_enable(self, on=True):
        self._check(self.lib.sx_profile_enable(self.h, 1 if on else 0), "sx_profile_enable")

    def profile_reset(self):
        self._check(self.lib.sx_profile_reset(self.h), "sx_profile_reset")

    def profile_read(self):
        arr = (_lib.KernelStat * len(_lib.KC_NAMES))()
        self._check(self.lib.sx_profile_read(self.h, arr), "sx_profile_read")
        return {name: {"launches": int(arr[i].launches), "ms": float(arr[i].ms), "alg_bytes": int(arr[i].alg_bytes)}
                for i, name in enumerate(_lib.KC_NAMES)}

    def last_stats(self):
        st = _lib.BuildStats()
        self._check(self.lib.sx_last_stats(self.h, C.byref(st)), "sx_last_stats")
        return st.as_dict()

    def force_general_path(self, on=True):
        """SX_FLAG_FORCE_GENERAL_PATH: pieces + names + prefix doubling even where the prefix-key sort would do."""
        self._check(self.lib.sx_ctx_set_flag(self.h, 1, 1 if on else 0), "sx_ctx_set_flag")

    def set_chain_max_entries(self, entries):
        """SX_FLAG_CHAIN_MAX_ENTRIES: longer induce rounds take the count / offsets / scatter launches."""
        self._check(self.lib.sx_ctx_set_flag(self.h, 2, int(entries)), "sx_ctx_set_flag")

    def set_no_direct_sort(self, on=True):
        """SX_FLAG_NO_DIRECT_SORT: wide alphabets take the LMS sort + induction even where the direct sort applies."""
        self._check(self.lib.sx_ctx_set_flag(self.h, 3, 1 if on else 0), "sx_ctx_set_flag")

    def set_prefix_symbols(self, symbols):
        """SX_FLAG_PREFIX_SYMBOLS: the prefix-key sort's first attempt takes this many symbols (0: by the size)."""
        self._check(self.lib.sx_ctx_set_flag(self.h, 4, int(symbols)), "sx_ctx_set_flag")

    def set_radix_digit_bits(self, bits):
        """SX_FLAG_RADIX_DIGIT_BITS: digit width of the LSD radix passes (8, 9, 10; 0: default)."""
        self._check(self.lib.sx_ctx_set_flag(self.h, 5, int(bits)), "sx_ctx_set_flag")

    def set_sort_mode(self, mode):
        """SX_FLAG_SORT_MODE: 0 choose, 1 LSD passes only, 2 hybrid sort wherever the key shape allows it (HBM passes on
        the top 24 key bits), 3 the same with the top 32 bits."""
        self._check(self.lib.sx_ctx_set_flag(self.h, 6, int(mode)), "sx_ctx_set_flag")

    def set_induce_batch(self, on=True):
        """SX_FLAG_INDUCE_BATCH_OFF: the self rounds of a bucket eight at a time (default) or a launch each"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 7, 0 if on else 1), "sx_ctx_set_flag")

    def set_induce_batch_min(self, entries):
        """SX_FLAG_INDUCE_BATCH_MIN: ranges longer than this take the eight-rounds-at-a-time form (negative: default)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 8, int(entries)), "sx_ctx_set_flag")

    def set_induce_attended(self, mode=1):
        """SX_FLAG_INDUCE_ATTENDED: 0 default (buckets queued one behind the other, a bucket the tail kernel could not
        finish is carried on by the host), 1 the host looks at every bucket's last range"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 9, int(mode)), "sx_ctx_set_flag")

    def set_induce_hoist(self, on=True):
        """SX_FLAG_INDUCE_NO_HOIST: texts of more than 8 symbols -- all buckets' LMS seeds / L-type entries scanned at once, up
        front, placed by the text's bigram counts (default), or by launches of each bucket's own (rounds 1 - 3)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 13, 0 if on else 1), "sx_ctx_set_flag")

    def set_text_keys(self, on=True):
        """SX_FLAG_TEXT_KEYS_OFF: the first radix pass of the direct sort, and of a four-letter text's LMS sort, computes
        its keys from the text (default) or reads them from a key kernel's output (rounds 1 - 3)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 14, 0 if on else 1), "sx_ctx_set_flag")

    def set_long_subbuckets(self, on=True):
        """SX_FLAG_LONG_SUBBUCKETS_OFF: the hybrid prefix-key sort lists sub-buckets too long for a workgroup and orders them
        by HBM passes of their own (default), or falls back to plain passes when it meets one (rounds 1 - 3)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 15, 0 if on else 1), "sx_ctx_set_flag")

    def set_local_sort_lean(self, on=True):
        """SX_FLAG_LOCAL_SORT_LEAN_OFF: the hybrid sort's LDS step by the lean kernel of round 5 (default), or every workgroup by
        the kernel of rounds 3 and 4 (the one the lean kernel leaves its crowded workgroups to)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 17, 0 if on else 1), "sx_ctx_set_flag")

    def set_small_direct_max(self, suffixes):
        """SX_FLAG_SMALL_DIRECT_MAX: texts of at most 16 symbols and at most this many suffixes are sorted directly
        (0: never; negative: default)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 16, int(suffixes)), "sx_ctx_set_flag")

    def set_recurse_min(self, symbols):
        """SX_FLAG_RECURSE_MIN: reduced strings of at most 255 names recurse from this length on (negative: default)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 11, int(symbols)), "sx_ctx_set_flag")

    def set_sample_min(self, suffixes):
        """SX_FLAG_SAMPLE_MIN: wide-alphabet texts of at least this many suffixes get the look at a sample (negative: default)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 12, int(suffixes)), "sx_ctx_set_flag")

    def set_copy_text_first(self, on=True):
        """SX_FLAG_COPY_TEXT_FIRST: a device copy of the text before the classification (True) or the copy made by the
        classification while it reads the caller's text (default)"""
        self._check(self.lib.sx_ctx_set_flag(self.h, 10, 1 if on else 0), "sx_ctx_set_flag")

    def trim(self):
        self.lib.sx_ctx_trim(self.h)

    def bind_to_numa_node(self):
        """pin the calling thread (and the threads it starts later) to the CPUs next to this context's GPU
        (stralg_amd_bind_thread_to_device); returns the NUMA node, or -1 when the box does not tell"""
        self.lib.stralg_amd_bind_thread_to_device.argtypes = [C.c_int]
        self.lib.stralg_amd_bind_thread_to_device.restype = C.c_int
        return int(self.lib.stralg_amd_bind_thread_to_device(self.device))


_tls = threading.local()


def default_context(device=None):
    """The calling thread's context (created on first use, like the C host layer's)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None or (device is not None and ctx.device != device):
        ctx = Context(device or 0)
        _tls.ctx = ctx
    return ctx


# ---------------------------------------------------------------------------
# reference-shaped results
# ---------------------------------------------------------------------------

class SuffixArray:
    """stralg/suffix_array.h:10-20: string (borrowed, with terminator), length, array."""

    def __init__(self, string, array):
        self.string = string
        self.length = int(array.size)
        self.array = array
        self.inverse = None
        self.lcp = None


class RemapTable:
    """stralg/remap.h:9-19."""

    def __init__(self, alphabet_size, table, rev_table):
        self.alphabet_size = alphabet_size
        self.table = table
        self.rev_table = rev_table


class BwtTable:
    """stralg/bwt.h:36-44; o_table / ro_table are (N+1, sigma) arrays: O(a, i) = o_table[i, a]."""

    def __init__(self, remap_table, sa, c_table, o_table, ro_table):
        self.remap_table = remap_table
        self.sa = sa
        self.c_table = c_table
        self.o_table = o_table
        self.ro_table = ro_table


def _symbols(x):
    """bytes-like or array without terminator -> uint8 array; stops at the first 0 like strlen."""
    a = np.frombuffer(bytes(x), dtype=np.uint8) if isinstance(x, (bytes, bytearray, memoryview)) else np.asarray(x, dtype=np.uint8)
    zero = np.flatnonzero(a == 0)
    return a[: zero[0]] if zero.size else a


def _with_terminator(a):
    out = np.zeros(a.size + 1, dtype=np.uint8)
    out[: a.size] = a
    return out


def sa_is_construction(remapped_string, alphabet_size, ctx=None):
    """stralg/suffix_array.h:31-35 (sa_is.c:466-509)."""
    ctx = ctx or default_context()
    text = _symbols(remapped_string)
    return SuffixArray(_with_terminator(text), ctx.sa_build(text, alphabet_size))


def sa_is_mem_construction(remapped_string, alphabet_size, ctx=None):
    """stralg/suffix_array.h:37-41 (sa_is_mem.c:471-494): same array, same device path."""
    return sa_is_construction(remapped_string, alphabet_size, ctx)


def skew_sa_construction(string, ctx=None):
    """stralg/suffix_array.h:26-29 (skew.c:388-395): raw bytes 1..255, alphabet fixed at 256."""
    return sa_is_construction(string, 256, ctx)


def alloc_remap_table(string):
    """stralg/remap.c:8-41: order-preserving dense codes, 0 reserved for the sentinel."""
    s = _symbols(string)
    present = np.zeros(256, dtype=bool)
    present[s] = True
    present[0] = False
    table = np.full(256, -1, dtype=np.int16)
    rev = np.full(128, -1, dtype=np.int16)
    table[0] = 0
    rev[0] = 0
    letters = np.flatnonzero(present)
    if letters.size > 127:
        raise StralgAmdError("more than 127 distinct letters: stralg's remap table cannot hold them (remap.h:14-18)")
    table[letters] = np.arange(1, letters.size + 1)
    rev[1: letters.size + 1] = letters
    return RemapTable(int(letters.size + 1), table, rev)


def remap(string, table):
    """stralg/remap.c:102-114; raises where the reference returns NULL (letter not in the table)."""
    s = _symbols(string)
    out = table.table[s]
    if (out < 0).any():
        raise StralgAmdError("remap: the string holds a letter that is not in the table")
    return out.astype(np.uint8)


def remap_string(string):
    """stralg/remap.c:155-165: returns (remapped symbols, alphabet_size)."""
    t = alloc_remap_table(string)
    return remap(string, t), t.alphabet_size


def init_bwt_table(sa, rsa, remap_table, ctx=None):
    """stralg/bwt.h:73-76 (bwt.c:22-89): C, O and (when rsa is given) RO tables."""
    ctx = ctx or default_context()
    sigma = remap_table.alphabet_size
    c, o = ctx.bwt_tables(sa.string[:-1], sa.array, sigma)
    ro = None
    if rsa is not None:
        _, ro = ctx.bwt_tables(rsa.string[:-1], rsa.array, sigma)
    return BwtTable(remap_table, sa, c, o, ro)


def build_complete_table(string, include_reverse=True, ctx=None):
    """stralg/bwt.h:156-160 (bwt.c:134-161): remap -> SA-IS -> [reverse, SA-IS] -> tables."""
    ctx = ctx or default_context()
    table = alloc_remap_table(string)
    remapped = remap(string, table)
    sigma = table.alphabet_size
    # one device pass per direction: the induced sort hands over the BWT with the suffix array
    sa_arr, c, o = ctx.build_tables(remapped, sigma)
    ro = None
    if include_reverse:
        _, _, ro = ctx.build_tables(remapped[::-1].copy(), sigma, want_sa=False)
    sa = SuffixArray(_with_terminator(remapped), sa_arr)
    return BwtTable(table, sa, c, o, ro)
