"""ctypes bindings for oracle/liboracle.so and, where it has been built, for the
unmodified reference library oracle/_ref/libstralg_ref.so.

TEST INFRASTRUCTURE ONLY -- see oracle/oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

__all__ = [
    "build", "sa_is", "sa_is_strict", "sa_naive", "bwt", "c_table", "o_table", "remap",
    "check_sa", "last_levels", "synth", "have_ref", "ref", "inverse", "lcp", "bwt_exact_search",
]


def build(ref=True):
    """Compile liboracle.so (and _ref/libstralg_ref.so when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if ref and os.path.isdir("/root/reference/stralg"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        lib = C.CDLL(path)
        P8, P32 = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        lib.oracle_sa_is.argtypes = [P8, C.c_size_t, C.c_uint32, P32]
        lib.oracle_sa_is.restype = C.c_int
        lib.oracle_sa_is_strict.argtypes = [P8, C.c_size_t, C.c_uint32, P32]
        lib.oracle_sa_is_strict.restype = C.c_int
        lib.oracle_sa_naive.argtypes = [P8, C.c_size_t, P32]
        lib.oracle_sa_naive.restype = C.c_int
        lib.oracle_bwt.argtypes = [P8, P32, C.c_size_t, P8]
        lib.oracle_bwt.restype = None
        lib.oracle_c_table.argtypes = [P8, C.c_size_t, C.c_uint32, P32]
        lib.oracle_c_table.restype = None
        lib.oracle_o_table.argtypes = [P8, P32, C.c_size_t, C.c_uint32, P32]
        lib.oracle_o_table.restype = None
        lib.oracle_remap.argtypes = [P8, C.c_size_t, P8, C.POINTER(C.c_int16)]
        lib.oracle_remap.restype = C.c_uint32
        lib.oracle_inverse.argtypes = [P32, C.c_size_t, P32]
        lib.oracle_inverse.restype = None
        lib.oracle_lcp.argtypes = [P8, P32, C.c_size_t, P32]
        lib.oracle_lcp.restype = None
        lib.oracle_bwt_exact_search.argtypes = [P32, P32, C.c_size_t, C.c_uint32, P8, C.c_size_t, P32, P32]
        lib.oracle_bwt_exact_search.restype = None
        lib.oracle_check_sa.argtypes = [P8, C.c_size_t, P32]
        lib.oracle_check_sa.restype = C.c_int
        lib.oracle_last_levels.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        lib.oracle_last_levels.restype = C.c_int
        lib.oracle_synth.argtypes = [P8, C.c_size_t, C.c_uint32, C.c_uint64]
        lib.oracle_synth.restype = None
        _LIB = lib
    return _LIB


def _text(t):
    """bytes / uint8 array of symbols (no terminator) -> contiguous array with a 0 appended."""
    a = np.frombuffer(bytes(t), dtype=np.uint8) if isinstance(t, (bytes, bytearray)) else np.asarray(t, dtype=np.uint8)
    buf = np.zeros(a.size + 1, dtype=np.uint8)
    buf[: a.size] = a
    return buf, a.size


def sa_is(text, sigma):
    """stralg/sa_is.c:466 sa_is_construction: suffix array (n+1 uint32) of text + sentinel."""
    buf, n = _text(text)
    out = np.empty(n + 1, dtype=np.uint32)
    if _lib().oracle_sa_is(_u8(buf), n, sigma, _u32(out)) != 0:
        raise MemoryError("oracle_sa_is")
    return out


def sa_is_strict(text, sigma):
    buf, n = _text(text)
    out = np.empty(n + 1, dtype=np.uint32)
    if _lib().oracle_sa_is_strict(_u8(buf), n, sigma, _u32(out)) != 0:
        raise MemoryError("oracle_sa_is_strict")
    return out


def sa_naive(text):
    buf, n = _text(text)
    out = np.empty(n + 1, dtype=np.uint32)
    _lib().oracle_sa_naive(_u8(buf), n, _u32(out))
    return out


def bwt(text, sa):
    buf, n = _text(text)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    out = np.empty(n + 1, dtype=np.uint8)
    _lib().oracle_bwt(_u8(buf), _u32(sa), n + 1, _u8(out))
    return out


def c_table(text, sigma):
    buf, n = _text(text)
    out = np.empty(sigma, dtype=np.uint32)
    _lib().oracle_c_table(_u8(buf), n + 1, sigma, _u32(out))
    return out


def o_table(text, sa, sigma):
    """Position-major O table, shape (N+1, sigma)."""
    buf, n = _text(text)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    out = np.empty((n + 2, sigma), dtype=np.uint32)
    _lib().oracle_o_table(_u8(buf), _u32(sa), n + 1, sigma, _u32(out))
    return out


def inverse(sa):
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    out = np.empty(sa.size, dtype=np.uint32)
    _lib().oracle_inverse(_u32(sa), sa.size, _u32(out))
    return out


def lcp(text, sa):
    """stralg/suffix_array.c:62-85 compute_lcp."""
    buf, n = _text(text)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    out = np.empty(n + 1, dtype=np.uint32)
    _lib().oracle_lcp(_u8(buf), _u32(sa), n + 1, _u32(out))
    return out


def bwt_exact_search(c, o, sigma, pattern):
    """stralg/bwt.c:164-199: final (L, R) of the backward search; o has shape (N+1, sigma)."""
    c = np.ascontiguousarray(c, dtype=np.uint32)
    o = np.ascontiguousarray(o, dtype=np.uint32)
    pat = np.ascontiguousarray(pattern, dtype=np.uint8)
    L, R = C.c_uint32(0), C.c_uint32(0)
    _lib().oracle_bwt_exact_search(_u32(c), _u32(o), o.shape[0] - 1, sigma, _u8(pat) if pat.size else _u8(np.zeros(1, np.uint8)),
                                   pat.size, C.byref(L), C.byref(R))
    return int(L.value), int(R.value)


def remap(raw):
    """stralg/remap.c: returns (remapped symbols without terminator, alphabet_size, table[256])."""
    buf, n = _text(raw)
    out = np.empty(n + 1, dtype=np.uint8)
    table = np.empty(256, dtype=np.int16)
    sigma = _lib().oracle_remap(_u8(buf), n, _u8(out), table.ctypes.data_as(C.POINTER(C.c_int16)))
    return out[:n].copy(), int(sigma), table


def fasta_pack(data):
    """bioinf/fasta.c packing: returns (malformed, packed image bytes, [(name, sequence), ...] in file order)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    out = np.zeros(buf.size + 2, dtype=np.uint8)
    plen = C.c_size_t(0)
    nrec = C.c_uint32(0)
    lib = _lib()
    lib.oracle_fasta_pack.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_uint32)]
    lib.oracle_fasta_pack.restype = C.c_int
    src = buf if buf.size else np.zeros(1, np.uint8)
    bad = lib.oracle_fasta_pack(_u8(src), buf.size, _u8(out), C.byref(plen), C.byref(nrec))
    packed = out[:plen.value].tobytes()
    return bool(bad), packed, fasta_records_of(packed, nrec.value)


def fasta_records_of(packed, n_records):
    """split a packed image into its (name, sequence) pairs"""
    parts = packed.split(b"\0")
    return [(parts[2 * r], parts[2 * r + 1]) for r in range(n_records)]


def check_sa(text, sa):
    buf, n = _text(text)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    assert sa.size == n + 1
    return bool(_lib().oracle_check_sa(_u8(buf), n, _u32(sa)))


def last_levels():
    ns = (C.c_uint64 * 64)()
    ms = (C.c_uint64 * 64)()
    k = _lib().oracle_last_levels(ns, ms, 64)
    return [(int(ns[i]), int(ms[i])) for i in range(k)]


def synth(n, sigma, seed):
    out = np.empty(n, dtype=np.uint8)
    _lib().oracle_synth(_u8(out), n, sigma, seed)
    return out


# ---------------------------------------------------------------------------
# The unmodified reference, oracle/_ref/libstralg_ref.so: built in the build container from /root/reference by
# oracle/Makefile; the built library (never the sources) travels to the GPU box with the snapshot, where only
# bench.py's cpu_baseline leg and the tests load it.  The product never does.
# ---------------------------------------------------------------------------

class _RefSA(C.Structure):
    # stralg/suffix_array.h:10-20
    _fields_ = [("string", C.POINTER(C.c_uint8)), ("length", C.c_uint32),
                ("array", C.POINTER(C.c_uint32)), ("inverse", C.POINTER(C.c_uint32)),
                ("lcp", C.POINTER(C.c_uint32))]


class _RefRemap(C.Structure):
    # stralg/remap.h:9-19
    _fields_ = [("alphabet_size", C.c_uint32), ("table", C.c_byte * 256),
                ("rev_table", C.c_byte * 128)]


class _RefBwt(C.Structure):
    # stralg/bwt.h:36-44
    _fields_ = [("remap_table", C.POINTER(_RefRemap)), ("sa", C.POINTER(_RefSA)),
                ("c_table", C.POINTER(C.c_uint32)), ("o_table", C.POINTER(C.c_uint32)),
                ("o_indices", C.POINTER(C.POINTER(C.c_uint32))),
                ("ro_table", C.POINTER(C.c_uint32)),
                ("ro_indices", C.POINTER(C.POINTER(C.c_uint32)))]


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libstralg_ref.so"))


class _Ref:
    """Thin driver around the reference's own entry points."""

    def __init__(self):
        lib = C.CDLL(os.path.join(_HERE, "_ref", "libstralg_ref.so"))
        P8 = C.POINTER(C.c_uint8)
        for name in ("sa_is_construction", "sa_is_mem_construction"):
            f = getattr(lib, name)
            f.argtypes = [P8, C.c_uint32]
            f.restype = C.POINTER(_RefSA)
        for name in ("skew_sa_construction", "qsort_sa_construction"):
            f = getattr(lib, name)
            f.argtypes = [P8]
            f.restype = C.POINTER(_RefSA)
        lib.free_suffix_array.argtypes = [C.POINTER(_RefSA)]
        lib.free_suffix_array.restype = None
        lib.remap_string.argtypes = [P8, P8]
        lib.remap_string.restype = C.c_uint32
        lib.build_complete_table.argtypes = [P8, C.c_bool]
        lib.build_complete_table.restype = C.POINTER(_RefBwt)
        lib.completely_free_bwt_table.argtypes = [C.POINTER(_RefBwt)]
        lib.completely_free_bwt_table.restype = None
        self.lib = lib

    def _sa(self, fn, buf, *args):
        sa = fn(_u8(buf), *args)
        out = np.ctypeslib.as_array(sa.contents.array, shape=(sa.contents.length,)).copy()
        self.lib.free_suffix_array(sa)
        return out

    def sa_is(self, text, sigma):
        buf, _ = _text(text)
        return self._sa(self.lib.sa_is_construction, buf, sigma)

    def sa_is_mem(self, text, sigma):
        buf, _ = _text(text)
        return self._sa(self.lib.sa_is_mem_construction, buf, sigma)

    def skew(self, text):
        buf, _ = _text(text)
        return self._sa(self.lib.skew_sa_construction, buf)

    def qsort(self, text):
        buf, _ = _text(text)
        return self._sa(self.lib.qsort_sa_construction, buf)

    def fasta(self, data):
        """load_fasta_records on a temporary file: (error code, [(name, sequence), ...] in ITERATION order,
        i.e. the reverse of the file order, bioinf/fasta.c:131-134)."""
        import tempfile

        class _Rec(C.Structure):
            _fields_ = [("name", C.c_char_p), ("seq", C.POINTER(C.c_uint8)), ("seq_len", C.c_uint32)]

        class _Iter(C.Structure):
            _fields_ = [("rec", C.c_void_p)]

        lib = self.lib
        lib.load_fasta_records.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        lib.load_fasta_records.restype = C.c_void_p
        lib.free_fasta_records.argtypes = [C.c_void_p]
        lib.init_fasta_iter.argtypes = [C.POINTER(_Iter), C.c_void_p]
        lib.next_fasta_record.argtypes = [C.POINTER(_Iter), C.POINTER(_Rec)]
        lib.next_fasta_record.restype = C.c_bool
        with tempfile.NamedTemporaryFile(suffix=".fa") as f:
            f.write(bytes(data))
            f.flush()
            err = C.c_int(0)
            devnull = os.open(os.devnull, os.O_WRONLY)
            saved = os.dup(2)
            os.dup2(devnull, 2)  # the loader reports every record on stderr
            try:
                h = lib.load_fasta_records(f.name.encode(), C.byref(err))
            finally:
                os.dup2(saved, 2)
                os.close(saved)
                os.close(devnull)
        if not h:
            return err.value, []
        it, rec, out = _Iter(), _Rec(), []
        lib.init_fasta_iter(C.byref(it), h)
        while lib.next_fasta_record(C.byref(it), C.byref(rec)):
            out.append((rec.name, bytes(bytearray(rec.seq[i] for i in range(rec.seq_len)))))
        lib.free_fasta_records(h)
        return err.value, out

    def serialise(self, raw, include_reverse=True):
        """bytes of write_complete_bwt_info(build_complete_table(raw, include_reverse)) (stralg/serialise.c:7-18)"""
        import tempfile
        lib = self.lib
        lib.write_complete_bwt_info_fname.argtypes = [C.c_char_p, C.POINTER(_RefBwt)]
        lib.write_complete_bwt_info_fname.restype = None
        buf, _ = _text(raw)
        t = lib.build_complete_table(_u8(buf), include_reverse)
        with tempfile.NamedTemporaryFile(suffix=".bwt") as f:
            lib.write_complete_bwt_info_fname(f.name.encode(), t)
            data = open(f.name, "rb").read()
        lib.completely_free_bwt_table(t)
        return data

    def remap_string(self, raw):
        buf, n = _text(raw)
        out = np.zeros(n + 1, dtype=np.uint8)
        sigma = self.lib.remap_string(_u8(out), _u8(buf))
        return out[:n].copy(), int(sigma)

    def lcp(self, text, sigma):
        """compute_lcp on the reference's own sa_is suffix array; returns (sa, inverse, lcp)."""
        buf, _ = _text(text)
        self.lib.compute_lcp.argtypes = [C.POINTER(_RefSA)]
        self.lib.compute_lcp.restype = None
        sa = self.lib.sa_is_construction(_u8(buf), sigma)
        self.lib.compute_lcp(sa)
        N = sa.contents.length
        out = tuple(np.ctypeslib.as_array(p, shape=(N,)).copy() for p in (sa.contents.array, sa.contents.inverse, sa.contents.lcp))
        self.lib.free_suffix_array(sa)
        return out

    def exact_search(self, raw, patterns):
        """(L, R) of init_bwt_exact_match_iter for every pattern (bytes of the raw alphabet)."""
        class Iter(C.Structure):
            _fields_ = [("sa", C.c_void_p), ("L", C.c_uint32), ("i", C.c_int64), ("R", C.c_uint32)]
        buf, _ = _text(raw)
        t = self.lib.build_complete_table(_u8(buf), False)
        self.lib.init_bwt_exact_match_iter.argtypes = [C.POINTER(Iter), C.POINTER(_RefBwt), C.POINTER(C.c_uint8)]
        self.lib.init_bwt_exact_match_iter.restype = None
        self.lib.remap.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(_RefRemap)]
        self.lib.remap.restype = C.c_void_p
        res = []
        for pat in patterns:
            pb, m = _text(pat)
            rp = np.zeros(m + 1, dtype=np.uint8)
            ok = self.lib.remap(_u8(rp), _u8(pb), t.contents.remap_table)
            if not ok:
                res.append(None)  # a letter that is not in the text's alphabet
                continue
            it = Iter()
            self.lib.init_bwt_exact_match_iter(C.byref(it), t, _u8(rp))
            res.append((int(it.L), int(it.R), rp[:m].copy()))
        self.lib.completely_free_bwt_table(t)
        return res

    def build_complete_table(self, raw, include_reverse=True):
        """Returns dict(sigma, remapped, sa, c, o[(N+1), sigma], ro or None)."""
        buf, n = _text(raw)
        t = self.lib.build_complete_table(_u8(buf), include_reverse)
        bt = t.contents
        sigma = int(bt.remap_table.contents.alphabet_size)
        N = int(bt.sa.contents.length)
        res = {
            "sigma": sigma,
            "remapped": np.ctypeslib.as_array(bt.sa.contents.string, shape=(N,)).copy()[:n],
            "sa": np.ctypeslib.as_array(bt.sa.contents.array, shape=(N,)).copy(),
            "c": np.ctypeslib.as_array(bt.c_table, shape=(sigma,)).copy(),
            "o": np.ctypeslib.as_array(bt.o_table, shape=((N + 1) * sigma,)).copy().reshape(N + 1, sigma),
            "ro": None,
        }
        if include_reverse:
            res["ro"] = np.ctypeslib.as_array(bt.ro_table, shape=((N + 1) * sigma,)).copy().reshape(N + 1, sigma)
        self.lib.completely_free_bwt_table(t)
        return res


def ref():
    global _REF
    if _REF is None:
        _REF = _Ref()
    return _REF
