"""Bit-exactness against the UNMODIFIED reference at BASELINE's sizes.  tests/golden/golden_big.npz holds what the
reference's sa_is_mem_construction (sa_is_mem.c:471-494) produced on synth(2^28 | 2^30, 5 | 256, 42)
(tests/golden/make_golden_big.py): SHA-256 of the whole suffix array and of every 2^26-entry chunk, every 2^20-th entry, the
BWT's SHA-256, the symbol counts (= C table and last O row).  reference_pin checks a device-resident array; host_table_pin
checks what a caller of the reference-named C API holds in host memory after build_complete_table / sa_is_construction
(stralg/bwt.c:134-161, sa_is.c:466-509): sa->array, c_table, the O rows through o_indices."""
import hashlib
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN_BIG = os.path.join(ROOT, "tests", "golden", "golden_big.npz")
CHUNK = 1 << 26


def fixture(n, sigma, seed=42):
    """(npz, key prefix) when golden_big.npz holds the reference's output for synth(n, sigma, seed), else (None, None)"""
    log2n = n.bit_length() - 1
    if n != 1 << log2n or seed != 42 or not os.path.exists(GOLDEN_BIG):
        return None, None
    z = np.load(GOLDEN_BIG)
    key = f"n{log2n}/s{sigma}"
    return (z, key) if key + "/sa_sha256" in z.files else (None, None)


def reference_pin(sa, n, sigma, seed):
    """SHA-256 of a device-resident suffix array (torch int32, downloaded in chunks) against the reference's.  None when no
    fixture exists for this text."""
    z, key = fixture(n, sigma, seed)
    if z is None:
        return None
    h = hashlib.sha256()
    for s0 in range(0, n + 1, CHUNK):
        h.update(sa[s0:s0 + CHUNK].cpu().numpy().tobytes())
    return {"fixture": f"tests/golden/golden_big.npz:{key}/sa_sha256", "sha256": h.hexdigest()[:16] + "...",
            "match": h.digest() == bytes(z[key + "/sa_sha256"]),
            "what": "SHA-256 of the whole suffix array vs the reference's sa_is_mem_construction (sa_is_mem.c:471-494) on the same text"}


def host_sa_pin(array, n, sigma, seed=42, whole=True):
    """array: numpy uint32 view of a host suffix array of n + 1 entries.  Chunk hashes (in parallel: hashlib releases the
    GIL), the sampled entries, and -- whole=True -- the SHA-256 of the whole array, against the reference's."""
    z, key = fixture(n, sigma, seed)
    if z is None:
        return None
    N = n + 1
    assert array.dtype == np.uint32 and array.size == N
    want_chunks = z[key + "/sa_chunk_sha256"]
    starts = list(range(0, N, CHUNK))

    def chunk_sha(s0):
        return hashlib.sha256(array[s0:s0 + CHUNK].data).digest()

    def whole_sha():
        h = hashlib.sha256()
        for s0 in starts:
            h.update(array[s0:s0 + CHUNK].data)
        return h.digest()

    with ThreadPoolExecutor(max_workers=min(8, len(starts) + 1)) as ex:
        fw = ex.submit(whole_sha) if whole else None
        got_chunks = list(ex.map(chunk_sha, starts))
        got_whole = fw.result() if fw else None
    bad = [k for k, g in enumerate(got_chunks) if g != bytes(want_chunks[k])]
    sampled = np.concatenate([array[:: 1 << 20], array[-1:]])
    out = {"fixture": f"tests/golden/golden_big.npz:{key}", "chunks": len(starts), "chunks_differing": bad,
           "sampled_match": bool((sampled == z[key + "/sa_sampled"]).all())}
    if whole:
        out["sha256"] = got_whole.hex()[:16] + "..."
        out["sha256_match"] = got_whole == bytes(z[key + "/sa_sha256"])
    out["match"] = not bad and out["sampled_match"] and (not whole or out["sha256_match"])
    return out


def host_table_pin(table, n, seed=42, rows_sampled=4096, whole=True):
    """table: cabi.BT (the struct build_complete_table returned) for the letters of synth(n, 5, seed).  Checks, on the
    host arrays a stralg caller reads: sa->length, sa->array (host_sa_pin), c_table and the last O row O(a, N) THROUGH
    o_indices against the reference's symbol counts (bwt.c:35-45, 47-65), O(a, 0) == 0, and for `rows_sampled` random
    rows i that O(., i+1) - O(., i) is one-hot at bwt[i] = string[sa[i] - 1] (bwt.c:13-20), again through o_indices (so
    the pointer table is checked where it is used).  The same for ro_indices against the counts (row 0, last row), when the
    table has a reverse direction."""
    t = table.contents if hasattr(table, "contents") else table
    sigma = int(t.remap_table.contents.alphabet_size)
    z, key = fixture(n, sigma, seed)
    if z is None:
        return None
    N = n + 1
    out = {"length_ok": int(t.sa.contents.length) == N}
    array = np.ctypeslib.as_array(t.sa.contents.array, shape=(N,))
    string = np.ctypeslib.as_array(t.sa.contents.string, shape=(N,))
    out["sa"] = host_sa_pin(array, n, sigma, seed, whole=whole)
    counts = z[key + "/counts"].astype(np.int64)
    c_want = np.concatenate([[0], np.cumsum(counts)[:-1]])
    out["c_table_match"] = bool((np.ctypeslib.as_array(t.c_table, shape=(sigma,)).astype(np.int64) == c_want).all())

    def row(idx, i):
        return np.ctypeslib.as_array(idx[i], shape=(sigma,)).astype(np.int64)

    out["o_last_row_match"] = bool((row(t.o_indices, N) == counts).all()) and bool((row(t.o_indices, 0) == 0).all())
    # the row pointers are o_table + i * sigma (bwt.c:53-57): first, last and sampled ones
    base = C_addr(t.o_table)
    rng = np.random.default_rng(seed)
    rows = np.unique(np.concatenate([[0, 1, N - 2, N - 1], rng.integers(0, N, size=rows_sampled)]))
    ok_ptr, ok_hot = True, True
    for i in rows.tolist():
        ok_ptr = ok_ptr and C_addr(t.o_indices[i]) == base + 4 * sigma * i and C_addr(t.o_indices[i + 1]) == base + 4 * sigma * (i + 1)
        p = int(array[i])
        b = int(string[p - 1]) if p else 0
        d = row(t.o_indices, i + 1) - row(t.o_indices, i)
        ok_hot = ok_hot and int(d[b]) == 1 and int(d.sum()) == 1
    out["o_indices_point_into_o_table"] = ok_ptr
    out["o_rows_sampled_one_hot_at_bwt"] = ok_hot
    out["rows_sampled"] = int(rows.size)
    ok = out["length_ok"] and out["sa"]["match"] and out["c_table_match"] and out["o_last_row_match"] and ok_ptr and ok_hot
    if t.ro_table:
        rbase = C_addr(t.ro_table)
        ro_ok = bool((row(t.ro_indices, N) == counts).all()) and bool((row(t.ro_indices, 0) == 0).all())
        for i in rows.tolist():
            ro_ok = ro_ok and C_addr(t.ro_indices[i]) == rbase + 4 * sigma * i
            d = row(t.ro_indices, i + 1) - row(t.ro_indices, i)
            ro_ok = ro_ok and int(d.sum()) == 1 and int(d.max()) == 1
        out["ro_rows_match_counts_and_step_by_one"] = ro_ok
        ok = ok and ro_ok
    out["match"] = bool(ok)
    return out


def C_addr(p):
    import ctypes as C
    return C.cast(p, C.c_void_p).value or 0
