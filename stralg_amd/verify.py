"""Size-independent checks of a build's results, on the device with torch (test and bench
infrastructure: nothing here is on the product path).

A suffix array is the unique array that is a permutation of 0..n with sa[0] = n whose
suffixes increase strictly; "suffix a < suffix b" is decided in O(1) from the first symbols
and the ranks of a+1 and b+1, so the whole proof is O(n) gathers.  The tables follow
stralg/bwt.c:13-20 (bwt), 35-45 (C) and 47-65 (O): bwt[i] = text[sa[i]-1] (0 for sa[i] = 0),
C = exclusive cumulative symbol counts, O row 0 = 0, O row i+1 - O row i = one-hot(bwt[i]),
O row N = the symbol counts.  Everything is checked in chunks so that 2^30 (and 2^32 - 2)
symbols fit next to the build's own buffers.
"""

CHUNK = 1 << 26


def verify_sa_on_device(text_u8, sa_i32, n, chunk=CHUNK):
    """text_u8: n symbols (torch uint8), sa_i32: n+1 entries (torch int32 holding uint32 bits)."""
    import torch
    dev = sa_i32.device
    N = n + 1
    if (int(sa_i32[0]) & 0xFFFFFFFF) != n:
        raise AssertionError("sa[0] is not n (the sentinel suffix)")
    wide = N >= (1 << 31)
    rdt = torch.int64 if wide else torch.int32
    rank = torch.full((N + 1,), -1, dtype=rdt, device=dev)
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        pos = sa_i32[s:e].long() & 0xFFFFFFFF
        if int(pos.max()) > n:
            raise AssertionError("an entry exceeds n")
        rank[pos] = torch.arange(s, e, dtype=rdt, device=dev)
    if not bool((rank[:N] >= 0).all()):
        raise AssertionError("not a permutation")
    T = torch.zeros(N + 1, dtype=torch.uint8, device=dev)
    T[:n] = text_u8[:n]
    for s in range(1, N - 1, chunk):
        e = min(N - 1, s + chunk)
        a = sa_i32[s:e].long() & 0xFFFFFFFF
        b = sa_i32[s + 1:e + 1].long() & 0xFFFFFFFF
        ca, cb = T[a], T[b]
        ok = (ca < cb) | ((ca == cb) & (rank[a + 1] < rank[b + 1]))
        if not bool(ok.all()):
            raise AssertionError(f"suffixes out of order in slots [{s}, {e})")
    return True


def symbol_counts(text_u8, n, sigma, chunk=1 << 28):
    """counts[a] of text[0..n) plus one sentinel (symbol 0), int64 on the device"""
    import torch
    counts = torch.zeros(sigma, dtype=torch.int64, device=text_u8.device)
    for s in range(0, n, chunk):
        counts += torch.bincount(text_u8[s:min(n, s + chunk)].long(), minlength=sigma)[:sigma]
    counts[0] += 1
    return counts


def verify_bwt_on_device(text_u8, sa_i32, bwt_u8, n, chunk=CHUNK):
    """bwt[i] = text[sa[i] - 1], 0 where sa[i] = 0 (stralg/bwt.c:13-20)"""
    import torch
    N = n + 1
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        p = sa_i32[s:e].long() & 0xFFFFFFFF
        want = torch.where(p == 0, torch.zeros_like(bwt_u8[s:e]), text_u8[(p - 1).clamp(min=0, max=max(n - 1, 0))])
        if not bool((bwt_u8[s:e] == want).all()):
            raise AssertionError(f"bwt differs from text[sa - 1] in slots [{s}, {e})")
    return True


def verify_tables_on_device(counts, bwt_u8, c_i32, o_i32, N, sigma, rows=1 << 24):
    """C and O tables against the BWT they were built from; counts = symbol_counts(...)"""
    import torch
    want_c = torch.cumsum(counts, 0) - counts
    if not bool(((c_i32.long() & 0xFFFFFFFF) == want_c).all()):
        raise AssertionError("C table is not the exclusive prefix sum of the symbol counts")
    o = o_i32.view(N + 1, sigma)
    if not bool((o[0] == 0).all()):
        raise AssertionError("O row 0 is not zero")
    if not bool(((o[N].long() & 0xFFFFFFFF) == counts).all()):
        raise AssertionError("O row N is not the symbol counts")
    for s in range(0, N, rows):
        e = min(N, s + rows)
        d = o[s + 1: e + 1] - o[s: e]
        onehot = torch.nn.functional.one_hot(bwt_u8[s:e].long(), sigma).to(torch.int32)
        if not bool((d == onehot).all()):
            raise AssertionError(f"O rows [{s}, {e}] do not step by one-hot(bwt)")
    return True


def verify_build_on_device(text_u8, n, sigma, sa_i32, bwt_u8=None, c_i32=None, o_i32=None):
    """everything a build hands over; returns the list of what was checked"""
    done = []
    verify_sa_on_device(text_u8, sa_i32, n)
    done.append("sa: permutation, sa[0]=n, strictly increasing suffixes")
    if bwt_u8 is not None:
        verify_bwt_on_device(text_u8, sa_i32, bwt_u8, n)
        done.append("bwt[i]=text[sa[i]-1]")
    if c_i32 is not None and o_i32 is not None and bwt_u8 is not None:
        verify_tables_on_device(symbol_counts(text_u8, n, sigma), bwt_u8, c_i32, o_i32, n + 1, sigma)
        done.append("C=exclusive cumsum; O row 0=0, row N=counts, row steps=one-hot(bwt)")
    return done
