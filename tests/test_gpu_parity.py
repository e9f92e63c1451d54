"""Parity on hardware, through the C-ABI of the product library: the HIP path
against the oracle, the golden vectors and size-independent properties.
Everything here is bit-exact (integer / index work)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from oracle import pyoracle
from stralg_amd.synth import synth, repeat_families

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _loaded_product_lib():
    with open("/proc/self/maps") as f:
        return any("stralg_amd/libstralg_amd.so" in line for line in f)


def test_native_library_is_loaded(gpu_ctx):
    assert _loaded_product_lib()
    assert gpu_ctx.lib.sx_device_count() >= 1


# ---- kernel-level ------------------------------------------------------------------

def test_radix_sort_pairs(gpu_ctx):
    import torch
    rng = np.random.default_rng(1)
    for n, lo, hi in ((1, 0, 64), (5, 0, 64), (2048, 0, 64), (2049, 3, 61), (1_000_003, 0, 64), (300_000, 8, 24)):
        keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
        if n > 10:
            keys[::7] = keys[3]
        vals = np.arange(n, dtype=np.uint32)
        ka = torch.from_numpy(keys.view(np.int64)).cuda()
        va = torch.from_numpy(vals.view(np.int32)).cuda()
        kb, vb = torch.empty_like(ka), torch.empty_like(va)
        in_b = gpu_ctx.prim_sort_pairs_dev(ka, va, kb, vb, n, lo, hi)
        ks = (kb if in_b else ka).cpu().numpy().view(np.uint64)
        vs = (vb if in_b else va).cpu().numpy().view(np.uint32)
        mask = np.uint64(((1 << hi) - 1) ^ ((1 << lo) - 1)) if hi < 64 else np.uint64(~np.uint64((1 << lo) - 1))
        order = np.argsort(keys & mask, kind="stable")
        assert (vs == order).all(), (n, lo, hi)
        assert (ks == keys[order]).all(), (n, lo, hi)


def test_exclusive_sum(gpu_ctx):
    import torch
    rng = np.random.default_rng(2)
    for n in (1, 2047, 2048, 2049, 5_000_000):
        x = rng.integers(0, 800, size=n, dtype=np.uint32)
        d = torch.from_numpy(x.view(np.int32)).cuda()
        out, tot = torch.empty_like(d), torch.zeros(1, dtype=torch.int32, device="cuda")
        gpu_ctx.prim_exclusive_sum_dev(d, out, n, tot)
        ref = np.concatenate(([0], np.cumsum(x, dtype=np.uint64)[:-1])).astype(np.uint32)
        assert (out.cpu().numpy().view(np.uint32) == ref).all(), n
        assert int(tot.cpu().numpy().view(np.uint32)[0]) == int(x.sum()), n


def test_classify_against_model(gpu_ctx):
    import torch
    import model
    rng = np.random.default_rng(3)
    inputs = [rng.integers(1, 5, size=100_000, dtype=np.uint8), rng.integers(1, 256, size=70_000, dtype=np.uint8),
              np.concatenate([np.full(9000, 2, np.uint8), [1], np.full(5000, 3, np.uint8), [4]]).astype(np.uint8),
              np.full(20_000, 7, np.uint8)]
    # runs that span thousands of tiles and the 4096-tile chunks of the resolve pass (types carried across chunks)
    big = rng.integers(1, 5, size=50_000_000, dtype=np.uint8)
    big[5_000_000:25_000_000] = 2
    big[25_000_000] = 1            # the run ends on a smaller symbol: L-type all along
    big[30_000_000:47_000_000] = 3
    big[47_000_000] = 4            # ... on a larger one: S-type
    inputs += [big, np.full(40_000_000, 3, np.uint8)]
    for x in inputs:
        n = x.size
        T = np.concatenate((x, np.zeros(1, np.uint8)))
        is_s, lms = model.types_and_lms(T)
        d = torch.from_numpy(x).cuda()
        flags = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
        h = [torch.zeros(256, dtype=torch.int32, device="cuda") for _ in range(3)]
        gpu_ctx.prim_classify_dev(d, n, flags, *h)
        assert (flags.cpu().numpy().astype(bool) == lms).all()
        assert (h[0].cpu().numpy() == np.bincount(T, minlength=256)).all()
        assert (h[1].cpu().numpy() == np.bincount(T[~is_s], minlength=256)).all()
        assert (h[2].cpu().numpy() == np.bincount(T[lms], minlength=256)).all()


# ---- golden vectors and oracle ---------------------------------------------------------

def test_golden_suffix_arrays(gpu_routed, golden):
    for name, c in golden.items():
        if c["sigma"] == c["sym"].size + 1:
            continue
        assert (gpu_routed.sa_build(c["sym"], c["sigma"]) == c["sa"]).all(), name


def test_golden_tables(gpu_routed, golden):
    import stralg_amd
    checked = 0
    for name, c in golden.items():
        if "o" not in c:
            continue
        t = stralg_amd.build_complete_table(bytes(c["raw"]), True, gpu_routed)
        assert t.remap_table.alphabet_size == c["sigma"], name
        assert (t.sa.array == c["sa"]).all(), name
        assert (t.c_table == c["c"]).all(), name
        assert (t.o_table == c["o"]).all(), name      # all (N+1) rows, incl. the last (quirk 7)
        assert (t.ro_table == c["ro"]).all(), name
        checked += 1
    assert checked > 20


def test_edges_and_errors(gpu_ctx):
    from stralg_amd.api import StralgAmdError
    assert gpu_ctx.sa_build(np.zeros(0, np.uint8), 5).tolist() == [0]
    assert gpu_ctx.sa_build(np.array([3], np.uint8), 5).tolist() == [1, 0]
    assert gpu_ctx.sa_build(np.array([1, 1, 2, 3], np.uint8), 5).tolist() == [4, 0, 1, 2, 3]  # quirk 3
    assert gpu_ctx.sa_build(np.array([1, 2, 3, 4], np.uint8), 5).tolist() == [4, 0, 1, 2, 3]  # distinct symbols
    with pytest.raises(StralgAmdError):
        gpu_ctx.sa_build(np.array([1, 7, 2], np.uint8), 5)        # symbol >= alphabet_size
    with pytest.raises(StralgAmdError):
        gpu_ctx.sa_build(np.array([1, 0, 2], np.uint8), 5)        # interior sentinel
    x = np.array([1, 2, 1], np.uint8)
    assert (gpu_ctx.sa_build(x, 256) == oracle.sa_is_strict(x, 256)).all()    # loose alphabet


def test_random_against_oracle(gpu_routed):
    rng = np.random.default_rng(4)
    for sigma in (2, 3, 5, 21, 128, 256):
        for n in (7, 300, 4096, 4097, 70_001):
            x = rng.integers(1, sigma, size=n, dtype=np.uint8)
            assert (gpu_routed.sa_build(x, sigma) == oracle.sa_is_strict(x, sigma)).all(), (sigma, n)


def test_general_path_against_oracle(gpu_ctx):
    """SX_FLAG_FORCE_GENERAL_PATH: pieces + names + prefix doubling on inputs the fast path would take"""
    rng = np.random.default_rng(14)
    gpu_ctx.force_general_path(True)
    try:
        for sigma, n in ((5, 1 << 20), (256, 300_000), (3, 100_000), (21, 4097)):
            x = rng.integers(1, sigma, size=n, dtype=np.uint8)
            assert (gpu_ctx.sa_build(x, sigma) == oracle.sa_is_strict(x, sigma)).all(), (sigma, n)
            assert gpu_ctx.last_stats()["lms_path"] == 2
    finally:
        gpu_ctx.force_general_path(False)


def test_tie_refinement_rounds(gpu_ctx):
    """planted repeats longer than one prefix key: refinement rounds, then fallback"""
    base = synth(1 << 20, 5, 9)
    paths = []
    for L in (0, 30, 70, 200, 5000):
        x = base.copy()
        for k in range(1, 40):
            if L:
                x[k * 20000: k * 20000 + L] = x[100: 100 + L]
        assert (gpu_ctx.sa_build(x, 5) == oracle.sa_is(x, 5)).all(), L
        st = gpu_ctx.last_stats()
        paths.append((L, st["lms_path"], st["doubling_rounds"]))
    assert paths[0][1] == 1 and paths[1][1] == 1 and paths[-1][1] == 2, paths


def test_static_key_shapes(gpu_ctx):
    """the key kernel's static forms for DNA-like texts (the prefix lengths of 64 Mi ... 4 Gi symbol inputs),
    forced on a small text, and the run-time form next to them: same suffix array, same BWT tables"""
    try:
        for sigma in (5, 6):  # (A C G T: two-bit window codes; with N: three-bit ones)
            x = synth(1 << 20, sigma, 23)
            x[500_000:500_060] = x[1000:1060]
            want = oracle.sa_is(x, sigma)
            c_want, o_want = oracle.c_table(x, sigma), oracle.o_table(x, want, sigma)
            for C in (12, 13, 14, 15, 16, 17, 18, 19, 20):
                gpu_ctx.set_prefix_symbols(C)
                sa, c, o = gpu_ctx.build_tables(x, sigma)
                assert gpu_ctx.last_stats()["key_slots"] == C
                assert (sa == want).all() and (c == c_want).all() and (o == o_want).all(), (sigma, C)
    finally:
        gpu_ctx.set_prefix_symbols(0)


def test_hybrid_prefix_sort(gpu_ctx):
    """the prefix-key sort's two forms against the oracle: LSD passes over all key bits (mode 1) and HBM passes on the
    top 24 bits + sub-buckets ordered in LDS (mode 2; sx_localsort.hip), which also lists the ties; key shapes of
    64 Mi ... 4 Gi symbol texts forced on small ones; a sub-bucket no workgroup can hold falls back to LSD passes;
    wider radix digits (9, 10 bits) for the LSD form"""
    import torch
    rng = np.random.default_rng(5)
    try:
        for sigma, n in ((5, 1 << 22), (6, (1 << 20) + 77), (5, 4097)):
            x = synth(n, sigma, 31)
            if n > 100_000:
                x[500_000:500_080] = x[1000:1080]  # ties beyond the key: refinement rounds
                x[700_000:700_030] = x[1000:1030]
            want = oracle.sa_is(x, sigma)
            bw_want = oracle.bwt(x, want)
            for C in ((14, 16, 17, 18) if sigma == 5 else (13, 16)):
                # (modes: plain passes; HBM passes on the top 24 / 32 key bits, then sub-buckets in LDS.  Four letters, 16 ... 18
                #  symbols, a hybrid mode: the first HBM pass computes the keys itself -- bit 3 --, or, with the switch off, a key
                #  kernel does as in rounds 1 - 3)
                for mode, text_keys in ((1, True), (2, True), (2, False), (3, True), (3, False)):
                    gpu_ctx.set_prefix_symbols(C)
                    gpu_ctx.set_sort_mode(mode)
                    gpu_ctx.set_text_keys(text_keys)
                    xd = torch.from_numpy(x).cuda()
                    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
                    bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
                    gpu_ctx.sa_bwt_build_dev(xd, n, sigma, sa, bw)
                    st = gpu_ctx.last_stats()
                    assert (st["sort_local"] & 5) == (0, 1, 5)[mode - 1] and st["key_slots"] == C, (sigma, n, C, mode, st)
                    assert bool(st["sort_local"] & 8) == (text_keys and mode > 1 and sigma == 5 and 15 <= C <= 18), (sigma, n, C, mode, text_keys, st)
                    if sigma == 5 and mode < 3:  # four-letter texts: dense keys with the hybrid sort, base-5 keys with plain passes
                        assert st["key_bits"] == (2 * C + (4 if C < 16 else 5) if mode == 2 else int(np.ceil(C * np.log2(5)))), st
                    assert (sa.cpu().numpy().view(np.uint32) == want).all(), (sigma, n, C, mode)
                    assert (bw.cpu().numpy() == bw_want).all(), (sigma, n, C, mode)
        gpu_ctx.set_text_keys(True)
        # 40 copies of a 60-symbol piece: equal keys crowd a bin of the counting pass, that workgroup takes stable passes
        x = synth(1 << 20, 5, 33)
        for i in range(40):
            x[300 + 4000 * i:360 + 4000 * i] = x[100:160]
        gpu_ctx.set_prefix_symbols(17)
        gpu_ctx.set_sort_mode(2)
        want = oracle.sa_is(x, 5)
        # (round 5: the lean kernel leaves a workgroup with a crowded bin to the kernel of rounds 3 and 4, whose stable passes
        #  set bit 1; SX_FLAG_LOCAL_SORT_LEAN_OFF: that kernel for every workgroup)
        for lean in (True, False):
            gpu_ctx.set_local_sort_lean(lean)
            sa = gpu_ctx.sa_build(x, 5)
            assert gpu_ctx.last_stats()["sort_local"] & 7 == (1 if lean else 3), (lean, gpu_ctx.last_stats())
            assert (sa == want).all(), lean
        # more copies: bins of two and four 64-member pieces for the lean kernel's waves, some copies differing in the key's last
        # symbols; 700 copies: more than the waves take (SX_LS2_TEAM_MAX 512) -- that workgroup is left to the other kernel (bit 1)
        for copies, variants, bit1 in ((90, 0, 0), (200, 7, 0), (130, 64, 0), (500, 3, 0), (700, 0, 2), (700, 5, 2)):
            x = synth(1 << 20, 5, 35 + copies)
            piece = x[100:160].copy()
            for i in range(copies):
                x[300 + 1400 * i:360 + 1400 * i] = piece
                if variants and i % 3 == 0:
                    x[300 + 1400 * i + 14 + (i // 3) % 4] = 1 + (i // 3) % variants % 4
            want = oracle.sa_is(x, 5)
            for lean in (True, False):
                gpu_ctx.set_local_sort_lean(lean)
                sa = gpu_ctx.sa_build(x, 5)
                assert gpu_ctx.last_stats()["sort_local"] & 7 == (1 | bit1 if lean else 3), (copies, variants, lean, gpu_ctx.last_stats())
                assert (sa == want).all(), (copies, variants, lean)
        gpu_ctx.set_local_sort_lean(True)
        # 60 variants of the piece that differ in the key's last three symbols, 30 copies each: crowded bins with many values
        x = synth(1 << 20, 5, 34)
        rng2 = np.random.default_rng(5)
        variants = rng2.choice(64, size=60, replace=False)
        at = 300
        for v in variants.tolist():
            piece = x[100:160].copy()
            for _ in range(30):
                x[at:at + 60] = piece
                x[at + 14:at + 17] = np.array([1 + (v >> 4 & 3), 1 + (v >> 2 & 3), 1 + (v & 3)], np.uint8)
                at += 500
        sa = gpu_ctx.sa_build(x, 5)
        assert (sa == oracle.sa_is(x, 5)).all()
        # one 14-symbol prefix in front of tens of thousands of LMS suffixes
        unit = np.array([1, 3, 2, 4, 4, 2, 3, 1, 1, 3, 2, 4, 2, 1], np.uint8)
        x = np.concatenate([np.concatenate([unit, rng.integers(1, 5, size=6, dtype=np.uint8)]) for _ in range(20000)])
        gpu_ctx.set_prefix_symbols(17)
        gpu_ctx.set_sort_mode(2)
        want = oracle.sa_is(x, 5)
        gpu_ctx.set_long_subbuckets(False)  # rounds 1 - 3: a sub-bucket no workgroup can hold -> plain passes
        sa = gpu_ctx.sa_build(x, 5)
        assert gpu_ctx.last_stats()["sort_local"] == 0
        assert (sa == want).all()
        gpu_ctx.set_long_subbuckets(True)   # round 4: listed, and ordered by HBM passes of their own
        sa = gpu_ctx.sa_build(x, 5)
        st = gpu_ctx.last_stats()
        assert st["sort_local"] & 1 and st["long_subbuckets"] > 0, st
        assert (sa == want).all()
        # a repeat family and AT-rich prefixes in a random text (what a genome looks like to the sort): 20 000 copies of a
        # 60-symbol element with 5 % of their symbols redrawn, 30 % A and T -- sub-buckets of tens of thousands of pairs beside
        # the ordinary ones; listed, gathered, ordered by HBM passes of their own (sx_long_subbuckets), with both first passes
        n = 1 << 24
        x = (1 + np.searchsorted(np.array([0.3, 0.5, 0.7]), rng.random(n))).astype(np.uint8)
        elem = rng.integers(1, 5, size=60, dtype=np.uint8)
        for a in (rng.choice(n // 64 - 2, size=20000, replace=False) * 64).tolist():
            c = elem.copy()
            mut = rng.random(60) < 0.05
            c[mut] = rng.integers(1, 5, size=int(mut.sum()), dtype=np.uint8)
            x[a:a + 60] = c
        want = oracle.sa_is(x, 5)
        gpu_ctx.set_prefix_symbols(0)
        for mode, text_keys in ((0, True), (2, False)):
            gpu_ctx.set_sort_mode(mode)
            gpu_ctx.set_text_keys(text_keys)
            d = torch.from_numpy(x).cuda()
            sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
            bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
            gpu_ctx.sa_bwt_build_dev(d, n, 5, sa, bw)
            st = gpu_ctx.last_stats()
            assert st["lms_path"] == 1 and st["sort_local"] & 1 and st["long_subbuckets"] > 0, (mode, st)
            assert (sa.cpu().numpy().view(np.uint32) == want).all(), mode
            assert (bw.cpu().numpy() == oracle.bwt(x, want)).all(), mode
        gpu_ctx.set_text_keys(True)
        # wider digits
        gpu_ctx.set_sort_mode(1)
        x = synth(1 << 21, 5, 32)
        want = oracle.sa_is(x, 5)
        for bits in (9, 10):
            gpu_ctx.set_radix_digit_bits(bits)
            assert (gpu_ctx.sa_build(x, 5) == want).all(), bits
    finally:
        gpu_ctx.set_prefix_symbols(0)
        gpu_ctx.set_sort_mode(0)
        gpu_ctx.set_radix_digit_bits(0)
        gpu_ctx.set_text_keys(True)
        gpu_ctx.set_long_subbuckets(True)


def test_runs_of_many_lengths(gpu_ctx):
    """poly-A tracts and runs of every symbol with lengths 1 ... 60 scattered over a random text: the shrinking rounds of
    a bucket, where entries leave after different numbers of rounds (the tail kernel takes eight rounds at once from
    the entries' windows; the all-in-a-run jump never applies here), for 4, 5 and 7 symbols"""
    rng = np.random.default_rng(21)
    for sigma in (5, 6, 8):
        n = 3000000
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
        starts = rng.integers(0, n - 64, size=n // 40)
        lens = rng.integers(1, 61, size=starts.size)
        syms = rng.integers(1, sigma, size=starts.size)
        for a, l, c in zip(starts.tolist(), lens.tolist(), syms.tolist()):
            x[a:a + l] = c
        x[:50] = 1  # a run at the very start of the text (windows shorter than the batch)
        x[n - 45:] = sigma - 1
        assert (gpu_ctx.sa_build(x, sigma) == oracle.sa_is(x, sigma)).all(), sigma


def test_thousands_of_runs_closed_form(gpu_ctx):
    """thousands of runs of 1 ... 330 symbols of every symbol alive in a bucket at once (poly-A tracts, microsatellites): the
    tail kernel's closed form -- run lengths by one look at the text, every round's place from their histogram --, with
    runs beyond its 255-symbol look, runs at both ends of the text, 1, 3 and 8 entries a thread; SA and BWT"""
    import torch
    rng = np.random.default_rng(41)
    for sigma, n, k in ((5, 1 << 21, 900), (5, 1 << 22, 9000), (8, 1 << 21, 6000), (3, 1 << 20, 2500)):
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
        starts = rng.integers(0, n - 400, size=k)
        lens = rng.integers(1, 331, size=k)
        syms = rng.integers(1, sigma, size=k)
        for a, l, c in zip(starts.tolist(), lens.tolist(), syms.tolist()):
            x[a:a + l] = c
        x[:270] = 1
        x[n - 300:] = sigma - 1
        want = oracle.sa_is(x, sigma)
        d = torch.from_numpy(x).cuda()
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
        gpu_ctx.sa_bwt_build_dev(d, n, sigma, sa, bw)
        assert (sa.cpu().numpy().view(np.uint32) == want).all(), (sigma, n, k)
        assert (bw.cpu().numpy() == oracle.bwt(x, want)).all(), (sigma, n, k)


def test_short_records_direct_sort(gpu_ctx):
    """SX_FLAG_SMALL_DIRECT_MAX (on by default outside the tests): texts of at most 16 symbols and 2^24 suffixes are sorted
    directly, all suffixes by prefix key (lms_path 3) -- a third of the launches of classification + LMS sort + induced
    passes; SA, BWT, C and O against the oracle for 3 ... 16 symbols, with repeats (tie refinement), runs, texts too
    repetitive for it (they go on to the usual path), and the limit itself"""
    rng = np.random.default_rng(17)
    try:
        gpu_ctx.set_small_direct_max(-1)
        for sigma, n in ((5, 100), (5, 70000), (3, 3000), (4, 1 << 20), (8, 3_000_001), (16, 1 << 22), (12, 9000), (5, 64), (5, 1 << 24), (6, 5_000_000)):
            x = rng.integers(1, sigma, size=n, dtype=np.uint8)
            if n > 1000:
                x[200:260] = x[700:760]
                x[n - 100:n - 60] = x[300:340]
                x[n // 2:n // 2 + 300] = 1 + (sigma > 3)
            want = oracle.sa_is(x, sigma)
            sa, c, o = gpu_ctx.build_tables(x, sigma)
            st = gpu_ctx.last_stats()
            assert st["lms_path"] == 3, (sigma, n, st)
            assert (sa == want).all() and (c == oracle.c_table(x, sigma)).all() and (o == oracle.o_table(x, want, sigma)).all(), (sigma, n)
        # too repetitive for a prefix sort (a period of 2, then all equal): on to the usual path, same answer
        for x, sigma in ((np.tile(np.array([1, 2], np.uint8), 3000), 3), (np.full(5000, 1, np.uint8), 2)):
            assert (gpu_ctx.sa_build(x, sigma) == oracle.sa_is(x, sigma)).all()
            assert gpu_ctx.last_stats()["lms_path"] != 3
        # the limit counts suffixes (n + 1)
        x = rng.integers(1, 5, size=4999, dtype=np.uint8)
        for limit, path in ((5000, 3), (4999, 1), (0, 1)):
            gpu_ctx.set_small_direct_max(limit)
            assert (gpu_ctx.sa_build(x, 5) == oracle.sa_is(x, 5)).all()
            assert gpu_ctx.last_stats()["lms_path"] == path, limit
    finally:
        gpu_ctx.set_small_direct_max(0)


def test_both_induce_round_forms(gpu_ctx):
    """every round through the chained launch (look-back across up to thousands of tiles), every round
    through the three-launch form, and the default mix"""
    x = synth(1 << 22, 5, 77)
    want = oracle.sa_is(x, 5)
    y = synth(1 << 20, 200, 78)
    want_y = oracle.sa_is(y, 200)
    z = synth(1 << 22, 12, 79)  # 11 symbols: 32-bit windows, buckets of 380 k entries (the radix-pass form, 47 tiles a round)
    z[1_000_000:1_020_000] = 3  # and a run that keeps the tail kernel busy
    want_z = oracle.sa_is(z, 12)
    try:
        for thr in (0, 1 << 14, 1 << 30, -1):
            gpu_ctx.set_chain_max_entries(thr)
            gpu_ctx.set_no_direct_sort(True)
            assert (gpu_ctx.sa_build(x, 5) == want).all(), thr
            assert (gpu_ctx.sa_build(y, 200) == want_y).all(), thr
            assert (gpu_ctx.sa_build(z, 12) == want_z).all(), thr
    finally:
        gpu_ctx.set_chain_max_entries(-1)
        gpu_ctx.set_no_direct_sort(False)


def test_long_runs(gpu_ctx):
    """runs of one symbol far longer than a tile: the tail kernel's run jump (closed-form rounds)"""
    rng = np.random.default_rng(21)
    r = lambda n: rng.integers(1, 5, size=n, dtype=np.uint8)
    cases = {
        "all-equal-300k": (np.full(300_000, 1, np.uint8), 2),
        "n-run": (np.concatenate([r(50_000), np.full(400_000, 5, np.uint8), r(50_000)]), 6),
        "a-run": (np.concatenate([r(50_000), np.full(400_000, 1, np.uint8), r(50_000)]), 5),
        "three-runs": (np.concatenate([np.full(150_000, 3, np.uint8), [1], np.full(170_001, 3, np.uint8), [4],
                                       np.full(160_000, 3, np.uint8), [2], r(5000)]).astype(np.uint8), 5),
        "forty-runs": (np.concatenate([np.concatenate([r(500), np.full(20_000 + 37 * i, 4, np.uint8)])
                                       for i in range(40)]), 5),
        "run-at-both-ends": (np.concatenate([np.full(100_000, 2, np.uint8), r(30_000), np.full(100_000, 2, np.uint8)]), 5),
        "bytes-run": (np.concatenate([rng.integers(1, 200, size=20_000, dtype=np.uint8), np.full(300_000, 77, np.uint8),
                                      rng.integers(1, 200, size=20_000, dtype=np.uint8)]), 200),
    }
    for name, (x, sigma) in cases.items():
        assert (gpu_ctx.sa_build(x, sigma) == oracle.sa_is_strict(x, sigma)).all(), name


def test_very_long_runs(gpu_ctx):
    """runs that outlast the tail kernel's steps (more than 256 x 4096 symbols): the device-wide run jump, in the L
    pass (run followed by a smaller symbol), in the S pass (by a larger one), at the start of the text, as the whole
    text, and two runs of the same symbol whose jumps alternate; BWT and tables through the same build"""
    rng = np.random.default_rng(31)
    r = lambda n, hi=5: rng.integers(1, hi, size=n, dtype=np.uint8)
    M = 1 << 20
    cases = {
        "all-equal-5M": (np.full(5 * M, 1, np.uint8), 2),
        "l-type-run": (np.concatenate([r(M), np.full(3 * M, 4, np.uint8), [1], r(M)]).astype(np.uint8), 5),
        "s-type-run": (np.concatenate([r(M), np.full(3 * M, 2, np.uint8), [4], r(M)]).astype(np.uint8), 5),
        "run-at-start": (np.concatenate([np.full(2 * M + 5, 3, np.uint8), r(M)]), 5),
        "n-runs": (np.concatenate([r(M), np.full(4 * M, 5, np.uint8), r(M), np.full(2 * M + 77, 5, np.uint8), r(1000)]), 6),
        "bytes-run": (np.concatenate([r(100_000, 200), np.full(2 * M, 77, np.uint8), r(100_000, 200)]), 200),
    }
    for name, (x, sigma) in cases.items():
        want = oracle.sa_is_strict(x, sigma)
        assert (gpu_ctx.sa_build(x, sigma) == want).all(), name
        if sigma <= 8:
            sa, c, o = gpu_ctx.build_tables(x, sigma)
            assert (sa == want).all() and (c == oracle.c_table(x, sigma)).all(), name
            assert (o == oracle.o_table(x, want, sigma)).all(), name


def test_structured_against_oracle(gpu_routed):
    rng = np.random.default_rng(5)
    cases = {
        "all-equal": np.full(3000, 1, np.uint8),
        "runs": np.repeat(rng.integers(1, 5, size=3000, dtype=np.uint8), 23),
        "tile-runs": np.concatenate([rng.integers(1, 5, size=4090, dtype=np.uint8), np.full(9000, 3, np.uint8),
                                     rng.integers(1, 5, size=4000, dtype=np.uint8)]),
        "periodic": np.tile(rng.integers(1, 5, size=97, dtype=np.uint8), 3000),
        "two-long-lms": np.concatenate([np.full(20000, 1, np.uint8), [2], np.full(20000, 1, np.uint8), [2]]).astype(np.uint8),
        "long-pieces": np.tile(np.concatenate([np.full(100, 2, np.uint8), [1]]).astype(np.uint8), 500),
    }
    for name, x in cases.items():
        sigma = int(x.max()) + 1
        assert (gpu_routed.sa_build(x, sigma) == oracle.sa_is_strict(x, sigma)).all(), name


@pytest.mark.parametrize("sigma,n", [(5, 1 << 24), (256, 1 << 22)])
def test_benchmark_shaped_against_oracle(gpu_ctx, sigma, n):
    x = synth(n, sigma, 42)
    got = gpu_ctx.sa_build(x, sigma)
    want = oracle.sa_is(x, sigma)
    assert (got == want).all()
    if sigma <= 128:
        c, o = gpu_ctx.bwt_tables(x[: 1 << 20], oracle.sa_is(x[: 1 << 20], sigma), sigma)
        assert (c == oracle.c_table(x[: 1 << 20], sigma)).all()
        assert (o == oracle.o_table(x[: 1 << 20], oracle.sa_is(x[: 1 << 20], sigma), sigma)).all()


def test_fused_sa_bwt_tables(gpu_routed):
    """sx_sa_bwt_build_dev + sx_bwt_tables_from_bwt_dev (what bench.py times) and sx_build_tables"""
    import torch
    rng = np.random.default_rng(16)
    for sigma, n in ((5, 1 << 20), (5, 1023), (4, 77), (3, 9000), (6, 300_001), (7, 5000), (8, 70_000), (17, 50_000), (128, 20_000),
                     (200, 30_000)):
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
        want = oracle.sa_is(x, sigma)
        d = torch.from_numpy(x).cuda()
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
        gpu_routed.sa_bwt_build_dev(d, n, sigma, sa, bw)
        assert (sa.cpu().numpy().view(np.uint32) == want).all(), (sigma, n)
        assert (bw.cpu().numpy() == oracle.bwt(x, want)).all(), (sigma, n)
        if sigma <= 128:
            c = torch.zeros(sigma, dtype=torch.int32, device="cuda")
            o = torch.empty((n + 2) * sigma, dtype=torch.int32, device="cuda")
            gpu_routed.bwt_tables_from_bwt_dev(bw, n + 1, sigma, c, o)
            assert (c.cpu().numpy().view(np.uint32) == oracle.c_table(x, sigma)).all(), (sigma, n)
            assert (o.cpu().numpy().view(np.uint32).reshape(n + 2, sigma) == oracle.o_table(x, want, sigma)).all()
            sa2, c2, o2 = gpu_routed.build_tables(x, sigma)
            assert (sa2 == want).all() and (o2 == oracle.o_table(x, want, sigma)).all(), (sigma, n)


def test_unaligned_bwt_buffer(gpu_routed):
    """the caller's BWT buffer is the induction's symbol-byte array: the counting launches read it in aligned
    16-byte pieces, or byte by byte when the caller's pointer is not 16-byte aligned; both round forms"""
    import torch
    rng = np.random.default_rng(26)
    try:
        for sigma, n in ((5, 300_007), (3, 70_001), (21, 200_003)):
            x = rng.integers(1, sigma, size=n, dtype=np.uint8)
            x[1000:3000] = x[5000]  # a run: many rounds of one bucket
            want = oracle.sa_is(x, sigma)
            d = torch.from_numpy(x).cuda()
            for chain_max in (2048, 524288):
                gpu_routed.set_chain_max_entries(chain_max)
                for off in (0, 1, 7):
                    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
                    buf = torch.zeros(n + 1 + 16, dtype=torch.uint8, device="cuda")
                    bw = buf[off: off + n + 1]
                    gpu_routed.sa_bwt_build_dev(d, n, sigma, sa, bw)
                    assert (sa.cpu().numpy().view(np.uint32) == want).all(), (sigma, chain_max, off)
                    assert (bw.cpu().numpy() == oracle.bwt(x, want)).all(), (sigma, chain_max, off)
                    assert int(buf[off + n + 1:].sum()) == 0 and int(buf[:off].sum()) == 0  # nothing outside
    finally:
        gpu_routed.set_chain_max_entries(-1)


def test_wide_alphabet_tables(gpu_ctx):
    rng = np.random.default_rng(6)
    # (32, 33, 64, 65, 128: the edges of the wide kernel's tile classes, where its LDS rows are largest)
    for sigma, n in ((9, 5000), (21, 70_000), (128, 30_000), (32, 16_382), (33, 9000), (64, 65_538), (65, 9000), (127, 9000)):
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
        sa = oracle.sa_is(x, sigma)
        c, o = gpu_ctx.bwt_tables(x, sa, sigma)
        assert (c == oracle.c_table(x, sigma)).all() and (o == oracle.o_table(x, sa, sigma)).all(), (sigma, n)


# ---- the reference-named C entry points (include/stralg_compat.h) ------------------------

def test_reference_named_c_api(gpu_ctx, golden):
    class SA(C.Structure):
        _fields_ = [("string", C.POINTER(C.c_uint8)), ("length", C.c_uint32), ("array", C.POINTER(C.c_uint32)),
                    ("inverse", C.c_void_p), ("lcp", C.c_void_p)]

    class RT(C.Structure):
        _fields_ = [("alphabet_size", C.c_uint32), ("table", C.c_byte * 256), ("rev_table", C.c_byte * 128)]

    class BT(C.Structure):
        _fields_ = [("remap_table", C.POINTER(RT)), ("sa", C.POINTER(SA)), ("c_table", C.POINTER(C.c_uint32)),
                    ("o_table", C.POINTER(C.c_uint32)), ("o_indices", C.POINTER(C.POINTER(C.c_uint32))),
                    ("ro_table", C.POINTER(C.c_uint32)), ("ro_indices", C.POINTER(C.POINTER(C.c_uint32)))]

    lib = gpu_ctx.lib
    lib.build_complete_table.argtypes = [C.c_char_p, C.c_bool]
    lib.build_complete_table.restype = C.POINTER(BT)
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    for name in ("ref/mississippi", "ref/serialise", "ref/fasta2", "struct/periodic"):
        c = golden[name]
        t = lib.build_complete_table(bytes(c["raw"]), True).contents
        N, sigma = t.sa.contents.length, t.remap_table.contents.alphabet_size
        assert sigma == c["sigma"] and N == c["sa"].size
        assert (np.ctypeslib.as_array(t.sa.contents.array, shape=(N,)) == c["sa"]).all()
        assert (np.ctypeslib.as_array(t.c_table, shape=(sigma,)) == c["c"]).all()
        assert (np.ctypeslib.as_array(t.o_table, shape=(N + 1, sigma)) == c["o"]).all()
        assert (np.ctypeslib.as_array(t.ro_table, shape=(N + 1, sigma)) == c["ro"]).all()
        # the O(a, i) macro of bwt.h:49 goes through the row pointers
        for i in (0, N // 2, N):
            assert [t.o_indices[i][a] for a in range(sigma)] == c["o"][i].tolist()
        lib.completely_free_bwt_table(t)
    # the three constructors share one result (match_test.c:479,517,539)
    lib.sa_is_construction.argtypes = [C.c_char_p, C.c_uint32]
    lib.sa_is_construction.restype = C.POINTER(SA)
    lib.skew_sa_construction.argtypes = [C.c_char_p]
    lib.skew_sa_construction.restype = C.POINTER(SA)
    lib.free_suffix_array.argtypes = [C.POINTER(SA)]
    lib.free_suffix_array.restype = None
    c = golden["ref/ababacabac"]
    buf = C.create_string_buffer(bytes(c["sym"]))
    a = lib.sa_is_construction(buf, c["sigma"])
    b = lib.skew_sa_construction(buf)
    assert np.ctypeslib.as_array(a.contents.array, shape=(11,)).tolist() == [10, 0, 6, 2, 8, 4, 1, 7, 3, 9, 5]
    assert np.ctypeslib.as_array(b.contents.array, shape=(11,)).tolist() == [10, 0, 6, 2, 8, 4, 1, 7, 3, 9, 5]
    lib.free_suffix_array(a)
    lib.free_suffix_array(b)
    # sa_is_mem_construction (sa_is_mem.c:471-494) and skew_sa_construction (skew.c:388-395) against the oracle, not
    # against each other: a remapped text for the first, raw bytes (sigma = 256, skew.c:375) for the second
    lib.sa_is_mem_construction.argtypes = [C.c_char_p, C.c_uint32]
    lib.sa_is_mem_construction.restype = C.POINTER(SA)
    for name in ("ref/mississippi", "ref/modest-proposal", "struct/fibonacci", "rand/s5/n65536"):
        if name not in golden:
            continue
        c = golden[name]
        buf = C.create_string_buffer(bytes(c["sym"]))
        a = lib.sa_is_mem_construction(buf, c["sigma"])
        assert (np.ctypeslib.as_array(a.contents.array, shape=(c["sa"].size,)) == c["sa"]).all(), name
        lib.free_suffix_array(a)
    x = synth(300_000, 5, 8)
    buf = C.create_string_buffer(bytes(x))
    want = oracle.sa_is(x, 5)
    a = lib.sa_is_mem_construction(buf, 5)
    assert (np.ctypeslib.as_array(a.contents.array, shape=(x.size + 1,)) == want).all()
    lib.free_suffix_array(a)
    raw = np.frombuffer(b"the modest proposal of a text with blanks, UPPER case and 8-bit bytes \xe9\xff " * 500, np.uint8)
    buf = C.create_string_buffer(bytes(raw))
    b = lib.skew_sa_construction(buf)
    assert (np.ctypeslib.as_array(b.contents.array, shape=(raw.size + 1,)) == oracle.sa_is(raw, 256)).all()
    lib.free_suffix_array(b)


def test_next_rows_inverse_lcp_search(gpu_ctx, golden):
    """SURVEY 8f rows 3 and 4 on the device: compute_inverse / compute_lcp and batched exact BWT search,
    against the reference's vectors and the oracle"""
    import torch
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_next.npz"))
    for name in sorted({k.rsplit("/", 1)[0] for k in z.files}):
        c = golden[name]
        if name + "/lcp" in z.files:
            inv, lcp = gpu_ctx.inverse_lcp(c["sym"], c["sa"])
            assert (inv == z[name + "/inverse"]).all() and (lcp == z[name + "/lcp"]).all(), name
        if name + "/lr" in z.files:
            pats, offs, lr = z[name + "/patterns"], z[name + "/offsets"], z[name + "/lr"]
            dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).cuda()
            l = torch.zeros(lr.shape[0], dtype=torch.int32, device="cuda")
            r = torch.zeros_like(l)
            gpu_ctx.bwt_exact_search_dev(dev(c["c"], np.int32), dev(c["o"], np.int32), c["sa"].size, c["sigma"],
                                         dev(pats, np.uint8), dev(offs, np.int32), lr.shape[0], l, r)
            assert (l.cpu().numpy().view(np.uint32) == lr[:, 0]).all(), name
            assert (r.cpu().numpy().view(np.uint32) == lr[:, 1]).all(), name
    # larger, against the oracle
    x = synth(1 << 22, 5, 5)
    sa = gpu_ctx.sa_build(x, 5)
    inv, lcp = gpu_ctx.inverse_lcp(x, sa)
    assert (inv == oracle.inverse(sa)).all() and (lcp == oracle.lcp(x, sa)).all()
    y = np.tile(synth(3000, 5, 6), 40)  # repetitive: long common prefixes
    sa = gpu_ctx.sa_build(y, 5)
    assert (gpu_ctx.inverse_lcp(y, sa)[1] == oracle.lcp(y, sa)).all()
    # from 2^23 entries on the inverse takes three passes and (round 5) the LCP goes through Phi: phi[sa[j]] = sa[j - 1] by a
    # permutation scatter, PLCP in text order, lcp[inv[i]] = plcp[i] by a second scatter -- random DNA, a text with long repeats
    # (the samples' invariant over many chunks), and a length that is no multiple of anything
    for x in (synth((1 << 24) + 12345, 5, 7), np.tile(synth(70_001, 5, 8), 130)):
        sa = gpu_ctx.sa_build(x, 5)
        inv, lcp = gpu_ctx.inverse_lcp(x, sa)
        assert (inv == oracle.inverse(sa)).all() and (lcp == oracle.lcp(x, sa)).all()


def test_c_batch_farm(gpu_ctx, golden):
    """stralg_amd_build_tables_batch: independent records, one host thread (and context) per listed device;
    here two threads share GPU 0, which also exercises concurrent contexts"""
    class SA(C.Structure):
        _fields_ = [("string", C.POINTER(C.c_uint8)), ("length", C.c_uint32), ("array", C.POINTER(C.c_uint32)),
                    ("inverse", C.c_void_p), ("lcp", C.c_void_p)]

    class BT(C.Structure):
        _fields_ = [("remap_table", C.c_void_p), ("sa", C.POINTER(SA)), ("c_table", C.POINTER(C.c_uint32)),
                    ("o_table", C.POINTER(C.c_uint32)), ("o_indices", C.c_void_p),
                    ("ro_table", C.POINTER(C.c_uint32)), ("ro_indices", C.c_void_p)]

    lib = gpu_ctx.lib
    names = ["ref/fasta0", "ref/fasta1", "ref/fasta2", "ref/fasta3", "ref/fasta4", "ref/mississippi", "struct/periodic"]
    raws = [bytes(golden[n]["raw"]) for n in names]
    arr = (C.c_char_p * len(raws))(*raws)
    out = (C.POINTER(BT) * len(raws))()
    devs = (C.c_int * 2)(0, 0)
    lib.stralg_amd_build_tables_batch.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_bool, C.POINTER(C.c_int), C.c_int,
                                                  C.POINTER(C.POINTER(BT))]
    lib.stralg_amd_build_tables_batch.restype = C.c_int
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    assert lib.stralg_amd_build_tables_batch(arr, len(raws), True, devs, 2, out) == 0
    for n, t in zip(names, out):
        c = golden[n]
        N, sigma = t.contents.sa.contents.length, c["sigma"]
        assert (np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(N,)) == c["sa"]).all(), n
        assert (np.ctypeslib.as_array(t.contents.o_table, shape=(N + 1, sigma)) == c["o"]).all(), n
        assert (np.ctypeslib.as_array(t.contents.ro_table, shape=(N + 1, sigma)) == c["ro"]).all(), n
        lib.completely_free_bwt_table(t)


def test_c_batch_farm_unequal_records(gpu_ctx):
    """records of very unequal length (3 M ... 20 k symbols) over three lanes: dealt longest first to the least loaded
    lane (stralg_amd_lpt_assign), every worker pinned to its GPU's NUMA node; results against the oracle"""
    class SA(C.Structure):
        _fields_ = [("string", C.POINTER(C.c_uint8)), ("length", C.c_uint32), ("array", C.POINTER(C.c_uint32)),
                    ("inverse", C.c_void_p), ("lcp", C.c_void_p)]

    class BT(C.Structure):
        _fields_ = [("remap_table", C.c_void_p), ("sa", C.POINTER(SA)), ("c_table", C.POINTER(C.c_uint32)),
                    ("o_table", C.POINTER(C.c_uint32)), ("o_indices", C.c_void_p),
                    ("ro_table", C.POINTER(C.c_uint32)), ("ro_indices", C.c_void_p)]

    lib = gpu_ctx.lib
    lengths = [3_000_000, 20_000, 1_500_000, 700_000, 50_000, 1_400_000, 123_457]
    letters = np.frombuffer(b"\0ACGT", dtype=np.uint8)
    syms = [synth(n, 5, 500 + i) for i, n in enumerate(lengths)]
    raws = [letters[x].tobytes() for x in syms]
    arr = (C.c_char_p * len(raws))(*raws)
    out = (C.POINTER(BT) * len(raws))()
    devs = (C.c_int * 3)(0, 0, 0)
    lib.stralg_amd_build_tables_batch.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_bool, C.POINTER(C.c_int), C.c_int,
                                                  C.POINTER(C.POINTER(BT))]
    lib.stralg_amd_build_tables_batch.restype = C.c_int
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    assert lib.stralg_amd_build_tables_batch(arr, len(raws), False, devs, 3, out) == 0
    for x, t in zip(syms, out):
        N = t.contents.sa.contents.length
        assert N == x.size + 1
        want = oracle.sa_is(x, 5)
        assert (np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(N,)) == want).all(), N
        o = np.ctypeslib.as_array(t.contents.o_table, shape=(N + 1, 5))
        assert (o[N] == np.bincount(np.concatenate((x, [0])), minlength=5)).all()
        assert not t.contents.ro_table
        lib.completely_free_bwt_table(t)
    # the NUMA binding: the node the library reports for GPU 0 is the node it binds a thread to
    lib.stralg_amd_bind_thread_to_device.argtypes = [C.c_int]
    import threading
    got = []
    th = threading.Thread(target=lambda: got.append((lib.stralg_amd_bind_thread_to_device(0), os.sched_getaffinity(0))))
    th.start()
    th.join()
    node = lib.sx_device_numa_node(0)
    assert got[0][0] == node or got[0][0] == -1
    if node >= 0 and got[0][0] == node:
        cpus = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
        want_set = set()
        for part in cpus.split(","):
            a, _, b = part.partition("-")
            want_set |= set(range(int(a), int(b or a) + 1))
        assert got[0][1] == want_set & os.sched_getaffinity(0) or got[0][1] <= want_set


def test_c_batch_farm_many_short_records(gpu_ctx):
    """40 records of 3 k ... 300 k symbols on one listed device: the farm runs four workers (contexts, streams, host
    threads) on it; every record against the oracle, with one worker ($STRALG_AMD_FARM_WORKERS=1) and with the default"""
    class SA(C.Structure):
        _fields_ = [("string", C.POINTER(C.c_uint8)), ("length", C.c_uint32), ("array", C.POINTER(C.c_uint32)),
                    ("inverse", C.c_void_p), ("lcp", C.c_void_p)]

    class BT(C.Structure):
        _fields_ = [("remap_table", C.c_void_p), ("sa", C.POINTER(SA)), ("c_table", C.POINTER(C.c_uint32)),
                    ("o_table", C.POINTER(C.c_uint32)), ("o_indices", C.c_void_p),
                    ("ro_table", C.POINTER(C.c_uint32)), ("ro_indices", C.c_void_p)]

    lib = gpu_ctx.lib
    rng = np.random.default_rng(77)
    lengths = [int(v) for v in rng.integers(3000, 300000, size=40)]
    letters = np.frombuffer(b"\0ACGT", dtype=np.uint8)
    syms = [synth(n, 5, 900 + i) for i, n in enumerate(lengths)]
    raws = [letters[x].tobytes() for x in syms]
    want = [oracle.sa_is(x, 5) for x in syms]
    arr = (C.c_char_p * len(raws))(*raws)
    devs = (C.c_int * 1)(0)
    lib.stralg_amd_build_tables_batch.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_bool, C.POINTER(C.c_int), C.c_int,
                                                  C.POINTER(C.POINTER(BT))]
    lib.stralg_amd_build_tables_batch.restype = C.c_int
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    lib.stralg_amd_farm_workers_per_device.argtypes = [C.POINTER(C.c_size_t), C.c_size_t, C.c_int]
    la = (C.c_size_t * len(lengths))(*lengths)
    old = os.environ.pop("STRALG_AMD_FARM_WORKERS", None)
    try:
        assert lib.stralg_amd_farm_workers_per_device(la, len(lengths), 1) == 4
        for setting in (None, "1"):
            if setting:
                os.environ["STRALG_AMD_FARM_WORKERS"] = setting
            out = (C.POINTER(BT) * len(raws))()
            assert lib.stralg_amd_build_tables_batch(arr, len(raws), False, devs, 1, out) == 0
            for x, w, t in zip(syms, want, out):
                N = t.contents.sa.contents.length
                assert N == x.size + 1
                assert (np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(N,)) == w).all(), (setting, N)
                o = np.ctypeslib.as_array(t.contents.o_table, shape=(N + 1, 5))
                assert (o == oracle.o_table(x, w, 5)).all(), (setting, N)
                lib.completely_free_bwt_table(t)
    finally:
        os.environ.pop("STRALG_AMD_FARM_WORKERS", None)
        if old is not None:
            os.environ["STRALG_AMD_FARM_WORKERS"] = old


def test_bench_refuses_two_ranks_on_one_gpu():
    """Two ranks of `bench.py --gpus 2` that end up on ONE device (both with LOCAL_RANK 0 here: what a launcher that hands
    out the same local rank, or a masked HIP_VISIBLE_DEVICES, does to a driver): N ranks must mean N GPUs, so rank 0 prints
    ONE JSON line that carries `error` and every rank exits non-zero -- unless the builder's
    rehearsal switch STRALG_BENCH_SHARE_GPU=1 says that sharing is meant."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = {k: v for k, v in os.environ.items() if k != "STRALG_BENCH_SHARE_GPU"}
    procs = []
    for r in range(2):
        env = dict(base, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   STRALG_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "200000", "--steps", "1",
                                       "--warmup", "0", "--no-e2e", "--no-cpu", "--no-egress", "--no-ro"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode != 0 for p in procs), [(p.returncode, o[1][-500:]) for p, o in zip(procs, outs)]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    doc = json.loads(lines[0])
    assert "error" in doc and doc["value"] is None and doc["n_gpus"] == 2 and "more than one" in doc["error"], doc


def test_c_harness_runs(tmp_path):
    """a plain C caller of the reference-named API (restated performance/suffix_array_construction.c)"""
    import subprocess
    exe = tmp_path / "harness"
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-D_GNU_SOURCE", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "sa_construction_harness.c"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "stralg_amd"), "-lstralg_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "stralg_amd")])
    for args in (["-n", "65536", "-k", "dna", "-r", "1", "-t"], ["-n", "49000", "-k", "equal", "-r", "1"],
                 ["-n", "200000", "-k", "ascii", "-r", "1", "-t"], ["-n", "0", "-r", "1", "-t"]):
        out = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (args, out.stderr[-500:])
        assert "SA-IS " in out.stdout and "Skew " in out.stdout


# ---- BASELINE.json's full sizes: size-independent properties -------------------------------

def _verify_sa_on_device(text_u8, sa_i32, n):
    """permutation + strictly increasing suffixes, O(n) on the GPU with torch (stralg_amd/verify.py)."""
    from stralg_amd.verify import verify_sa_on_device
    verify_sa_on_device(text_u8, sa_i32, n)


def test_beyond_31_bits(gpu_ctx):
    """n = 2^31 + 12345: positions no longer fit a signed 32-bit integer (the API promises n <= 2^32 - 2)"""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs ~150 GiB of device memory")
    n = (1 << 31) + 12345
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    gpu_ctx.synth_dev(text, n, 5, 99)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    gpu_ctx.sa_build_dev(text, n, 5, sa)
    gpu_ctx.trim()  # the verification below needs the memory
    _verify_sa_on_device(text, sa, n)


# BASELINE.json configs[1..3] at their sizes: 256 MiB DNA, 1 GiB DNA (SA + BWT + C/O), 1 GiB sigma = 256.
def test_induced_passes_of_wide_alphabets_at_size(gpu_ctx):
    """LMS sort + induced-sort passes where the direct sort would apply, at sizes whose buckets take the radix-pass
    round form with its one-launch offsets (255 symbols: 2 M entries a bucket) and its three-launch offsets (11
    symbols: 12 M entries a bucket, 1500 tiles a round): suffix array, BWT and tables checked on the device"""
    import torch
    from stralg_amd import verify
    try:
        gpu_ctx.set_no_direct_sort(True)
        for log2n, sigma in ((27, 12), (28, 256)):
            n = 1 << log2n
            N = n + 1
            text = torch.empty(n, dtype=torch.uint8, device="cuda")
            gpu_ctx.synth_dev(text, n, sigma, 5 + log2n)
            sa = torch.empty(N, dtype=torch.int32, device="cuda")
            bwt = torch.empty(N, dtype=torch.uint8, device="cuda")
            gpu_ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
            assert gpu_ctx.last_stats()["lms_path"] == 1
            tables = sigma <= 128
            c = torch.zeros(sigma, dtype=torch.int32, device="cuda") if tables else None
            o = torch.empty((N + 1) * sigma, dtype=torch.int32, device="cuda") if tables else None
            if tables:
                gpu_ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c, o)
            gpu_ctx.trim()
            verify.verify_build_on_device(text, n, sigma, sa, bwt, c, o)
            del text, sa, bwt, c, o
            torch.cuda.empty_cache()
    finally:
        gpu_ctx.set_no_direct_sort(False)


def _reference_pins(log2n, sigma, sa, bw=None, c=None, o=None):
    """BASELINE.json configs[1]-[3] against the UNMODIFIED reference's own output at size
    (tests/golden/golden_big.npz, written by tests/golden/make_golden_big.py from oracle/_ref's sa_is_mem_construction,
    sa_is_mem.c:471-494, on the same splitmix64 text): SHA-256 of the whole suffix array (and of every 2^26-entry chunk,
    so that a mismatch is located), every 2^20-th entry, the BWT's SHA-256, and the symbol counts = C table and last O row."""
    import hashlib
    import torch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_big.npz"))
    key = f"n{log2n}/s{sigma}"
    N = (1 << log2n) + 1
    assert int(z[key + "/seed"][0]) == 42 and sa.numel() == N
    want_sampled = z[key + "/sa_sampled"]
    got_sampled = torch.cat([sa[:: 1 << 20], sa[-1:]]).cpu().numpy().view(np.uint32)
    assert (got_sampled == want_sampled).all(), f"{key}: sampled suffix-array entries differ from the reference's"
    h, chunk = hashlib.sha256(), 1 << 26
    for k, s0 in enumerate(range(0, N, chunk)):
        b = sa[s0:s0 + chunk].cpu().numpy().tobytes()  # (int32 storage: the same little-endian bytes as the reference's u32)
        assert hashlib.sha256(b).digest() == bytes(z[key + "/sa_chunk_sha256"][k]), f"{key}: suffix-array chunk {k} differs"
        h.update(b)
    assert h.digest() == bytes(z[key + "/sa_sha256"]), f"{key}: SHA-256 of the suffix array differs from the reference's"
    if bw is not None:
        hb = hashlib.sha256()
        for s0 in range(0, N, 1 << 28):
            hb.update(bw[s0:s0 + (1 << 28)].cpu().numpy().tobytes())
        assert hb.digest() == bytes(z[key + "/bwt_sha256"]), f"{key}: SHA-256 of the BWT differs from the reference's"
    if c is not None:
        counts = z[key + "/counts"].astype(np.int64)
        assert (c.cpu().numpy().astype(np.int64) == np.concatenate([[0], np.cumsum(counts)[:-1]])).all()  # bwt.c:22-31
        assert (o[N * sigma:(N + 1) * sigma].cpu().numpy().astype(np.int64) == counts).all()  # the row behind the last position
    return key


@pytest.mark.parametrize("log2n,sigma", [(28, 5), (30, 5), (28, 256), (30, 256)])
def test_full_size_properties(gpu_ctx, log2n, sigma):
    import torch
    from stralg_amd import verify
    n = 1 << log2n
    N = n + 1
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    gpu_ctx.synth_dev(text, n, sigma, 42)
    head = text[: 1 << 16].cpu().numpy()
    assert (head == synth(1 << 16, sigma, 42)).all()
    sa = torch.empty(N, dtype=torch.int32, device="cuda")
    if sigma > 128:
        # suffix array only (BWT tables are defined for sigma <= 128, stralg/remap.h:14-18)
        gpu_ctx.sa_build_dev(text, n, sigma, sa)
        assert gpu_ctx.last_stats()["lms_path"] == 3  # random bytes take the direct sort of all suffixes
        verify.verify_sa_on_device(text, sa, n)
        # ... and the LMS sort + induced-sort passes on the same input (BASELINE.json configs[3]'s "wide-alphabet
        # LDS-histogram path", at its full size too) give the same array
        sa2 = torch.empty_like(sa)
        gpu_ctx.set_no_direct_sort(True)
        try:
            gpu_ctx.sa_build_dev(text, n, sigma, sa2)
            st = gpu_ctx.last_stats()
            assert st["lms_path"] in (1, 2) and st["induce_rounds"] > 500
        finally:
            gpu_ctx.set_no_direct_sort(False)
        assert bool((sa2 == sa).all())
        _reference_pins(log2n, sigma, sa)  # ... and it is the reference's own array (SHA-256, sampled entries)
        return
    # the fused calls bench.py times: suffix array + BWT from the induced-sort passes, then C and O from that BWT
    c = torch.zeros(sigma, dtype=torch.int32, device="cuda")
    o = torch.empty((N + 1) * sigma, dtype=torch.int32, device="cuda")
    bw = torch.empty(N, dtype=torch.uint8, device="cuda")
    gpu_ctx.sa_bwt_build_dev(text, n, sigma, sa, bw)
    st = gpu_ctx.last_stats()
    # what was timed is what is checked: four letters at these sizes take the hybrid sort (bit 0) whose first HBM pass lists
    # the LMS suffixes and computes their keys itself (bit 3: radix_scatter_lms_kernel), with the one key symbol more
    assert st["lms_path"] == 1 and st["sort_local"] & 9 == 9 and st["key_slots"] == (17 if log2n == 28 else 18), st
    gpu_ctx.bwt_tables_from_bwt_dev(bw, N, sigma, c, o)
    gpu_ctx.trim()  # the checks need the memory more than the library's cached workspace does
    done = verify.verify_build_on_device(text, n, sigma, sa, bw, c, o)
    assert len(done) == 3
    _reference_pins(log2n, sigma, sa, bw, c, o)  # the reference's own SA / BWT / counts at this size, by SHA-256
    if log2n <= 28:
        # the unfused entry points (sa_is_construction, then init_bwt_table's gather) must hand over the same
        sa2, bw2 = torch.empty_like(sa), torch.empty_like(bw)
        c2, o2 = torch.zeros_like(c), torch.empty_like(o)
        gpu_ctx.sa_build_dev(text, n, sigma, sa2)
        gpu_ctx.bwt_tables_dev(text, sa2, N, sigma, c2, o2, bw2)
        assert bool((sa2 == sa).all()) and bool((bw2 == bw).all())
        assert bool((c2 == c).all()) and bool((o2 == o).all())


@pytest.mark.parametrize("log2n", [28, 30])
def test_host_pointer_path_at_baseline_sizes(gpu_ctx, log2n):
    """The drop-in call itself at BASELINE's sizes: build_complete_table(letters, true) and
    sa_is_construction(symbols, 5) on HOST memory -- strlen + remap, the pager and the pinned slabs, more than 4 GiB of D2H,
    the o_indices / ro_indices fill, the block cache -- checked on the arrays a stralg caller reads: sa->array against the
    reference's SHA-256 / chunk hashes / sampled entries (golden_big.npz: the unmodified sa_is_mem_construction), c_table and
    O(a, N) through o_indices against the reference's counts, every row of o_table and ro_table against the device path's
    tables (themselves pinned to the reference by test_full_size_properties: stralg/bwt.c:134-161, sa_is.c:466-509)."""
    import psutil
    import torch
    from stralg_amd import verify
    from stralg_amd.benchlegs import cabi, pins
    n, sigma = 1 << log2n, 5
    N = n + 1
    with_ro_bytes = (N + 1) * sigma * 4 * 2 + N * 4 + (N + 1) * 8 * 2 + 4 * n
    fwd_bytes = (N + 1) * sigma * 4 + N * 4 + (N + 1) * 8 + 4 * n
    avail = psutil.virtual_memory().available
    if avail < fwd_bytes * 1.2:
        pytest.skip(f"host has {avail >> 30} GiB free; the forward tables of 2^{log2n} symbols need {fwd_bytes >> 30} GiB")
    include_reverse = avail >= with_ro_bytes * 1.2
    lib = cabi.declare(gpu_ctx.lib)
    text = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
    gpu_ctx.synth_dev(text, n, sigma, 42)
    text[n] = 0
    x = text.cpu().numpy()                      # symbols 1 .. 4, terminated
    letters = np.frombuffer(b"\0ACGT", dtype=np.uint8)[x]  # the byte string a caller holds (terminated: x[n] = 0)
    try:
        t = lib.build_complete_table(letters.ctypes.data, include_reverse)
        assert t.contents.remap_table.contents.alphabet_size == sigma
        assert (np.ctypeslib.as_array(t.contents.sa.contents.string, shape=(N,)) == x).all(), "sa->string is not the remapped record"
        pin = pins.host_table_pin(t, n)
        assert pin is not None and pin["match"], pin
        # every row of the host tables against the device path on the same record
        host_o = np.ctypeslib.as_array(t.contents.o_table, shape=((N + 1) * sigma,))
        host_ro = np.ctypeslib.as_array(t.contents.ro_table, shape=((N + 1) * sigma,)) if include_reverse else None
        host_sa = np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(N,))
        sa = torch.empty(N, dtype=torch.int32, device="cuda")
        bw = torch.empty(N, dtype=torch.uint8, device="cuda")
        c = torch.zeros(sigma, dtype=torch.int32, device="cuda")
        o = torch.empty((N + 1) * sigma, dtype=torch.int32, device="cuda")

        def same(host, dev, what):
            step = 1 << 28
            for s0 in range(0, host.size, step):
                up = torch.from_numpy(host[s0:s0 + step].view(np.int32)).cuda()
                assert bool((up == dev[s0:s0 + step]).all()), f"{what}: host entries [{s0}, {s0 + step}) differ from the device path's"
                del up

        for direction, host_table in (("forward", host_o), ("reverse", host_ro)):
            if host_table is None:
                continue
            d_text = text
            if direction == "reverse":
                d_text = torch.empty_like(text)
                gpu_ctx.reverse_dev(text, n, d_text)
            gpu_ctx.sa_bwt_build_dev(d_text, n, sigma, sa, bw)
            gpu_ctx.bwt_tables_from_bwt_dev(bw, N, sigma, c, o)
            gpu_ctx.trim()
            assert len(verify.verify_build_on_device(d_text, n, sigma, sa, bw, c, o)) == 3
            if direction == "forward":
                _reference_pins(log2n, sigma, sa, bw, c, o)
                same(host_sa, sa, "sa->array")
            same(host_table, o, "o_table" if direction == "forward" else "ro_table")
            if direction == "reverse":
                del d_text
        del sa, bw, c, o, host_o, host_ro, host_sa
        torch.cuda.empty_cache()
        lib.completely_free_bwt_table(t)
        # sa_is_construction / sa_is_mem_construction on the remapped symbols in host memory (sa_is.c:466-509, sa_is_mem.c:471-494)
        for fn in (lib.sa_is_construction,) + ((lib.sa_is_mem_construction,) if log2n <= 28 else ()):
            a = fn(x.ctypes.data, sigma)
            assert a.contents.length == N and C.cast(a.contents.string, C.c_void_p).value == x.ctypes.data  # borrowed, not copied
            pin = pins.host_sa_pin(np.ctypeslib.as_array(a.contents.array, shape=(N,)), n, sigma)
            lib.free_suffix_array(a)
            assert pin["match"], pin
    finally:
        lib.stralg_amd_release()  # the calling thread's context and its cached host blocks (up to 62 GiB here)
        gpu_ctx.trim()
        torch.cuda.empty_cache()


def test_wide_induction_forms_agree_at_size(gpu_ctx):
    """More than 8 buckets through the LMS sort + induced-sort passes at sizes where round 4's forms are all taken: every
    bucket's other-region round up front (bigram counts; sx_induce_wide.hpp) against a launch set per bucket
    (SX_FLAG_INDUCE_NO_HOIST), second rounds of 8 - 32 thousand entries taken by the tail kernel tile after tile, and the
    direct sort of all suffixes with its first pass computing the keys or reading a key kernel's -- one suffix array,
    the sorted permutation (verify.py)."""
    import torch
    from stralg_amd import verify
    for log2n, sigma in ((27, 64), (26, 200), (25, 12)):
        n = 1 << log2n
        text = torch.empty(n, dtype=torch.uint8, device="cuda")
        gpu_ctx.synth_dev(text, n, sigma, 77 + log2n)
        got = []
        try:
            variants = [(True, True, True), (True, False, True)]
            if sigma >= 17:  # (the direct sort takes alphabets of 17 symbols and more)
                variants += [(False, True, True), (False, True, False)]
            for no_direct, hoist, text_keys in variants:
                gpu_ctx.set_no_direct_sort(no_direct)
                gpu_ctx.set_induce_hoist(hoist)
                gpu_ctx.set_text_keys(text_keys)
                sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
                gpu_ctx.sa_build_dev(text, n, sigma, sa)
                st = gpu_ctx.last_stats()
                assert (st["lms_path"] in (1, 2)) == no_direct, (log2n, sigma, st)
                got.append((sa, st["induce_rounds"]))
        finally:
            gpu_ctx.set_no_direct_sort(False)
            gpu_ctx.set_induce_hoist(True)
            gpu_ctx.set_text_keys(True)
        verify.verify_sa_on_device(text, got[0][0], n)
        for sa, _ in got[1:]:
            assert bool((sa == got[0][0]).all()), (log2n, sigma)
        assert got[0][1] < got[1][1]  # (the up-front form queues about half the rounds)
        del got, text
        torch.cuda.empty_cache()


def test_wide_tables_beyond_launch_limit(gpu_ctx):
    """sigma > 8 tables at N = 2^30 + 1: one workgroup per 64-row tile would be 2^24 + 1 workgroups of 256
    threads, more threads than a launch can hold (the kernels loop over tiles instead).  The O rows are checked
    through the one-hot property, the C table against a bincount."""
    import torch
    sigma, N = 9, (1 << 30) + 1
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    bw = torch.randint(1, sigma, (N,), dtype=torch.uint8, device="cuda", generator=g)
    bw[12345] = 0
    c = torch.zeros(sigma, dtype=torch.int32, device="cuda")
    o = torch.empty((N + 1) * sigma, dtype=torch.int32, device="cuda")
    gpu_ctx.bwt_tables_from_bwt_dev(bw, N, sigma, c, o)
    counts = torch.bincount(bw[: 1 << 28].long(), minlength=sigma)
    for s in range(1 << 28, N, 1 << 28):
        counts += torch.bincount(bw[s: min(N, s + (1 << 28))].long(), minlength=sigma)
    assert bool((c.long() == torch.cumsum(counts, 0) - counts).all())
    o = o.view(N + 1, sigma)
    assert bool((o[0] == 0).all()) and bool((o[N].long() == counts).all())
    step = 1 << 24
    for s in range(0, N, step):
        e = min(N, s + step)
        d = o[s + 1: e + 1] - o[s: e]
        onehot = torch.nn.functional.one_hot(bw[s:e].long(), sigma).to(torch.int32)
        assert bool((d == onehot).all())


# ---- FASTA ingest and remap ---------------------------------------

def test_fasta_golden(gpu_ctx, golden_fasta):
    from conftest import check_fasta
    check_fasta(gpu_ctx.fasta_records, golden_fasta)


def test_fasta_pack_byte_soup(gpu_ctx):
    """the packing kernels classify a thread's 16 bytes four at a time in their words (round 5): any bytes in any order, around
    the 16-byte and 4096-byte boundaries and up to a megabyte, against the C restatement of bioinf/fasta.c:26-135"""
    from conftest import check_fasta_soup
    check_fasta_soup(gpu_ctx, np.random.default_rng(19), 500,
                     [0, 1, 2, 15, 16, 17, 31, 33, 4095, 4096, 4097, 8191, 8193, 12288, 20000, 65536, 262145, 1 << 20])


def test_fasta_reference_named_c_api(gpu_ctx, golden_fasta, tmp_path):
    """load_fasta_records and friends (bioinf/fasta.h) as the reference's fasta_test.c drives them"""
    class Rec(C.Structure):
        _fields_ = [("name", C.c_char_p), ("seq", C.POINTER(C.c_uint8)), ("seq_len", C.c_uint32)]

    class It(C.Structure):
        _fields_ = [("rec", C.c_void_p)]

    lib = gpu_ctx.lib
    lib.load_fasta_records.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.load_fasta_records.restype = C.c_void_p
    lib.free_fasta_records.argtypes = [C.c_void_p]
    lib.number_of_fasta_records.argtypes = [C.c_void_p]
    lib.number_of_fasta_records.restype = C.c_uint32
    lib.lookup_fasta_record_by_name.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Rec)]
    lib.lookup_fasta_record_by_name.restype = C.c_bool
    lib.init_fasta_iter.argtypes = [C.POINTER(It), C.c_void_p]
    lib.next_fasta_record.argtypes = [C.POINTER(It), C.POINTER(Rec)]
    lib.next_fasta_record.restype = C.c_bool
    good = tmp_path / "ref.fa"
    good.write_bytes(golden_fasta["ref/ref.fa"]["file"])
    err = C.c_int(-1)
    h = lib.load_fasta_records(str(good).encode(), C.byref(err))
    assert h and err.value == 0 and lib.number_of_fasta_records(h) == 5  # fasta_test.c:24,53
    it, rec, seen = It(), Rec(), []
    lib.init_fasta_iter(C.byref(it), h)
    while lib.next_fasta_record(C.byref(it), C.byref(rec)):
        seen.append((rec.name, bytes(bytearray(rec.seq[i] for i in range(rec.seq_len)))))
    want = pyoracle.fasta_records_of(golden_fasta["ref/ref.fa"]["packed"], 5)
    assert seen == want[::-1]  # reverse file order (fasta.c:131-134; fasta-test-expected.txt starts with ref5)
    assert lib.lookup_fasta_record_by_name(h, b"ref2", C.byref(rec)) and rec.seq_len == 111  # fasta_test.c:57-66
    assert not lib.lookup_fasta_record_by_name(h, b"noname", C.byref(rec))
    lib.free_fasta_records(h)
    assert not lib.load_fasta_records(b"no such file", C.byref(err)) and err.value == 1  # CANNOT_OPEN_FILE
    bad = tmp_path / "malformed.fa"
    bad.write_bytes(golden_fasta["ref/malformed.fa"]["file"])
    assert not lib.load_fasta_records(str(bad).encode(), C.byref(err)) and err.value == 2  # MALFORMED_FILE
    assert not lib.load_fasta_records(str(bad).encode(), None)


def test_production_genomes_through_the_farm(gpu_ctx, golden_genomes, tmp_path):
    """the read-mapper's preprocessing loop (bwt_readmapper.c:54-62) on its own genomes
    (tools/readmappers/data/genomes/hg38-1000.fa, hg38-10000.fa): load_fasta_records ->
    stralg_amd_fasta_tables_batch (include_reverse) -> write_complete_bwt_info, every array and the index file
    against what the unmodified reference produced (tests/golden/golden_genomes.npz)"""
    from conftest import check_genomes
    check_genomes(gpu_ctx.lib, golden_genomes, tmp_path)


def test_fasta_to_tables_on_device(gpu_routed):
    """a FASTA image that never leaves the GPU: pack, per record remap + suffix array + BWT + C/O, each checked
    against the oracle working from the file on the host"""
    import torch
    rng = np.random.default_rng(21)
    parts, seqs = [], []
    for k, (n, letters) in enumerate(((50_000, b"ACGT"), (1, b"A"), (200_003, b"ACGTN"), (30_000, b"ACDEFGHIKLMNPQRSTVWY"))):
        seq = bytes(rng.choice(np.frombuffer(letters, dtype=np.uint8), size=n))
        seqs.append(seq)
        parts.append(b">rec%d description\n" % k + b"\n".join(seq[i:i + 60] for i in range(0, n, 60)) + b"\n")
    data = b"".join(parts)
    bad, packed_want, recs_want = pyoracle.fasta_pack(data)
    assert not bad and [s for _, s in recs_want] == seqs
    d_file = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    d_packed = torch.zeros(len(data) + 1, dtype=torch.uint8, device="cuda")
    d_term = torch.zeros(len(data) + 2, dtype=torch.int32, device="cuda")
    plen, nrec = gpu_routed.fasta_pack_dev(d_file, len(data), d_packed, d_term, d_term.numel())
    assert nrec == 4 and d_packed[:plen].cpu().numpy().tobytes() == packed_want
    term = d_term[: 2 * nrec].cpu().numpy().view(np.uint32)
    for r in range(nrec):
        s0, n = int(term[2 * r]) + 1, int(term[2 * r + 1]) - int(term[2 * r]) - 1
        assert n == len(seqs[r])
        d_sym = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
        sigma, table = gpu_routed.remap_dev(d_packed[s0:], n, d_sym)
        sym_want, sigma_want, table_want = oracle.remap(np.frombuffer(seqs[r], dtype=np.uint8))
        assert sigma == sigma_want and (table == table_want).all()
        assert (d_sym[:n].cpu().numpy() == sym_want).all() and int(d_sym[n]) == 0
        N = n + 1
        sa = torch.empty(N, dtype=torch.int32, device="cuda")
        bw = torch.empty(N, dtype=torch.uint8, device="cuda")
        c = torch.empty(sigma, dtype=torch.int32, device="cuda")
        o = torch.empty((N + 1) * sigma, dtype=torch.int32, device="cuda")
        gpu_routed.sa_bwt_build_dev(d_sym, n, sigma, sa, bw)
        gpu_routed.bwt_tables_from_bwt_dev(bw, N, sigma, c, o)
        sa_want = oracle.sa_is(sym_want, sigma)
        assert (sa.cpu().numpy().view(np.uint32) == sa_want).all(), r
        assert (c.cpu().numpy().view(np.uint32) == oracle.c_table(sym_want, sigma)).all()
        assert (o.cpu().numpy().view(np.uint32) == oracle.o_table(sym_want, sa_want, sigma).ravel()).all()


def test_fasta_record_with_reverse_against_oracle(gpu_routed):
    """What the production caller asks for (bwt_readmapper.c:57: build_complete_table(rec.seq, true)) for a record that
    never leaves the GPU: farm.FastaRecordJob with include_reverse -- image -> pack -> remap -> SA + BWT + C/O, then
    sx_reverse_dev, the reversed string's suffix array and the RO table (bwt.c:147-158) -- every array against the oracle
    (RO = the O table of the reversed string, bwt.c:67-88), at 4 Mi bases and at sizes around the reversal's 16-byte pieces."""
    import torch
    from stralg_amd import farm, workloads
    dev = torch.device("cuda", 0)
    for n, seed in ((1 << 22, 9), (1, 3), (15, 4), (16, 5), (17, 6), (4099, 7)):
        text = torch.from_numpy(synth(n, 5, seed)).to(dev)
        image = workloads.fasta_image(text, "rec")
        job = farm.FastaRecordJob(gpu_routed, image.cpu().pin_memory(), dev, tables=True, include_reverse=True)
        job.upload()
        assert job.build() == n + 1 and job.n == n
        letters = np.frombuffer(b"NACGTN", dtype=np.uint8)[text.cpu().numpy()]
        x, sigma, _ = oracle.remap(letters)  # (a short record may miss a letter: dense codes, remap.c:8-31)
        assert job.sigma == sigma
        rev = x[::-1].copy()
        assert (job.d_text[:n].cpu().numpy() == x).all()
        assert (job.d_rev[:n].cpu().numpy() == rev).all() and int(job.d_rev[n]) == 0
        sa_want, rsa_want = oracle.sa_is(x, sigma), oracle.sa_is(rev, sigma)
        assert (job.sa.cpu().numpy().view(np.uint32) == sa_want).all(), n
        assert (job.rsa.cpu().numpy().view(np.uint32) == rsa_want).all(), n
        assert (job.c.cpu().numpy().view(np.uint32) == oracle.c_table(x, sigma)).all()
        assert (job.rc.cpu().numpy().view(np.uint32) == oracle.c_table(rev, sigma)).all()
        assert (job.o.cpu().numpy().view(np.uint32) == oracle.o_table(x, sa_want, sigma).ravel()).all(), n
        assert (job.ro.cpu().numpy().view(np.uint32) == oracle.o_table(rev, rsa_want, sigma).ravel()).all(), n
        job.set_include_reverse(False)
        assert job.ro is None and job.build() == n + 1
        del job, text, image
        torch.cuda.empty_cache()


def test_fasta_record_at_full_size(gpu_ctx):
    """BASELINE.json configs[4]'s unit of work at its size: one FASTA record of 2^30 bases (60-column lines, as
    bench.py --gpus N gives every rank) from the file image in HBM through sx_fasta_pack_dev -> sx_remap_dev ->
    sx_sa_bwt_build_dev -> sx_bwt_tables_from_bwt_dev (bwt_readmapper.c:54-62), then the size-independent proofs:
    pack + remap reproduce the record; the suffix array is the sorted permutation; bwt = text[sa - 1]; C and every
    O row"""
    import torch
    from stralg_amd import farm, verify, workloads
    n = 1 << 30
    dev = torch.device("cuda", 0)
    text = torch.empty(n, dtype=torch.uint8, device=dev)
    gpu_ctx.synth_dev(text, n, 5, 77)
    image = workloads.fasta_image(text, "record0 a 1 GiB chromosome")
    assert image.numel() == len(b">record0 a 1 GiB chromosome\n") + n + (n + 59) // 60
    job = farm.FastaRecordJob(gpu_ctx, image.cpu(), dev, tables=True)
    del image
    job.upload()
    assert job.build() == n + 1
    assert job.n == n and job.sigma == 5
    assert bool((job.d_text[:n] == text).all()) and int(job.d_text[n]) == 0
    st = gpu_ctx.last_stats()
    assert st["lms_path"] == 1 and st["n"] == n
    del text
    job.d_file = job.d_packed = None
    torch.cuda.empty_cache()
    gpu_ctx.trim()
    done = verify.verify_build_on_device(job.d_text, n, 5, job.sa, job.bwt, job.c, job.o)
    assert len(done) == 3


def test_fasta_large_image(gpu_ctx):
    """64 MiB of sequence in 60-column lines: the packed image against the oracle's"""
    import torch
    rng = np.random.default_rng(4)
    parts = []
    for k in range(3):
        n = (20 << 20) + 12345 * k
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
        full = n // 60
        lines = np.concatenate([seq[: full * 60].reshape(full, 60), np.full((full, 1), ord("\n"), dtype=np.uint8)], axis=1)
        parts.append(b">chr%d\n" % k + lines.tobytes() + seq[full * 60:].tobytes() + b"\n")
    data = b"".join(parts)
    bad, packed_want, recs = pyoracle.fasta_pack(data)
    assert not bad and len(recs) == 3
    d_file = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    d_packed = torch.zeros(len(data) + 1, dtype=torch.uint8, device="cuda")
    plen, nrec = gpu_ctx.fasta_pack_dev(d_file, len(data), d_packed)
    assert nrec == 3 and plen == len(packed_want)
    assert d_packed[:plen].cpu().numpy().tobytes() == packed_want


def test_serialisation_bytes(gpu_ctx, tmp_path):
    """row 1 of the next scope: the reference's index file format, written from host tables and streamed from
    the device, byte for byte (tests/golden/golden_fasta.npz serial/*, made by the reference's own writer)"""
    from conftest import check_serialisation, serial_cases
    check_serialisation(gpu_ctx.lib, serial_cases(), tmp_path)


def test_streamed_index_large(gpu_ctx, tmp_path):
    """64 Mi symbols: the streamed file (several 32 MiB chunks per section) equals the file written from host tables"""
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    lib = gpu_ctx.lib
    lib.build_complete_table.argtypes = [C.c_char_p, C.c_bool]
    lib.build_complete_table.restype = C.c_void_p
    lib.write_complete_bwt_info_fname.argtypes = [C.c_char_p, C.c_void_p]
    lib.write_complete_bwt_info_fname.restype = None
    lib.completely_free_bwt_table.argtypes = [C.c_void_p]
    lib.completely_free_bwt_table.restype = None
    lib.stralg_amd_write_complete_bwt_info_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    raw = np.frombuffer(b"ACGT", dtype=np.uint8)[synth(1 << 26, 5, 3) - 1].tobytes()
    a, b = str(tmp_path / "host.bwt").encode(), str(tmp_path / "stream.bwt").encode()
    t = lib.build_complete_table(raw, False)
    lib.write_complete_bwt_info_fname(a, t)
    lib.completely_free_bwt_table(t)
    f = libc.fopen(b, b"wb")
    assert lib.stralg_amd_write_complete_bwt_info_stream(f, raw, False) == 0
    libc.fclose(f)
    import filecmp
    assert os.path.getsize(a) == 4 + (1 << 26) + 4 * ((1 << 26) + 1) + 388 + 4 * 5 + 4 * 5 * ((1 << 26) + 2) + 1
    assert filecmp.cmp(a, b, shallow=False)


@pytest.mark.parametrize("sigma,log2n", [(256, 24), (21, 22), (128, 20)])
def test_wide_alphabets_direct_sort_and_induction(gpu_ctx, sigma, log2n):
    """alphabets of 16+ symbols: the direct prefix sort of all suffixes (lms_path 3) and, with it switched off, the
    LMS sort + induction over many buckets; suffix array and BWT from both, against the oracle"""
    import torch
    n = 1 << log2n
    x = synth(n, sigma, 17)
    x[100:112] = x[1000:1012]
    x[5:17] = x[1000:1012]
    want = oracle.sa_is(x, sigma)
    want_bwt = oracle.bwt(x, want)
    d_text = torch.from_numpy(x).cuda()
    try:
        for no_direct in (False, True):
            gpu_ctx.set_no_direct_sort(no_direct)
            sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
            bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
            gpu_ctx.sa_bwt_build_dev(d_text, n, sigma, sa, bw)
            assert gpu_ctx.last_stats()["lms_path"] == (1 if no_direct else 3)
            assert (sa.cpu().numpy().view(np.uint32) == want).all(), no_direct
            assert (bw.cpu().numpy() == want_bwt).all(), no_direct
    finally:
        gpu_ctx.set_no_direct_sort(False)


def test_maximum_length(gpu_ctx):
    """n = 2^32 - 2, the longest text the reference's uint32_t lengths allow (suffix_array_internal.c:12): every
    32-bit index computation at its limit; checked on the device (permutation, suffixes strictly increasing)"""
    import gc
    import torch
    gpu_ctx.trim()  # earlier tests' workspace and torch's cached blocks go back first
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 230 * (1 << 30):
        pytest.skip("needs ~230 GiB of device memory")
    n = (1 << 32) - 2
    text = torch.empty(n, dtype=torch.uint8, device="cuda")
    gpu_ctx.synth_dev(text, n, 5, 7)
    sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    gpu_ctx.sa_build_dev(text, n, 5, sa)
    gpu_ctx.trim()
    N = n + 1
    assert int(sa[0]) & 0xFFFFFFFF == n
    pos = sa.long() & 0xFFFFFFFF
    del sa
    rank = torch.full((N + 1,), -1, dtype=torch.int64, device="cuda")
    for s0 in range(0, N, 1 << 30):  # (one scatter of 2^32 elements exceeds torch's own launch limits)
        e0 = min(N, s0 + (1 << 30))
        rank[pos[s0:e0]] = torch.arange(s0, e0, dtype=torch.int64, device="cuda")
    assert bool((rank[:N] >= 0).all()), "not a permutation"
    T = torch.zeros(N + 1, dtype=torch.uint8, device="cuda")
    for s0 in range(0, n, 1 << 30):
        T[s0:min(n, s0 + (1 << 30))] = text[s0:min(n, s0 + (1 << 30))]
    del text
    step = 1 << 28
    for s0 in range(1, N - 1, step):
        e0 = min(N - 1, s0 + step)
        a, b = pos[s0:e0], pos[s0 + 1:e0 + 1]
        ca, cb = T[a], T[b]
        ok = (ca < cb) | ((ca == cb) & (rank[a + 1] < rank[b + 1]))
        assert bool(ok.all()), f"suffixes out of order in slots [{s0}, {e0})"


def test_long_repeats_finish_by_comparison(gpu_ctx):
    """duplications far longer than the refinement keys (up to 50 000 symbols), diverged repeat families and
    microsatellites in 4 Mi symbols of biased DNA: small groups are settled by comparing the suffixes themselves,
    the build stays on the prefix-key path and matches the oracle"""
    n = 1 << 22
    rng = np.random.default_rng(5)
    x = rng.choice(np.array([1, 2, 3, 4], dtype=np.uint8), size=n, p=[0.3, 0.2, 0.2, 0.3])
    elem = rng.integers(1, 5, size=300, dtype=np.uint8)
    for pos in rng.integers(0, n - 400, size=n // 3000):
        copy = elem.copy()
        mut = rng.random(300) < 0.08
        copy[mut] = rng.integers(1, 5, size=int(mut.sum()), dtype=np.uint8)
        x[pos:pos + 300] = copy
    for L, count in ((100, 200), (1000, 30), (6000, 4), (50000, 1)):
        for _ in range(count):
            a, b = rng.integers(0, n - L - 1, size=2)
            x[b:b + L] = x[a:a + L]
    for pos in rng.integers(0, n - 400, size=n // 20000):
        unit = rng.integers(1, 5, size=int(rng.integers(1, 5)), dtype=np.uint8)
        L = int(rng.integers(20, 200))
        x[pos:pos + L] = np.resize(unit, L)
    assert (gpu_ctx.sa_build(x, 5) == oracle.sa_is(x, 5)).all()
    st = gpu_ctx.last_stats()
    assert st["lms_path"] == 1 and st["doubling_rounds"] >= 3, st


def test_repeat_families_refine_in_lds_and_by_sorting(gpu_ctx):
    """families of diverged repeats in 16 Mi symbols: groups of 9 .. 2048 tied suffixes are refined inside a
    workgroup's LDS, longer ones by radix sorts of their members only; the plain-passes mode (every group of more than
    8 through the radix sorts) gives the same suffix array; both match the oracle"""
    x = repeat_families(1 << 24, 11, ((30000, 60, 0.01), (5000, 200, 0.05), (2000, 300, 0.08), (300, 1000, 0.02)))
    ref = oracle.sa_is(x, 5)
    try:
        for mode, tiers in ((0, 3), (1, 2)):
            gpu_ctx.set_sort_mode(mode)
            assert (gpu_ctx.sa_build(x, 5) == ref).all(), mode
            st = gpu_ctx.last_stats()
            assert st["lms_path"] == 1 and st["refine_tiers"] == tiers, (mode, st)
    finally:
        gpu_ctx.set_sort_mode(0)


def test_duplication_beyond_the_comparison_cap(gpu_ctx):
    """a 4.5 Mi-symbol exact duplication in 64 Mi symbols of DNA: the pair comparisons (taken over by whole waves, 1024
    bytes a step) give up beyond 4 Mi symbols and the build falls back to the general path; a 1 Mi-symbol one is
    settled by them and the build stays on the prefix-key path"""
    n = 1 << 26
    x = synth(n, 5, 123)
    y = x.copy()
    y[40_000_000:40_000_000 + (1 << 20)] = y[1000:1000 + (1 << 20)]
    assert (gpu_ctx.sa_build(y, 5) == oracle.sa_is(y, 5)).all()
    assert gpu_ctx.last_stats()["lms_path"] == 1
    L = 4_700_000
    x[40_000_000:40_000_000 + L] = x[1000:1000 + L]
    assert (gpu_ctx.sa_build(x, 5) == oracle.sa_is(x, 5)).all()
    assert gpu_ctx.last_stats()["lms_path"] == 2


def test_near_identical_copies_double_by_waves(gpu_ctx):
    """16 copies of one random 2 Mi-symbol text with 0.1 % of the symbols replaced (what a collection of assemblies of one
    species looks like): the general path's doubling rounds order the groups of 16 by one wave each; the plain-passes
    mode orders them by radix sorts; both match the oracle.  More than 2^23 samples: names and first ranks take the
    two-pass permutation scatter"""
    rng = np.random.default_rng(23)
    one = rng.integers(1, 5, size=1 << 21, dtype=np.uint8)
    parts = []
    for _ in range(16):
        c = one.copy()
        hit = rng.random(c.size) < 0.001
        c[hit] = rng.integers(1, 5, size=int(hit.sum()), dtype=np.uint8)
        parts.append(c)
    x = np.concatenate(parts)
    want = oracle.sa_is(x, 5)
    try:
        for mode, bits in ((0, 4), (1, 8)):
            gpu_ctx.set_sort_mode(mode)
            assert (gpu_ctx.sa_build(x, 5) == want).all(), mode
            st = gpu_ctx.last_stats()
            assert st["lms_path"] == 2 and st["refine_tiers"] & bits and st["n_samples"] > 1 << 23, (mode, st)
    finally:
        gpu_ctx.set_sort_mode(0)


def test_tied_groups_at_the_tier_boundaries(gpu_ctx):
    """exact copies of one piece, as many as the refinement tiers' limits (8 | 9 .. 2048 | 2049 ..): groups of exactly that
    many tied suffixes, alone and together, match the oracle"""
    rng = np.random.default_rng(31)
    n = 1 << 22
    for counts in ((8,), (9,), (2047,), (2048,), (2049,), (4096,), (8, 9, 2048, 2049, 64, 3)):
        x = rng.integers(1, 5, size=n, dtype=np.uint8)
        at = 1000
        for g in counts:
            piece = rng.integers(1, 5, size=70, dtype=np.uint8)
            for _ in range(g):
                x[at:at + 70] = piece
                at += 70 + int(rng.integers(1, 30))
        assert at < n
        assert (gpu_ctx.sa_build(x, 5) == oracle.sa_is(x, 5)).all(), counts
        assert gpu_ctx.last_stats()["lms_path"] == 1, counts


def test_copies_at_the_wave_window_boundaries(gpu_ctx):
    """exact copies of a text, as many as the limits of the doubling rounds' wave tier (a window of 64 list slots, owned
    groups begin in its first 48: every group of up to 17 has an owner), alone and mixed with longer groups that only
    sometimes fit a window: the general path matches the oracle"""
    rng = np.random.default_rng(37)
    one = rng.integers(1, 5, size=40000, dtype=np.uint8)
    short = rng.integers(1, 5, size=900, dtype=np.uint8)
    for copies, extra in ((2, 0), (16, 0), (17, 0), (18, 0), (3, 17), (3, 18), (3, 40), (3, 48), (3, 64), (3, 65), (24, 0), (25, 0)):
        x = np.concatenate([one] * copies + [short] * extra + [one[:777]])
        assert (gpu_ctx.sa_build(x, 5) == oracle.sa_is(x, 5)).all(), (copies, extra)
        st = gpu_ctx.last_stats()
        assert st["lms_path"] == 2, (copies, extra, st)


def test_reduced_string_recursion(gpu_ctx):
    """texts whose reduced string has a handful of names (Fibonacci, Thue-Morse, periodic): the pipeline sorts the reduced
    string itself, level below level (sa_is.c:370-387), instead of doubling over all samples; bit-exact against the
    oracle at 4 Mi symbols, by the properties at 256 Mi, and against the doubling path on the same input"""
    import torch
    from stralg_amd import verify, workloads

    def fib(n):
        a, b = b"\x02", b"\x02\x01"
        while len(b) < n:
            a, b = b, b + a
        return np.frombuffer(b[:n], np.uint8).copy()

    tm = np.array([1], np.uint8)
    while tm.size < (1 << 22):
        tm = np.concatenate([tm, 3 - tm])
    rng = np.random.default_rng(8)
    try:
        for x, sigma in ((fib(1 << 22), 3), (tm, 3), (np.tile(rng.integers(1, 5, size=11, dtype=np.uint8), 400000), 5)):
            want = oracle.sa_is(x, sigma)
            for rmin, lv in ((-1, 1), (1 << 30, 0), (5000, 1)):  # (how many levels depends on the text: at least one)
                gpu_ctx.set_recurse_min(rmin)
                sa = gpu_ctx.sa_build(x, sigma)
                st = gpu_ctx.last_stats()
                assert (sa == want).all(), (sigma, rmin)
                assert st["lms_path"] == 2 and (st["recursion_levels"] >= lv if lv else st["recursion_levels"] == 0), str((rmin, st))
        gpu_ctx.set_recurse_min(-1)
        n = 1 << 28
        text, sigma = workloads.make_text(gpu_ctx, "periodic", n, 0, 1, torch.device("cuda", 0))
        sa = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        bw = torch.empty(n + 1, dtype=torch.uint8, device="cuda")
        gpu_ctx.sa_bwt_build_dev(text, n, sigma, sa, bw)
        assert gpu_ctx.last_stats()["recursion_levels"] >= 4
        gpu_ctx.trim()
        verify.verify_build_on_device(text, n, sigma, sa, bw, None, None)
    finally:
        gpu_ctx.set_recurse_min(-1)


def test_differential_fuzz():
    """tools/fuzz_gpu.py: 250 random (size, alphabet, structure, path flag) combinations against the oracle --
    suffix array, C and O tables from (text, sa) and from the fused build.  (This is the harness that found the
    undersized LDS rows of the wide O-table kernel at sigma = 32 and 64.)"""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "250", "77"], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "250 cases ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
