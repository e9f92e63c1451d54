"""The HIP kernel sources, executed by the CPU harness of tests/emu, against the
oracle and the golden vectors.  This checks kernel logic (ballot ranking, tile
carries, scans) where there is no GPU; the `-m gpu` tests repeat the parity checks
on hardware through the product library."""
import os

import numpy as np
import pytest

import oracle
from stralg_amd.synth import synth, repeat_families


def _sa(ctx, x, sigma):
    return ctx.sa_build(np.asarray(x, dtype=np.uint8), sigma)


def test_tiny_and_edges(emu_routed):
    assert _sa(emu_routed, [], 5).tolist() == [0]
    assert _sa(emu_routed, [3], 5).tolist() == [1, 0]
    for x in ([1, 1], [1, 2], [2, 1], [1, 2, 1], [2, 2, 1, 2], [1, 1, 2, 3]):
        assert (_sa(emu_routed, x, 5) == oracle.sa_is_strict(np.array(x, np.uint8), 5)).all(), x
    # loose alphabet_size: the true suffix array, not sort_SA's shortcut (quirk 3)
    assert _sa(emu_routed, [1, 1, 2, 3], 5).tolist() == [4, 0, 1, 2, 3]


def test_golden_small(emu_routed, golden):
    for name, c in golden.items():
        n = c["sym"].size
        if n > 4200 or c["sigma"] > 21 or c["sigma"] == n + 1:
            continue
        assert (_sa(emu_routed, c["sym"], c["sigma"]) == c["sa"]).all(), name


def test_random_and_tile_boundaries(emu_routed):
    rng = np.random.default_rng(3)
    for n in (4095, 4096, 4097, 9000):
        x = rng.integers(1, 5, size=n, dtype=np.uint8)
        assert (_sa(emu_routed, x, 5) == oracle.sa_is(x, 5)).all(), n
    x = rng.integers(1, 256, size=600, dtype=np.uint8)
    assert (_sa(emu_routed, x, 256) == oracle.sa_is(x, 256)).all()


def test_both_lms_paths(emu_ctx):
    """prefix-key LMS sort (with tie refinement rounds) and the general path agree with the oracle"""
    base = oracle.synth(6000, 5, 9)
    seen = set()
    for L in (0, 25, 70, 100):
        x = base.copy()
        if L:
            x[3000:3000 + L] = x[100:100 + L]
            x[4500:4500 + L] = x[100:100 + L]
        want = oracle.sa_is(x, 5)
        for force in (False, True):
            emu_ctx.force_general_path(force)
            assert (_sa(emu_ctx, x, 5) == want).all(), (L, force)
            st = emu_ctx.last_stats()
            seen.add((st["lms_path"], st["doubling_rounds"] > 0))
    emu_ctx.force_general_path(False)
    assert (1, True) in seen and any(p == 2 for p, _ in seen)


def test_near_identical_copies_double_by_waves(emu_ctx):
    """a collection of near-identical sequences (8 copies of one random text, 0.3 % of the symbols replaced): every LMS
    suffix is tied with seven others for hundreds of symbols, the general path's prefix-doubling rounds order the
    small groups by one wave each (doubling_wave_groups_kernel); groups that cross a wave's window or hold more than
    64 members, and every group with SX_FLAG_SORT_MODE 1, take the radix sorts"""
    rng = np.random.default_rng(17)
    one = rng.integers(1, 5, size=8000, dtype=np.uint8)
    parts = []
    for _ in range(8):
        c = one.copy()
        hit = rng.random(c.size) < 0.003
        c[hit] = rng.integers(1, 5, size=int(hit.sum()), dtype=np.uint8)
        parts.append(c)
    x = np.concatenate(parts)
    want = oracle.sa_is(x, 5)
    assert (_sa(emu_ctx, x, 5) == want).all()
    st = emu_ctx.last_stats()
    assert st["lms_path"] == 2 and st["refine_tiers"] & 4 and st["doubling_rounds"] >= 5, st
    emu_ctx.set_sort_mode(1)
    try:
        assert (_sa(emu_ctx, x, 5) == want).all()
        st = emu_ctx.last_stats()
        assert st["lms_path"] == 2 and st["refine_tiers"] & 12 == 8, st
    finally:
        emu_ctx.set_sort_mode(0)
    # 100 copies of a short piece on top: their groups are longer than a wave's window and take the radix sorts in the same
    # rounds in which the waves order the groups of 8
    y = np.concatenate([x] + [one[:200]] * 100)
    assert (_sa(emu_ctx, y, 5) == oracle.sa_is(y, 5)).all()
    st = emu_ctx.last_stats()
    assert st["lms_path"] == 2 and st["refine_tiers"] & 12 == 12, st
    # 100 copies of a text and nothing else: long groups on average, the rounds go straight to the radix sorts
    z = np.concatenate([one[:500]] * 100 + [one[:300]])
    assert (_sa(emu_ctx, z, 5) == oracle.sa_is(z, 5)).all()
    st = emu_ctx.last_stats()
    assert st["lms_path"] == 2 and st["refine_tiers"] & 8, st


def test_static_key_shapes(emu_ctx):
    """the key kernel's static forms for DNA-like texts (prefix lengths of 64 Mi ... 4 Gi symbol inputs), forced
    on small texts: keys and embedded windows by dot products"""
    rng = np.random.default_rng(17)
    x = rng.integers(1, 5, size=9000, dtype=np.uint8)
    x[5000:5040] = x[200:240]  # some ties for the refinement rounds as well
    y = np.concatenate([x[:40], x[:40], x[3:2000]])  # LMS positions inside the first symbols of the text
    z = rng.integers(1, 6, size=7000, dtype=np.uint8)  # five symbols (DNA with N): three-bit window codes
    z[3000:3050] = z[100:150]
    try:
        for text, sigma in ((x, 5), (y, 5), (z, 6), (np.concatenate([z[:30], z[:30], z[:900]]), 6)):
            want = oracle.sa_is(text, sigma)
            for C in (12, 13, 14, 15, 16, 17, 18, 19):
                emu_ctx.set_prefix_symbols(C)
                assert (_sa(emu_ctx, text, sigma) == want).all(), (sigma, C)
                assert emu_ctx.last_stats()["key_slots"] == C
    finally:
        emu_ctx.set_prefix_symbols(0)


def test_hybrid_prefix_sort(emu_ctx):
    """SX_FLAG_SORT_MODE 2: HBM passes on the top 24 key bits, sub-buckets ordered in LDS (sx_localsort.hip), ties listed
    by the same kernel; a sub-bucket too long for a workgroup falls back to LSD passes"""
    rng = np.random.default_rng(11)
    emu_ctx.set_sort_mode(2)
    try:
        # (mode 3: four HBM passes on the top 32 key bits, what texts of 2 Gi symbols and skewed frequencies take)
        for mode, sigma, syms, n in ((2, 5, 17, 9000), (2, 5, 17, 4097), (2, 5, 14, 20000), (2, 6, 16, 12000), (2, 5, 18, 30000),
                                     (3, 5, 17, 9000), (3, 5, 18, 30000), (3, 6, 16, 12000), (3, 5, 15, 5000)):
            emu_ctx.set_sort_mode(mode)
            x = rng.integers(1, sigma, size=n, dtype=np.uint8)
            x[700:760] = x[100:160]      # ties beyond the key: refinement rounds after the local sort
            x[n - 300:n - 260] = x[100:140]
            emu_ctx.set_prefix_symbols(syms)
            want = oracle.sa_is(x, sigma)
            # (four letters, 15 ... 18 key symbols: the first HBM pass lists the LMS suffixes and computes their keys itself --
            #  radix_scatter_lms_kernel --, bit 3 of sort_local; SX_FLAG_TEXT_KEYS_OFF keeps the key kernel)
            for text_keys in (True, False):
                emu_ctx.set_text_keys(text_keys)
                sa, bw = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint8)
                emu_ctx.sa_bwt_build_dev(x, n, sigma, sa, bw)
                st = emu_ctx.last_stats()
                keyed = 8 if (text_keys and sigma == 5 and 15 <= syms <= 18) else 0
                assert st["lms_path"] == 1 and st["sort_local"] == (1 if mode == 2 else 5) + keyed, (mode, sigma, syms, n, text_keys, st)
                assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), (mode, sigma, syms, n, text_keys)
            emu_ctx.set_text_keys(True)
        emu_ctx.set_sort_mode(2)
        # an LMS suffix at every other position (low, high, low, high ...): more than a radix tile's worth in a block of six
        # classification tiles, which the keyed first pass then takes as two halves (lms_slot_span)
        n = 60000
        x = np.empty(n, np.uint8)
        x[0::2] = rng.integers(1, 3, size=n // 2)
        x[1::2] = rng.integers(3, 5, size=n // 2)
        emu_ctx.set_prefix_symbols(18)
        want = oracle.sa_is(x, 5)
        for text_keys in (True, False):
            emu_ctx.set_text_keys(text_keys)
            sa, bw = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint8)
            emu_ctx.sa_bwt_build_dev(x, n, 5, sa, bw)
            st = emu_ctx.last_stats()
            assert st["n_lms"] > n // 2 - 2 and st["sort_local"] & 9 == (9 if text_keys else 1), st
            assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), text_keys
        emu_ctx.set_text_keys(True)
        # 40 copies of a 60-symbol piece: equal keys crowd a bin of the counting pass, that workgroup takes stable passes
        x = rng.integers(1, 5, size=20000, dtype=np.uint8)
        for i in range(40):
            x[300 + 400 * i:360 + 400 * i] = x[100:160]
        emu_ctx.set_prefix_symbols(17)
        want = oracle.sa_is(x, 5)
        # (round 5: the lean kernel orders a crowded bin by its waves -- 64 members a wave, counted against all the bin's
        #  members passed from lane to lane --; SX_FLAG_LOCAL_SORT_LEAN_OFF: the kernel of rounds 3 and 4 for every workgroup,
        #  whose stable passes set bit 1)
        for lean in (True, False):
            emu_ctx.set_local_sort_lean(lean)
            sa = _sa(emu_ctx, x, 5)
            assert emu_ctx.last_stats()["sort_local"] & 7 == (1 if lean else 3), (lean, emu_ctx.last_stats())
            assert (sa == want).all(), lean
        # 90 and 200 copies (two and four pieces of 64 members a bin), some of them differing in the key's last symbols; and
        # 700 copies: more than the waves take (SX_LS2_TEAM_MAX 512) -- that workgroup is left to the other kernel (bit 1)
        for copies, variants, bit1 in ((90, 0, 0), (200, 7, 0), (130, 64, 0), (700, 0, 2), (700, 5, 2)):
            n = 400 * copies + 5000
            x = rng.integers(1, 5, size=n, dtype=np.uint8)
            piece = x[100:160].copy()
            for i in range(copies):
                x[300 + 400 * i:360 + 400 * i] = piece
                if variants and i % 3 == 0:
                    x[300 + 400 * i + 14 + (i // 3) % 4] = 1 + (i // 3) % variants % 4
            want = oracle.sa_is(x, 5)
            for lean in (True, False):
                emu_ctx.set_local_sort_lean(lean)
                sa = _sa(emu_ctx, x, 5)
                assert emu_ctx.last_stats()["sort_local"] & 7 == (1 | bit1 if lean else 3), (copies, variants, lean, emu_ctx.last_stats())
                assert (sa == want).all(), (copies, variants, lean)
        # skewed symbol counts (a genome's): long sub-buckets behind the frequent symbols' prefixes next to short ones, with ties
        # beyond the key
        for syms, n, p in ((17, 40000, (0.55, 0.25, 0.15, 0.05)), (14, 30000, (0.4, 0.4, 0.1, 0.1)), (18, 25000, (0.7, 0.1, 0.1, 0.1))):
            x = rng.choice(np.arange(1, 5, dtype=np.uint8), size=n, p=p)
            x[900:1000] = x[200:300]
            x[n - 500:n - 420] = x[200:280]
            emu_ctx.set_prefix_symbols(syms)
            want = oracle.sa_is(x, 5)
            for lean in (True, False):
                emu_ctx.set_local_sort_lean(lean)
                sa, bw = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint8)
                emu_ctx.sa_bwt_build_dev(x, n, 5, sa, bw)
                st = emu_ctx.last_stats()
                assert st["lms_path"] == 1 and st["sort_local"] & 1, (syms, n, lean, st)
                assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), (syms, n, lean)
        emu_ctx.set_prefix_symbols(17)
        emu_ctx.set_local_sort_lean(True)
        # one 12-symbol prefix in front of thousands of LMS suffixes: a sub-bucket no workgroup can hold
        unit = np.array([1, 3, 2, 4, 4, 2, 3, 1, 1, 3, 2, 4, 2, 1], np.uint8)
        reps = 6500
        x = np.concatenate([np.concatenate([unit, rng.integers(1, 5, size=6, dtype=np.uint8)]) for _ in range(reps)])
        emu_ctx.set_prefix_symbols(17)
        want = oracle.sa_is(x, 5)
        for long_on in (False, True):  # (rounds 1 - 3: plain passes; round 4: listed, and ordered by HBM passes of their own)
            emu_ctx.set_long_subbuckets(long_on)
            sa = _sa(emu_ctx, x, 5)
            st = emu_ctx.last_stats()
            assert (st["sort_local"] & 1, st["long_subbuckets"] > 0) == ((1, True) if long_on else (0, False)), st
            assert (sa == want).all(), long_on
        # 7000 copies of a 30-symbol piece scattered over a random text: a handful of sub-buckets no workgroup can hold with a
        # tenth of the pairs -- listed by the workgroups that meet them and ordered by HBM passes of their own
        # (sx_long_subbuckets); without the list (rounds 1 - 3) the whole sort falls back to plain passes
        n = 1 << 20
        x = rng.integers(1, 5, size=n, dtype=np.uint8)
        piece = np.array([2, 4, 1, 3, 3, 1, 4, 2, 1, 2, 4, 3, 1, 1, 3, 2, 4, 4, 1, 2, 3, 1, 4, 2, 2, 3, 1, 4, 3, 2], np.uint8)
        for a in (rng.choice(n // 32 - 2, size=7000, replace=False) * 32).tolist():
            x[a:a + 30] = piece
        want = oracle.sa_is(x, 5)
        for long_on, text_keys in ((True, True), (True, False), (False, True)):
            emu_ctx.set_long_subbuckets(long_on)
            emu_ctx.set_text_keys(text_keys)
            sa, bw = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint8)
            emu_ctx.sa_bwt_build_dev(x, n, 5, sa, bw)
            st = emu_ctx.last_stats()
            assert (st["sort_local"] & 1, st["long_subbuckets"] > 0) == ((1, True) if long_on else (0, False)), st
            assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), (long_on, text_keys)
        emu_ctx.set_text_keys(True)
        # the direct sort of all suffixes of a wide alphabet through the same path
        x = synth(30000, 256, 5)
        x[100:112] = x[1000:1012]
        emu_ctx.set_prefix_symbols(0)
        sa, bw = np.zeros(x.size + 1, np.uint32), np.zeros(x.size + 1, np.uint8)
        emu_ctx.sa_bwt_build_dev(x, x.size, 256, sa, bw)
        st = emu_ctx.last_stats()
        want = oracle.sa_is(x, 256)
        assert st["lms_path"] == 3, st
        assert (sa == want).all() and (bw == oracle.bwt(x, want)).all()
    finally:
        emu_ctx.set_sort_mode(0)
        emu_ctx.set_prefix_symbols(0)
        emu_ctx.set_text_keys(True)
        emu_ctx.set_long_subbuckets(True)
        emu_ctx.set_local_sort_lean(True)


def test_long_repeats_finish_by_comparison(emu_ctx):
    """duplications far longer than the refinement keys: pairs are settled by comparing the suffixes themselves
    (lms_path stays 1), a repeat with many copies goes through the sorting rounds"""
    base = oracle.synth(200000, 5, 21)  # (few survivors relative to the LMS count: that is when the extra rounds apply)
    for copies, L in ((1, 1000), (2, 500), (12, 100)):
        x = base.copy()
        for k in range(copies):
            x[100000 + k * (L + 37): 100000 + k * (L + 37) + L] = x[500:500 + L]
        assert (_sa(emu_ctx, x, 5) == oracle.sa_is(x, 5)).all(), (copies, L)
        st = emu_ctx.last_stats()
        assert st["lms_path"] == 1 and st["doubling_rounds"] >= 3, (copies, L, st)
    # long comparisons are taken over by the whole wave (1024 bytes a step); one of them here runs into the sentinel
    x = base.copy()
    x[-3000:] = x[700:3700]
    x[60000:62500] = x[700:3200]
    assert (_sa(emu_ctx, x, 5) == oracle.sa_is(x, 5)).all()
    assert emu_ctx.last_stats()["lms_path"] == 1


def test_repeat_families_refine_in_lds_and_by_sorting(emu_ctx):
    """families of diverged repeats leave groups of hundreds and thousands of tied suffixes: groups of 9 .. 2048 members
    are refined inside a workgroup's LDS (refine_mid_groups_kernel), the members of longer ones by radix sorts of the
    compacted sub-list; with SX_FLAG_SORT_MODE 1 every group of more than 8 takes the radix sorts"""
    x = repeat_families(500000, 3, ((2600, 36, 0.002), (100, 120, 0.03), (30, 300, 0.0)))
    ref = oracle.sa_is(x, 5)
    assert (_sa(emu_ctx, x, 5) == ref).all()
    st = emu_ctx.last_stats()
    assert st["lms_path"] == 1 and st["refine_tiers"] == 3, st
    # groups of up to some hundred members only: nothing is left for the radix sorts
    y = repeat_families(200000, 4, ((200, 120, 0.03), (40, 300, 0.0)))
    ref_y = oracle.sa_is(y, 5)
    assert (_sa(emu_ctx, y, 5) == ref_y).all()
    st = emu_ctx.last_stats()
    assert st["lms_path"] == 1 and st["refine_tiers"] == 1, st
    emu_ctx.set_sort_mode(1)
    try:
        assert (_sa(emu_ctx, y, 5) == ref_y).all()
        st = emu_ctx.last_stats()
        assert st["lms_path"] == 1 and st["refine_tiers"] == 2, st
    finally:
        emu_ctx.set_sort_mode(0)


def test_very_long_run(emu_ctx):
    """a run that outlasts the tail kernel's 256 steps of 4096 rounds: the device-wide run jump"""
    rng = np.random.default_rng(33)
    x = np.concatenate([rng.integers(1, 5, size=3000, dtype=np.uint8), np.full(256 * 4096 + 70_000, 3, np.uint8), [1],
                        rng.integers(1, 5, size=3000, dtype=np.uint8)]).astype(np.uint8)
    assert (_sa(emu_ctx, x, 5) == oracle.sa_is(x, 5)).all()
    # two runs of the same symbol alive in the bucket at once (the jump takes both entries), S-type this time
    y = np.concatenate([x[:500], np.full(256 * 4096 + 9000, 2, np.uint8), [4], x[:700], np.full(256 * 4096 + 30_000, 2, np.uint8),
                        [3], x[:300]]).astype(np.uint8)
    assert (_sa(emu_ctx, y, 5) == oracle.sa_is(y, 5)).all()


def test_runs_of_many_lengths(emu_ctx):
    """poly-A tracts and runs of every symbol with lengths 1 ... 60 scattered over a random text: the shrinking rounds of
    a bucket, where entries leave after different numbers of rounds (the tail kernel takes eight rounds at once from
    the entries' windows; the all-in-a-run jump never applies here), for 4, 5 and 7 symbols"""
    rng = np.random.default_rng(21)
    for sigma in (5, 6, 8):
        n = 30000
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
        starts = rng.integers(0, n - 64, size=n // 40)
        lens = rng.integers(1, 61, size=starts.size)
        syms = rng.integers(1, sigma, size=starts.size)
        for a, l, c in zip(starts.tolist(), lens.tolist(), syms.tolist()):
            x[a:a + l] = c
        x[:50] = 1  # a run at the very start of the text (windows shorter than the batch)
        x[n - 45:] = sigma - 1
        assert (_sa(emu_ctx, x, sigma) == oracle.sa_is(x, sigma)).all(), sigma


def test_runs_closed_form(emu_ctx):
    """runs of 1 ... 330 symbols with differing lengths alive in a bucket at once: the tail kernel's closed form (the harness
    takes it from 4 entries on) -- in both passes, with runs beyond its 255-symbol look (those are carried on), at both
    ends of the text, and with several entries a thread; SA and BWT against the oracle"""
    rng = np.random.default_rng(7)
    cases = [(5, 60000, 350), (8, 60000, 350)]
    if os.environ.get("STRALG_EMU_ASAN") != "1":
        cases.append((3, 60000, 200))
    for sigma, n, k in cases:
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
        starts = rng.integers(0, n - 400, size=k)
        lens = rng.integers(1, 331, size=k)
        syms = rng.integers(1, sigma, size=k)
        for a, l, c in zip(starts.tolist(), lens.tolist(), syms.tolist()):
            x[a:a + l] = c
        x[:270] = 1
        x[n - 300:] = sigma - 1
        want = oracle.sa_is(x, sigma)
        sa, bw = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint8)
        emu_ctx.sa_bwt_build_dev(x, n, sigma, sa, bw)
        assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), sigma
    if os.environ.get("STRALG_EMU_ASAN") != "1":
        # 2600 runs of the largest symbol alive at once: three entries a thread
        n = 400000
        x = rng.integers(1, 5, size=n, dtype=np.uint8)
        starts = np.sort(rng.choice(n // 150 - 2, size=2600, replace=False)) * 150
        for a, l in zip(starts.tolist(), rng.integers(20, 140, size=2600).tolist()):
            x[a:a + l] = 4
        assert (_sa(emu_ctx, x, 5) == oracle.sa_is(x, 5)).all()


def test_short_records_direct_sort(emu_ctx):
    """SX_FLAG_SMALL_DIRECT_MAX (on by default outside the tests): texts of at most 16 symbols and 2^24 suffixes are sorted
    directly, all suffixes by prefix key (lms_path 3) -- a third of the launches of classification + LMS sort + induced
    passes; SA, BWT, C and O against the oracle for 3 ... 16 symbols, with repeats (tie refinement), runs, texts too
    repetitive for it (they go on to the usual path), and the limit itself"""
    rng = np.random.default_rng(17)
    try:
        emu_ctx.set_small_direct_max(-1)
        for sigma, n in ((5, 100), (5, 70000), (3, 3000), (4, 20000), (8, 40000), (16, 30000), (12, 9000), (5, 64), (6, 200000)):
            x = rng.integers(1, sigma, size=n, dtype=np.uint8)
            if n > 1000:
                x[200:260] = x[700:760]
                x[n - 100:n - 60] = x[300:340]
                x[n // 2:n // 2 + 300] = 1 + (sigma > 3)
            want = oracle.sa_is(x, sigma)
            sa, c, o = emu_ctx.build_tables(x, sigma)
            st = emu_ctx.last_stats()
            assert st["lms_path"] == 3, (sigma, n, st)
            assert (sa == want).all() and (c == oracle.c_table(x, sigma)).all() and (o == oracle.o_table(x, want, sigma)).all(), (sigma, n)
        # too repetitive for a prefix sort (a period of 2, then all equal): on to the usual path, same answer
        for x, sigma in ((np.tile(np.array([1, 2], np.uint8), 3000), 3), (np.full(5000, 1, np.uint8), 2)):
            assert (emu_ctx.sa_build(x, sigma) == oracle.sa_is(x, sigma)).all()
            assert emu_ctx.last_stats()["lms_path"] != 3
        # the limit counts suffixes (n + 1)
        x = rng.integers(1, 5, size=4999, dtype=np.uint8)
        for limit, path in ((5000, 3), (4999, 1), (0, 1)):
            emu_ctx.set_small_direct_max(limit)
            assert (emu_ctx.sa_build(x, 5) == oracle.sa_is(x, 5)).all()
            assert emu_ctx.last_stats()["lms_path"] == path, limit
    finally:
        emu_ctx.set_small_direct_max(0)


def test_both_induce_round_forms(emu_ctx):
    """large rounds (count / offsets / scatter launches; for more than 8 buckets the radix-pass form over tiles of 8192
    entries, with its one-launch and its three-launch offsets) and small rounds (one chained launch, the tail kernel)"""
    rng = np.random.default_rng(12)
    x = np.concatenate([np.full(5000, 1, np.uint8), rng.integers(1, 5, size=9000, dtype=np.uint8)])
    want = oracle.sa_is(x, 5)
    y = rng.integers(1, 12, size=60000, dtype=np.uint8)  # 11 symbols: 32-bit windows, buckets of ~5500 entries
    y[20000:20040] = y[100:140]
    want_y = oracle.sa_is(y, 12)
    z = rng.integers(1, 40, size=30000, dtype=np.uint8)  # 39 symbols: 64-bit windows
    z[10000:13000] = 7                                   # a bucket of 3000 entries with a long run
    want_z = oracle.sa_is(z, 40)
    try:
        for thr in (0, 2048, 5000, 1 << 19):
            emu_ctx.set_chain_max_entries(thr)
            assert (_sa(emu_ctx, x, 5) == want).all(), thr
        for thr in (0, 1000, -1):
            emu_ctx.set_chain_max_entries(thr)
            emu_ctx.set_no_direct_sort(True)
            for t, sg, w in ((y, 12, want_y), (z, 40, want_z)):
                sa, bw = np.zeros(t.size + 1, np.uint32), np.zeros(t.size + 1, np.uint8)
                emu_ctx.sa_bwt_build_dev(t, t.size, sg, sa, bw)
                assert (sa == w).all() and (bw == oracle.bwt(t, w)).all(), (thr, sg)
    finally:
        emu_ctx.set_chain_max_entries(-1)
        emu_ctx.set_no_direct_sort(False)


def test_bwt_tables(emu_ctx, golden):
    for name in ("ref/mississippi", "ref/serialise", "struct/periodic", "ref/fasta0"):
        c = golden[name]
        ct, ot = emu_ctx.bwt_tables(c["sym"], c["sa"], c["sigma"])
        assert (ct == c["c"]).all() and (ot == c["o"]).all(), name
    # wide-alphabet O kernel (8 < sigma <= 128) and the last row across a tile edge
    rng = np.random.default_rng(9)
    for sigma, n in ((21, 700), (128, 300), (32, 1100), (33, 600), (64, 900), (65, 300), (5, 1023), (5, 1024), (5, 2049), (3, 1500),
                     (4, 1025), (6, 2100), (7, 1030), (8, 1500), (2, 700)):
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
        sa = oracle.sa_is(x, sigma)
        ct, ot = emu_ctx.bwt_tables(x, sa, sigma)
        assert (ct == oracle.c_table(x, sigma)).all(), (sigma, n)
        assert (ot == oracle.o_table(x, sa, sigma)).all(), (sigma, n)


def test_fused_build_tables(emu_routed, golden):
    """sx_build_tables: the BWT handed over by the induced sort's symbol windows (incl. refills
    when a window runs dry: long runs, monotone stretches, byte alphabets)"""
    import stralg_amd
    for name in ("ref/mississippi", "struct/all-a", "struct/decreasing", "struct/runs", "ref/fasta1"):
        c = golden[name]
        t = stralg_amd.build_complete_table(bytes(c["raw"]), True, emu_routed)
        assert (t.sa.array == c["sa"]).all() and (t.c_table == c["c"]).all(), name
        assert (t.o_table == c["o"]).all() and (t.ro_table == c["ro"]).all(), name
    rng = np.random.default_rng(10)
    for sigma, n in ((5, 3000), (100, 900), (17, 1500)):
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
        sa, ct, ot = emu_routed.build_tables(x, sigma)
        want = oracle.sa_is(x, sigma)
        assert (sa == want).all() and (ct == oracle.c_table(x, sigma)).all(), (sigma, n)
        assert (ot == oracle.o_table(x, want, sigma)).all(), (sigma, n)


def test_inverse_of_a_non_permutation_stays_inside(emu_ctx):
    """sx_sa_inverse_lcp on an array that is not a permutation (a duplicated value, hence a missing one, in windows of
    the two-pass form that differ): SX_E_ARG, and no store outside the arrays.  The window that misses a value has an
    unwritten slot in the pair buffer, which holds whatever the SORT slab held before -- so the slab is dirtied by a
    build first (the harness's fresh slab would be zeros); under tests/test_emu_asan.py an out-of-range store is a report."""
    from stralg_amd import StralgAmdError
    x = synth(9000, 5, 77)
    emu_ctx.sa_build(x, 5)  # leaves sort keys (large words) in the slab the pairs are staged in
    sa = oracle.sa_is(x, 5)
    # (windows of 256 targets in the harness, their fine windows of 64: the last pair overflows a fine window only)
    for src, dst in ((10, 8000), (8000, 10), (10, 200)):
        bad = sa.copy()
        bad[np.flatnonzero(sa == dst)[0]] = src  # value `src` twice, `dst` never: one window overflows, one has a slot left
        with pytest.raises(StralgAmdError):
            emu_ctx.inverse_lcp(x, bad, want_lcp=False)
    bad = sa.copy()
    bad[np.flatnonzero(sa == 301)[0]] = 300  # both in one window: its count is right, nothing to notice (compute_inverse,
    emu_ctx.inverse_lcp(x, bad, want_lcp=False)  # suffix_array.c:55-62, checks nothing either) -- but every store stays inside
    bad = sa.copy()
    bad[5] = 0xFFFFFFF0  # a value beyond N
    with pytest.raises(StralgAmdError):
        emu_ctx.inverse_lcp(x, bad, want_lcp=False)
    inv, _ = emu_ctx.inverse_lcp(x, sa, want_lcp=False)  # the context is as usable as before
    assert (inv == oracle.inverse(sa)).all()


def test_next_rows(emu_ctx, golden):
    """inverse, LCP and batched exact search kernels against the reference's vectors"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_next.npz"))
    for name in ("ref/mississippi", "ref/repetitive", "struct/fibonacci", "struct/all-a", "rand/s21/n1000"):
        c = golden[name]
        inv, lcp = emu_ctx.inverse_lcp(c["sym"], c["sa"])
        assert (inv == z[name + "/inverse"]).all() and (lcp == z[name + "/lcp"]).all(), name
    # LCP over many chunks of 64 positions, long common prefixes (the chunk starts take their values from the sampled
    # levels): a Fibonacci string, a string of period 7, random DNA with a 3000-symbol duplication
    fib_a, fib_b = np.array([1], np.uint8), np.array([1, 2], np.uint8)
    while fib_b.size < 5000:
        fib_a, fib_b = fib_b, np.concatenate([fib_b, fib_a])
    dup = synth(9000, 5, 4)
    dup[5000:8000] = dup[500:3500]
    for x, sigma in ((fib_b, 3), (np.tile(np.array([1, 2, 2, 1, 3, 1, 2], np.uint8), 700), 4), (dup, 5)):
        sa = oracle.sa_is(x, sigma)
        inv, lcp = emu_ctx.inverse_lcp(x, sa)
        assert (inv == oracle.inverse(sa)).all() and (lcp == oracle.lcp(x, sa)).all()
    for name in ("ref/mississippi", "ref/fasta3", "struct/periodic"):
        c = golden[name]
        pats, offs, lr = z[name + "/patterns"], z[name + "/offsets"], z[name + "/lr"]
        l, r = np.zeros(lr.shape[0], np.uint32), np.zeros(lr.shape[0], np.uint32)
        emu_ctx.bwt_exact_search_dev(np.ascontiguousarray(c["c"]), np.ascontiguousarray(c["o"]), c["sa"].size, c["sigma"],
                                     pats, offs, lr.shape[0], l, r)
        assert (l == lr[:, 0]).all() and (r == lr[:, 1]).all(), name
    # thousands of patterns at once: they are searched in the order of their last symbols (a radix sort of the pattern
    # numbers); patterns of the text, mutated patterns, a one-symbol and an empty one, each against the oracle's search
    x = synth(6000, 5, 9)
    sa = oracle.sa_is(x, 5)
    ct, ot = oracle.c_table(x, 5), oracle.o_table(x, sa, 5)
    rng = np.random.default_rng(3)
    pats, offs = [], [0]
    for k in range(5000):
        m = int(rng.integers(0, 24)) if k % 50 else int(k % 3 == 0)
        a = int(rng.integers(0, x.size - 24))
        pt = x[a:a + m].copy()
        if k % 3 == 0 and m:
            pt[int(rng.integers(0, m))] = int(rng.integers(1, 5))
        pats.append(pt)
        offs.append(offs[-1] + m)
    flat = np.concatenate(pats).astype(np.uint8) if offs[-1] else np.zeros(1, np.uint8)
    offs = np.array(offs, np.uint32)
    l, r = np.zeros(5000, np.uint32), np.zeros(5000, np.uint32)
    emu_ctx.bwt_exact_search_dev(np.ascontiguousarray(ct), np.ascontiguousarray(ot), sa.size, 5, flat, offs, 5000, l, r)
    for k in range(0, 5000, 7):
        want = oracle.bwt_exact_search(ct, ot, 5, pats[k])
        got = (int(l[k]), int(r[k]))
        assert got == want or (got[0] >= got[1] and want[0] >= want[1]), (k, got, want)


def test_fasta_ingest_and_remap(emu_routed, golden_fasta):
    """FASTA packing (scan + compaction) and remap kernels against the reference's vectors"""
    from conftest import check_fasta
    check_fasta(emu_routed.fasta_records, golden_fasta)
    rng = np.random.default_rng(5)
    for n in (0, 1, 15, 16, 17, 4097, 70001):
        x = rng.choice(np.frombuffer(b"ACGTNRYKM-", dtype=np.uint8), size=n).astype(np.uint8)
        out = np.full(n + 1, 99, dtype=np.uint8)
        sigma, table = emu_routed.remap_dev(x if n else None, n, out)
        want, want_sigma, want_table = oracle.remap(x) if n else (np.zeros(0, np.uint8), 1, None)
        assert sigma == want_sigma and (out[:n] == want).all() and out[n] == 0, n
        if n:
            assert (table == want_table).all()
    with pytest.raises(Exception):  # more than 127 distinct symbols (remap.h:14-18)
        x = np.arange(1, 200, dtype=np.uint8)
        emu_routed.remap_dev(x, x.size, np.zeros(x.size + 1, np.uint8))


def test_fasta_pack_byte_soup(emu_ctx):
    """the packing kernels classify a thread's 16 bytes four at a time in their words (round 5): any bytes in any order, around
    the 16-byte and 4096-byte boundaries, against the C restatement of bioinf/fasta.c:26-135"""
    from conftest import check_fasta_soup
    check_fasta_soup(emu_ctx, np.random.default_rng(17), 250, [0, 1, 2, 15, 16, 17, 31, 33, 4095, 4096, 4097, 8191, 8193, 12288, 20000])


def test_wide_alphabets_direct_sort_and_induction(emu_ctx):
    """alphabets of 16+ symbols: the direct prefix sort of all suffixes (lms_path 3) and, with it switched off, the
    LMS sort + induction over many buckets; suffix array and BWT from both"""
    for sigma, n in ((256, 12000), (21, 9000), (100, 5000), (128, 3000), (18, 5000)):
        x = synth(n, sigma, 3)
        x[100:112] = x[1000:1012]
        x[5:17] = x[1000:1012]  # ties beyond the first key: refinement rounds
        want = oracle.sa_is(x, sigma)
        rounds = {}
        # (round 4's switches: every bucket's other-region round up front / a launch set per bucket; the direct sort's
        #  first pass computing its keys / reading a key kernel's)
        for no_direct, hoist, text_keys in ((False, True, True), (False, True, False), (True, True, True), (True, False, True)):
            emu_ctx.set_no_direct_sort(no_direct)
            emu_ctx.set_induce_hoist(hoist)
            emu_ctx.set_text_keys(text_keys)
            sa, bw = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint8)
            emu_ctx.sa_bwt_build_dev(x, n, sigma, sa, bw)
            st = emu_ctx.last_stats()
            assert st["lms_path"] == (1 if no_direct else 3), (sigma, no_direct)
            assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), (sigma, no_direct, hoist, text_keys)
            rounds[(no_direct, hoist)] = st["induce_rounds"]
        assert rounds[(True, True)] < rounds[(True, False)]  # about half the rounds are queued
    emu_ctx.set_no_direct_sort(False)
    emu_ctx.set_induce_hoist(True)
    emu_ctx.set_text_keys(True)


def test_serialisation_bytes(emu_ctx, tmp_path):
    """write_complete_bwt_info, the streaming writer and read-back against the reference's byte streams"""
    from conftest import check_serialisation, serial_cases
    check_serialisation(emu_ctx.lib, serial_cases(), tmp_path)


def test_production_genome_through_the_farm(emu_ctx, golden_genomes, tmp_path):
    """the read-mapper's loop on its own 50 000-base genome (tools/readmappers/data/genomes/hg38-1000.fa) over the
    CPU execution harness: load_fasta_records -> stralg_amd_fasta_tables_batch -> write_complete_bwt_info, against
    the unmodified reference's arrays and index file"""
    from conftest import check_genomes
    check_genomes(emu_ctx.lib, golden_genomes, tmp_path, names=("hg38-1000.fa",))


def test_induce_round_batches_and_unattended_passes(emu_ctx):
    """round 3's forms of the induced-sort passes, forced onto small texts: eight self rounds of a bucket by one count /
    scan / scatter (SX_FLAG_INDUCE_BATCH_MIN), rounds queued in the three-launch form alone, the wide scatter's two steps
    a tile, buckets queued one behind the other without a wait (SX_FLAG_INDUCE_ATTENDED), and the host carrying on a
    bucket whose tail kernel left word that it could not finish it; against the oracle"""
    rng = np.random.default_rng(31)
    # (the AddressSanitizer run of this file, tests/test_emu_asan.py, takes the small cases only)
    full = os.environ.get("STRALG_EMU_ASAN") != "1"

    def check(x, sigma):
        want = oracle.sa_is_strict(x, sigma)
        sa = emu_ctx.sa_build(x, sigma)
        assert (sa == want).all(), (sigma, x.size)
        return emu_ctx.last_stats()

    try:
        for bmin in (0, 50) if full else (0,):
            emu_ctx.set_induce_batch_min(bmin)
            for sigma, n in ((2, 2049), (3, 5000), (5, 20000), (6, 5000), (8, 9000)):
                check(rng.integers(1, sigma, size=n, dtype=np.uint8), sigma)
            x = synth(12000, 5, 3)
            x[1000:1400] = 1
            x[5000:5030] = 4
            x[9000:9300] = 2
            x[9500:11900] = 3
            check(x, 5)
        emu_ctx.set_induce_batch_min(-1)
        emu_ctx.set_induce_batch(False)
        check(synth(9000, 5, 9), 5)
        emu_ctx.set_induce_batch(True)
        # wide alphabets through the induction: the radix-pass round form with every chain_max, unattended by default
        emu_ctx.set_no_direct_sort(True)
        for cm in (-1, 0, 300) if full else (0,):
            emu_ctx.set_chain_max_entries(cm)
            for sigma, n in ((9, 5000), (21, 12000), (256 if cm < 0 else 40, 6000 if cm < 0 else 12000)) if full else ((21, 3000),):
                x = rng.integers(1, sigma, size=n, dtype=np.uint8)
                x[1000:1040] = x[1000]
                st = check(x, sigma)
                assert st["induce_redo"] == 0 and st["long_runs"] == 0
        emu_ctx.set_chain_max_entries(-1)
        x = rng.integers(1, 20, size=14000, dtype=np.uint8)
        x[3000:12000] = 7  # a run that fills a whole classification tile: attended from the start
        assert check(x, 20)["long_runs"] == 1
        emu_ctx.set_no_direct_sort(False)
        # 9000 runs of 20 symbols alive in one bucket: more than the tail kernel holds after the queued rounds, so the
        # tail kernel leaves word, the launches behind it do nothing, and the host carries the bucket on (both alphabets'
        # kernels); attended, the same text needs no such thing
        unit = np.array([1] * 20 + [2, 3], np.uint8)
        x = np.tile(unit, 9000)
        x[21::22] = rng.integers(2, 5, size=9000, dtype=np.uint8)
        assert check(x, 5)["induce_redo"] >= 1
        if full:
            # more than 8 buckets: the tail kernel takes rounds of up to four tiles, so 9000 runs alive at once are its own
            # business (round 4) ...
            y = x.copy()
            y[21::22] = rng.integers(2, 30, size=9000, dtype=np.uint8)
            emu_ctx.set_no_direct_sort(True)
            assert check(y, 30)["induce_redo"] == 0
            # ... what it leaves to the host is a bucket whose runs outlast its steps (1024; the harness is built with 96):
            # 200 runs of 1 .. 200 symbols -- every round ends one of them, the all-in-a-run jump never applies
            parts = []
            for k in range(200):
                parts += [np.full(1 + k, 1, np.uint8), rng.integers(2, 30, size=3, dtype=np.uint8)]
            assert check(np.concatenate(parts), 30)["induce_redo"] >= 1
            emu_ctx.set_no_direct_sort(False)
        check(synth(9000, 5, 11), 5)
        emu_ctx.set_induce_attended(1)
        assert check(x, 5)["induce_redo"] == 0
    finally:
        emu_ctx.set_induce_attended(0)
        emu_ctx.set_induce_batch_min(-1)
        emu_ctx.set_induce_batch(True)
        emu_ctx.set_chain_max_entries(-1)
        emu_ctx.set_no_direct_sort(False)


def test_reduced_string_recursion(emu_ctx):
    """sa_is.c:370-387 where it costs nothing new: a reduced string of at most 255 names is a remapped byte text, which the
    pipeline sorts itself, level below level in child contexts (SX_FLAG_RECURSE_MIN brings the levels onto small texts):
    Fibonacci, Thue-Morse, periodic strings; the same texts through prefix doubling; the BWT and tables on top"""
    rng = np.random.default_rng(5)

    def fib(n):
        a, b = b"\x02", b"\x02\x01"
        while len(b) < n:
            a, b = b, b + a
        return np.frombuffer(b[:n], np.uint8).copy()

    tm = np.array([1], np.uint8)
    while tm.size < 20000:
        tm = np.concatenate([tm, 3 - tm])
    cases = [(fib(30000), 3), (fib(50000) + 2, 5), (np.tile(rng.integers(1, 5, size=7, dtype=np.uint8), 4000), 5),
             (np.tile(rng.integers(1, 5, size=50, dtype=np.uint8), 500), 5), (tm, 3),
             (np.tile(rng.integers(1, 40, size=9, dtype=np.uint8), 3000), 40)]
    try:
        levels = []
        for rmin in (100, 3000, -1):
            emu_ctx.set_recurse_min(rmin)
            for x, sigma in cases:
                want = oracle.sa_is_strict(x, sigma)
                sa, bw = np.zeros(x.size + 1, np.uint32), np.zeros(x.size + 1, np.uint8)
                emu_ctx.sa_bwt_build_dev(x, x.size, sigma, sa, bw)
                assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), (rmin, sigma, x.size)
                levels.append((rmin, emu_ctx.last_stats()["recursion_levels"]))
        assert max(l for r, l in levels if r == 100) >= 5 and all(l == 0 for r, l in levels if r == -1), levels
        emu_ctx.set_recurse_min(100)
        x = fib(20000)
        sa, c, o = emu_ctx.build_tables(x, 3)
        want = oracle.sa_is_strict(x, 3)
        assert (sa == want).all() and (c == oracle.c_table(x, 3)).all() and (o == oracle.o_table(x, want, 3)).all()
    finally:
        emu_ctx.set_recurse_min(-1)


def test_doubling_group_numbers_survive_splits(emu_ctx):
    """collections of near-identical sequences through the general path: groups of as many members as copies lose a
    member or two a round; the sub-group that holds the group's number keeps it (no rank is written for its members),
    the others take their middle position -- wave tier, the radix sub-list for long groups, and the plain-passes mode"""
    rng = np.random.default_rng(11)
    try:
        emu_ctx.force_general_path(True)
        seen = 0
        for sigma, L, copies, div, mode in ((5, 1500, 16, 0.001, 0), (5, 700, 39, 0.01, 0), (9, 2500, 28, 0.001, 0),
                                            (28, 380, 25, 0.001, 0), (28, 3000, 21, 0.0, 0), (5, 2000, 12, 0.01, 1),
                                            (3, 2900, 34, 0.001, 1)):
            g = rng.integers(1, sigma, size=L, dtype=np.uint8)
            parts = []
            for _ in range(copies):
                x = g.copy()
                mm = rng.random(L) < div
                x[mm] = rng.integers(1, sigma, size=int(mm.sum()), dtype=np.uint8)
                parts.append(x)
            x = np.concatenate(parts)
            emu_ctx.set_sort_mode(mode)
            sa, bw = np.zeros(x.size + 1, np.uint32), np.zeros(x.size + 1, np.uint8)
            emu_ctx.sa_bwt_build_dev(x, x.size, sigma, sa, bw)
            want = oracle.sa_is_strict(x, sigma)
            assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), (sigma, L, copies, div, mode)
            st = emu_ctx.last_stats()
            assert st["lms_path"] == 2 and st["doubling_rounds"] >= 3, st
            seen |= st["refine_tiers"]
        assert seen & 4 and seen & 8, seen  # the wave tier and the radix sorts both ordered groups
    finally:
        emu_ctx.force_general_path(False)
        emu_ctx.set_sort_mode(0)


def test_look_at_a_sample_before_the_prefix_sorts(emu_ctx):
    """texts of more than 8 symbols: the tied share of a sample under the longest 63-bit key decides whether the prefix-key
    sorts are tried at all (a word text ties most of its suffixes and goes to the general path at once; random symbols
    keep the direct sort) -- same arrays either way (SX_FLAG_SAMPLE_MIN brings the look onto small texts)"""
    rng = np.random.default_rng(9)
    words = [rng.integers(2, 28, size=int(rng.integers(2, 9)), dtype=np.uint8) for _ in range(200)]
    w = 1.0 / np.arange(1, 201) ** 1.2
    ids = rng.choice(200, size=8000, p=w / w.sum())
    x = np.concatenate([np.concatenate([words[i], [1]]) for i in ids]).astype(np.uint8)
    y = rng.integers(1, 28, size=40000, dtype=np.uint8)
    try:
        for smin, nd in ((1000, False), (1000, True), (-1, False)):
            emu_ctx.set_sample_min(smin)
            emu_ctx.set_no_direct_sort(nd)
            sa, bw = np.zeros(x.size + 1, np.uint32), np.zeros(x.size + 1, np.uint8)
            emu_ctx.sa_bwt_build_dev(x, x.size, 28, sa, bw)
            want = oracle.sa_is_strict(x, 28)
            assert (sa == want).all() and (bw == oracle.bwt(x, want)).all(), (smin, nd)
            st = emu_ctx.last_stats()
            assert st["lms_path"] == 2 and (st["sample_tied_permille"] > 300) == (smin > 0), (smin, nd, st)
            got = emu_ctx.sa_build(y, 28)
            st = emu_ctx.last_stats()
            assert (got == oracle.sa_is_strict(y, 28)).all() and st["lms_path"] == (1 if nd else 3), (smin, nd, st)
            assert st["sample_tied_permille"] <= 5, st
    finally:
        emu_ctx.set_sample_min(-1)
        emu_ctx.set_no_direct_sort(False)


def test_primitives(emu_ctx):
    rng = np.random.default_rng(1)
    n = 5000
    keys = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
    keys[::7] = keys[3]  # duplicates: stability matters
    vals = np.arange(n, dtype=np.uint32)
    ka, va, kb, vb = keys.copy(), vals.copy(), np.zeros_like(keys), np.zeros_like(vals)
    in_b = emu_ctx.prim_sort_pairs_dev(ka, va, kb, vb, n, 0, 64)
    ks, vs = (kb, vb) if in_b else (ka, va)
    order = np.argsort(keys, kind="stable")
    assert (ks == keys[order]).all() and (vs == order).all()
    # wider digits (fewer passes): 9 and 10 bits, a last pass narrower than the digit, more than one tile
    n2 = 20000
    keys2 = rng.integers(0, 1 << 40, size=n2, dtype=np.uint64)
    keys2[::5] = keys2[11]
    for bits, lo, hi in ((9, 0, 40), (10, 0, 40), (10, 3, 37), (9, 22, 40)):
        emu_ctx.set_radix_digit_bits(bits)
        try:
            ka, va = keys2.copy(), np.arange(n2, dtype=np.uint32)
            kb, vb = np.zeros_like(ka), np.zeros_like(va)
            in_b = emu_ctx.prim_sort_pairs_dev(ka, va, kb, vb, n2, lo, hi)
        finally:
            emu_ctx.set_radix_digit_bits(0)
        ks, vs = (kb, vb) if in_b else (ka, va)
        field = (keys2 >> np.uint64(lo)) & np.uint64((1 << (hi - lo)) - 1)
        order = np.argsort(field, kind="stable")
        assert (ks == keys2[order]).all() and (vs == order).all(), (bits, lo, hi)
    x = rng.integers(0, 1000, size=70000, dtype=np.uint32)
    out, tot = np.zeros_like(x), np.zeros(1, np.uint32)
    emu_ctx.prim_exclusive_sum_dev(x, out, x.size, tot)
    ref = np.concatenate(([0], np.cumsum(x, dtype=np.uint64)[:-1])).astype(np.uint32)
    assert (out == ref).all() and tot[0] == x.sum()


def test_threaded_host_layer(emu_ctx, golden, monkeypatch):
    """build_complete_table's host side in its threaded form (remap by slices, reversal, row pointers, the pager
    behind the downloads): forced onto small records through STRALG_AMD_PARALLEL_MIN / STRALG_AMD_HOST_THREADS"""
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

    lib = emu_ctx.lib
    lib.build_complete_table.argtypes = [C.c_char_p, C.c_bool]
    lib.build_complete_table.restype = C.POINTER(BT)
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    for threads, pmin in (("5", "64"), ("3", "1"), ("64", "16")):
        monkeypatch.setenv("STRALG_AMD_HOST_THREADS", threads)
        monkeypatch.setenv("STRALG_AMD_PARALLEL_MIN", pmin)
        for name in ("ref/repetitive", "ref/mississippi", "ref/fasta3"):
            c = golden[name]
            t = lib.build_complete_table(bytes(c["raw"]), True)
            N, sigma = t.contents.sa.contents.length, c["sigma"]
            assert t.contents.remap_table.contents.alphabet_size == sigma
            assert (np.ctypeslib.as_array(t.contents.sa.contents.string, shape=(N,))[:-1] == c["sym"]).all(), name
            assert (np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(N,)) == c["sa"]).all(), name
            assert (np.ctypeslib.as_array(t.contents.o_table, shape=(N + 1, sigma)) == c["o"]).all(), name
            assert (np.ctypeslib.as_array(t.contents.ro_table, shape=(N + 1, sigma)) == c["ro"]).all(), name
            base = C.addressof(t.contents.o_table.contents)
            for i in (0, 1, N // 2, N):
                assert C.addressof(t.contents.o_indices[i].contents) == base + 4 * sigma * i
            lib.completely_free_bwt_table(t)


def _c_structs():
    """stralg/suffix_array.h:10-20, remap.h:9-19, bwt.h:36-44 as ctypes structures"""
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

    return SA, RT, BT


def test_thread_exit_releases_its_context(emu_ctx, golden):
    """A caller written against the reference's API never calls stralg_amd_release(): a worker thread that builds tables
    and exits must not keep its device context (GiBs of workspace on a GPU).  The context hangs on a pthread key whose
    destructor runs at thread exit (stralg_host.c)."""
    import ctypes as C
    import threading
    SuffixArray, _, _ = _c_structs()
    lib = emu_ctx.lib
    lib.stralg_amd_live_contexts.restype = C.c_int
    lib.sa_is_construction.argtypes = [C.c_char_p, C.c_uint32]
    lib.sa_is_construction.restype = C.POINTER(SuffixArray)
    lib.free_suffix_array.argtypes = [C.POINTER(SuffixArray)]
    lib.free_suffix_array.restype = None
    c = golden["ref/mississippi"]
    text = bytes(c["sym"]) + b"\0"
    before = lib.stralg_amd_live_contexts()
    seen = []

    def work(release):
        sa = lib.sa_is_construction(text, int(c["sigma"]))
        seen.append((lib.stralg_amd_live_contexts(), np.ctypeslib.as_array(sa.contents.array, shape=(sa.contents.length,)).copy()))
        lib.free_suffix_array(sa)
        if release:
            lib.stralg_amd_release()

    for release in (False, True, False):
        th = threading.Thread(target=work, args=(release,))
        th.start()
        th.join()
        # (Thread.join returns when the thread's Python side is done; its pthread keys' destructors run a moment later)
        import time
        for _ in range(400):
            if lib.stralg_amd_live_contexts() == before:
                break
            time.sleep(0.01)
        assert lib.stralg_amd_live_contexts() == before, "the exited thread's context is still alive"
    assert all(live == before + 1 and (arr == c["sa"]).all() for live, arr in seen)


def test_farm_goes_on_past_a_record_it_cannot_build(emu_ctx, golden):
    """stralg_amd_build_tables_batch: a record with more letters than a remap table holds (stralg/remap.h:14-18) gets
    out[k] = NULL and is counted in the return value; the other records -- also the ones of the same lane -- are built."""
    import ctypes as C
    _, _, BT = _c_structs()
    lib = emu_ctx.lib
    lib.stralg_amd_build_tables_batch.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_bool, C.POINTER(C.c_int), C.c_int,
                                                  C.POINTER(C.POINTER(BT))]
    lib.stralg_amd_build_tables_batch.restype = C.c_int
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    too_many = bytes(range(1, 200)) * 3  # 199 distinct letters
    names = ["ref/mississippi", None, "ref/repetitive", "ref/fasta3", None]
    raws = [bytes(golden[k]["raw"]) if k else too_many for k in names]
    arr = (C.c_char_p * len(raws))(*raws)
    out = (C.POINTER(BT) * len(raws))()
    devs = (C.c_int * 2)(0, 0)
    assert lib.stralg_amd_build_tables_batch(arr, len(raws), True, devs, 2, out) == 2
    for k, name in enumerate(names):
        if name is None:
            assert not out[k]
            continue
        c, t = golden[name], out[k]
        N = t.contents.sa.contents.length
        assert (np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(N,)) == c["sa"]).all(), name
        assert (np.ctypeslib.as_array(t.contents.ro_table, shape=(N + 1, c["sigma"])) == c["ro"]).all(), name
        lib.completely_free_bwt_table(t)


def test_fasta_farm_reports_the_records_it_could_not_build(emu_ctx, tmp_path):
    """stralg_amd_fasta_tables_batch_ex: the number of records that could not be built comes back through
    *n_failed, their slots are NULL, the others are built from the record lengths the loader already has."""
    import ctypes as C
    _, _, BT = _c_structs()
    lib = emu_ctx.lib
    fa = tmp_path / "mixed.fa"
    wide = "".join(chr(c) for c in range(48, 123) if chr(c).isalnum()) * 3  # 62 distinct letters: fine
    fa.write_bytes(b">one\nACGTACGTTTGA\nACGT\n>two\n" + bytes(range(33, 62)) + bytes(range(63, 127)) + bytes(range(128, 200))
                   + b"\n>three\n" + wide.encode() + b"\n>four\nmississippi\n")
    lib.load_fasta_records.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.load_fasta_records.restype = C.c_void_p
    lib.free_fasta_records.argtypes = [C.c_void_p]
    lib.stralg_amd_fasta_tables_batch_ex.argtypes = [C.c_void_p, C.c_bool, C.POINTER(C.c_int), C.c_int, C.POINTER(C.POINTER(BT)),
                                                     C.POINTER(C.c_size_t)]
    lib.stralg_amd_fasta_tables_batch_ex.restype = C.c_int
    lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
    lib.completely_free_bwt_table.restype = None
    err = C.c_int(0)
    h = lib.load_fasta_records(str(fa).encode(), C.byref(err))
    assert h and err.value == 0
    out = (C.POINTER(BT) * 4)()
    devs = (C.c_int * 1)(0)
    failed = C.c_size_t(99)
    assert lib.stralg_amd_fasta_tables_batch_ex(h, False, devs, 1, out, C.byref(failed)) == 4
    assert failed.value == 1
    built = [bool(out[k]) for k in range(4)]  # iteration order = reverse file order (bioinf/fasta.c:131-134)
    assert built == [True, True, False, True]
    t = out[0]
    assert np.ctypeslib.as_array(t.contents.sa.contents.array, shape=(12,)).tolist() == [11, 10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2]
    for k in range(4):
        if out[k]:
            lib.completely_free_bwt_table(out[k])
    lib.free_fasta_records(h)


def test_reverse_on_the_device(emu_ctx):
    """sx_reverse_dev (the reversed copy build_complete_table sorts for the RO table, bwt.c:147-151): sizes around its
    16-byte pieces, an unaligned destination, the terminator, and the overlap check"""
    from stralg_amd import StralgAmdError
    rng = np.random.default_rng(8)
    for n in (0, 1, 15, 16, 17, 31, 32, 33, 1000, 4099):
        x = rng.integers(1, 6, size=n, dtype=np.uint8)
        for shift in (0, 3):
            buf = np.full(n + 1 + shift + 16, 0xEE, dtype=np.uint8)
            out = buf[shift:shift + n + 1]
            emu_ctx.reverse_dev(x, n, out)
            assert (out[:n] == x[::-1]).all() and out[n] == 0 and (buf[:shift] == 0xEE).all() and (buf[shift + n + 1:] == 0xEE).all()
    both = np.zeros(64, dtype=np.uint8)
    with pytest.raises(StralgAmdError):
        emu_ctx.reverse_dev(both[:32], 32, both[8:])


def test_host_block_cache(golden):
    """The per-thread cache of large host blocks (stralg_host.c): arrays freed through completely_free_bwt_table come back
    to the thread's next build_complete_table (the read-mapper's loop build -> write -> free -> build stops unmapping and
    first-touching its tables), a smaller cached block makes way for a larger one, and stralg_amd_release() drops the cache.
    In a child process: the threshold ($STRALG_AMD_HOST_CACHE_MIN, normally 64 MiB) is read once."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
from stralg_amd.api import Context
ctx = Context(0, lib_path=%r)
lib = ctx.lib
class SA(C.Structure):
    _fields_ = [("string", C.c_void_p), ("length", C.c_uint32), ("array", C.c_void_p), ("inverse", C.c_void_p), ("lcp", C.c_void_p)]
class BT(C.Structure):
    _fields_ = [("remap_table", C.c_void_p), ("sa", C.POINTER(SA)), ("c_table", C.c_void_p), ("o_table", C.c_void_p),
                ("o_indices", C.c_void_p), ("ro_table", C.c_void_p), ("ro_indices", C.c_void_p)]
lib.build_complete_table.argtypes = [C.c_char_p, C.c_bool]
lib.build_complete_table.restype = C.POINTER(BT)
lib.completely_free_bwt_table.argtypes = [C.POINTER(BT)]
lib.completely_free_bwt_table.restype = None
rng = np.random.default_rng(1)
def text(n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n))
def blocks(t):
    return {t.contents.o_table, t.contents.ro_table, t.contents.o_indices, t.contents.ro_indices, t.contents.sa.contents.array}
a = lib.build_complete_table(text(3000), True)
first = blocks(a)
lib.completely_free_bwt_table(a)
b = lib.build_complete_table(text(2900), True)       # a little shorter: every large array fits a cached block
assert blocks(b) <= first, "the second table's arrays are not the first one's blocks"
lib.completely_free_bwt_table(b)
big = lib.build_complete_table(text(40000), True)    # ten times longer: nothing cached fits; freed, its blocks evict the small ones
big_blocks = blocks(big)
assert not (big_blocks & first)
lib.completely_free_bwt_table(big)
again = lib.build_complete_table(text(39000), True)
assert blocks(again) <= big_blocks, "the long record's blocks were not kept"
lib.completely_free_bwt_table(again)
lib.stralg_amd_release()
lib.stralg_amd_host_cache_bytes.restype = C.c_size_t
assert lib.stralg_amd_host_cache_bytes() == 0
# the bound is one for ALL threads (ADVICE round 4): four threads build and free at once; together they never hold more
# than the cap ($STRALG_AMD_HOST_CACHE_BYTES here: about one long record's arrays), and what a thread holds goes when it exits
import threading
cap = int(__import__("os").environ["STRALG_AMD_HOST_CACHE_BYTES"])
seen = []
def work(seed):
    r = np.random.default_rng(seed)
    for _ in range(3):
        t = lib.build_complete_table(bytes(r.choice(np.frombuffer(b"ACGT", np.uint8), size=30000)), True)
        lib.completely_free_bwt_table(t)
        seen.append(lib.stralg_amd_host_cache_bytes())
ths = [threading.Thread(target=work, args=(k,)) for k in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
assert max(seen) <= cap and max(seen) > 0, (max(seen), cap)
import time
for _ in range(200):  # (Thread.join returns before the OS thread has run its key destructors)
    if lib.stralg_amd_host_cache_bytes() == 0: break
    time.sleep(0.01)
assert lib.stralg_amd_host_cache_bytes() == 0, "exited threads still hold cached blocks"
print("ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
       os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libstralg_amd_emu.so"))
    env = dict(os.environ, STRALG_AMD_HOST_CACHE_MIN="4096", STRALG_AMD_HOST_CACHE_BYTES=str(3 << 20))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
