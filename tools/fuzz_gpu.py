"""differential fuzz on the GPU: random sizes, alphabets and repeat structures against the oracle (SA, BWT, C, O)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import stralg_amd, oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = stralg_amd.Context(0)
t0 = time.time()
paths = {}
for k in range(cases):
    sigma = int(rng.choice([2, 3, 4, 5, 5, 5, 6, 7, 8, 9, 16, 17, 21, 32, 33, 64, 65, 127, 128, 200, 256]))
    n = int(rng.choice([1, 2, 3, 17, 255, 1023, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8193, 16385, 65537, 100003, 300007,
                        1048577, 2500001, 2500001, 9000001]))
    n = max(1, n + int(rng.integers(-3, 4)))
    kind = int(rng.integers(0, 7))
    if sigma == 2:
        x = np.ones(n, dtype=np.uint8)
    else:
        x = rng.integers(1, sigma, size=n, dtype=np.uint8)
    if kind == 1 and n > 64:      # planted repeats
        for _ in range(int(rng.integers(1, 20))):
            L = int(rng.integers(2, max(3, n // 4)))
            a, b = rng.integers(0, n - L, size=2)
            x[b:b + L] = x[a:a + L]
    elif kind == 2 and n > 8:     # runs
        for _ in range(int(rng.integers(1, 30))):
            L = int(rng.integers(1, max(2, n // 8)))
            a = int(rng.integers(0, n - L))
            x[a:a + L] = x[a]
    elif kind == 3:               # periodic
        p = int(rng.integers(1, 9))
        x = np.resize(x[:p], n)
    elif kind == 6 and n > 4096:  # families of diverged repeats: groups of tens to thousands of tied suffixes
        for _ in range(int(rng.integers(1, 4))):
            L = int(rng.integers(8, 200))
            copies = int(rng.integers(10, max(11, min(12000, n // (3 * L)))))
            el = x[:L].copy()
            div = float(rng.choice([0.0, 0.002, 0.02, 0.08]))
            for pos in rng.choice(n // L - 1, size=copies, replace=False) * L:
                c = el.copy()
                mm = rng.random(L) < div
                c[mm] = rng.integers(1, sigma, size=int(mm.sum()), dtype=np.uint8) if sigma > 2 else 1
                x[pos:pos + L] = c
    elif kind == 4 and sigma > 3: # skewed
        x = np.where(rng.random(n) < 0.9, 1, x).astype(np.uint8)
    flag = int(rng.integers(0, 4))
    ctx.set_chain_max_entries(int(rng.choice([-1, -1, 0, 2048, 8192, 65536, 524288, 4194304])))  # which induce rounds are chained
    ctx.set_sort_mode(int(rng.choice([0, 1, 2, 2, 3])))  # LSD passes / top bits in HBM passes + sub-buckets in LDS (where the key shape allows)
    ctx.set_radix_digit_bits(int(rng.choice([0, 0, 0, 9, 10])))
    ctx.force_general_path(flag == 1)
    ctx.set_no_direct_sort(flag == 2)
    ctx.set_prefix_symbols(int(rng.choice([0, 0, 0, 12, 14, 15, 16, 17, 18, 19, 23])))  # the key kernel's static forms too
    # round 3's forms of the induced-sort passes: eight rounds at a time from ranges of any length on, every self round a
    # launch of its own, the host looking at every bucket or queuing them one behind the other, the text copied first or
    # by the classification
    ctx.set_induce_batch_min(int(rng.choice([-1, -1, 0, 100, 3000])))
    ctx.set_induce_batch(bool(rng.integers(0, 5)))
    ctx.set_induce_attended(int(rng.choice([0, 0, 1])))
    ctx.set_copy_text_first(bool(rng.integers(0, 4) == 0))
    ctx.set_recurse_min(int(rng.choice([-1, -1, 50, 3000])))  # reduced strings of few names sorted by the pipeline itself
    ctx.set_sample_min(int(rng.choice([-1, -1, 2000, 100000])))  # the look at a sample before the prefix sorts (sigma > 8)
    # round 4: more than 8 buckets -- every bucket's other-region round up front (bigram counts) or a launch set of its own;
    # the direct sort's first pass computing its keys from the text or reading a key kernel's
    ctx.set_induce_hoist(bool(rng.integers(0, 4)))
    ctx.set_text_keys(bool(rng.integers(0, 4)))
    ctx.set_small_direct_max(int(rng.choice([0, 0, -1, 1 << 30])))  # short records of few symbols sorted directly, or through SA-IS
    ctx.set_long_subbuckets(bool(rng.integers(0, 4)))  # (the hybrid sort lists sub-buckets too long for a workgroup, or falls back)
    ctx.set_local_sort_lean(bool(rng.integers(0, 4)))  # round 5: the LDS step's lean kernel (+ the other one for crowded workgroups), or the other one for all
    # (alphabet_size == n + 1 with repeated symbols: the reference's shortcut leaves garbage)
    want = oracle.sa_is_strict(x, sigma) if sigma == n + 1 else oracle.sa_is(x, sigma)
    sa = np.zeros(n + 1, np.uint32)
    got = ctx.sa_build(x, sigma)
    st = ctx.last_stats()
    pk = (st["lms_path"], st["sort_local"], st["refine_tiers"], min(st["induce_redo"], 1), st["long_runs"], min(st["recursion_levels"], 2))
    paths[pk] = paths.get(pk, 0) + 1
    assert (got == want).all(), ("SA", k, sigma, n, kind, flag)
    if sigma <= 128 and n < 70000:
        want_c, want_o = oracle.c_table(x, sigma), oracle.o_table(x, want, sigma).ravel()
        c, o = ctx.bwt_tables(x, want, sigma)
        assert (c == want_c).all(), ("C", k, sigma, n, kind)
        assert (o.ravel() == want_o).all(), ("O", k, sigma, n, kind)
        sa2, c2, o2 = ctx.build_tables(x, sigma)  # the fused build: BWT from the induction windows / the sort payload
        assert (sa2 == want).all() and (c2 == want_c).all() and (o2.ravel() == want_o).all(), ("fused", k, sigma, n, kind, flag)
    if k % 1000 == 999: print(f"{k + 1} cases, {time.time() - t0:.0f} s", flush=True)  # (a run that stays silent for minutes is taken for hung)
ctx.force_general_path(False); ctx.set_no_direct_sort(False); ctx.set_chain_max_entries(-1); ctx.set_prefix_symbols(0)
ctx.set_sort_mode(0); ctx.set_radix_digit_bits(0)
ctx.set_induce_batch_min(-1); ctx.set_induce_batch(True); ctx.set_induce_attended(0); ctx.set_copy_text_first(False); ctx.set_recurse_min(-1); ctx.set_sample_min(-1)
ctx.set_induce_hoist(True); ctx.set_text_keys(True); ctx.set_long_subbuckets(True); ctx.set_small_direct_max(-1); ctx.set_local_sort_lean(True)
print(f"{cases} cases ok in {time.time()-t0:.0f} s; paths taken: {paths}")
