"""The oracle (oracle/oracle.c) against the golden vectors and, where the
unmodified reference has been built (build container only), against the
reference itself.  CPU only."""
import numpy as np
import pytest

import oracle
from stralg_amd.synth import synth


def test_known_answers():
    # tests/stralg/suffix_array_test.c:19-32
    sym, sigma, _ = oracle.remap(b"ababacabac")
    assert oracle.sa_is(sym, sigma).tolist() == [10, 0, 6, 2, 8, 4, 1, 7, 3, 9, 5]
    # tests/stralg/bwt_test.c:16-19 and SURVEY.md 8c
    sym, sigma, _ = oracle.remap(b"mississippi")
    sa = oracle.sa_is(sym, sigma)
    assert sa.tolist() == [11, 10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2]
    assert oracle.c_table(sym, sigma).tolist() == [0, 1, 5, 6, 8]
    assert oracle.bwt(sym, sa).tolist() == [1, 3, 4, 4, 2, 0, 3, 1, 4, 4, 1, 1]
    # bwt_test.c:32-38 (letter-major literal, transposed here)
    o = oracle.o_table(sym, sa, sigma)
    expected = np.array([
        [0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1],
        [0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 3, 4],
        [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1],
        [0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2],
        [0, 0, 0, 1, 2, 2, 2, 2, 2, 3, 4, 4, 4]], dtype=np.uint32)
    assert (o.T == expected).all()


def test_golden_vectors(golden):
    for name, c in golden.items():
        sa = oracle.sa_is(c["sym"], c["sigma"])
        assert (sa == c["sa"]).all(), name
        assert oracle.check_sa(c["sym"], sa), name
        if "o" in c:
            sym, sigma, _ = oracle.remap(c["raw"])
            assert sigma == c["sigma"] and (sym == c["sym"]).all(), name
            assert (oracle.c_table(sym, sigma) == c["c"]).all(), name
            assert (oracle.o_table(sym, sa, sigma) == c["o"]).all(), name
            rsym = sym[::-1].copy()
            rsa = oracle.sa_is(rsym, sigma)
            assert (oracle.o_table(rsym, rsa, sigma) == c["ro"]).all(), name


def test_next_rows_golden(golden):
    """inverse / LCP (suffix_array.c:53-85) and exact-search intervals (bwt.c:164-199) from the reference"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_next.npz"))
    names = sorted({k.rsplit("/", 1)[0] for k in z.files})
    n_lcp = n_search = 0
    for name in names:
        c = golden[name]
        if name + "/lcp" in z.files:
            assert (oracle.inverse(c["sa"]) == z[name + "/inverse"]).all(), name
            assert (oracle.lcp(c["sym"], c["sa"]) == z[name + "/lcp"]).all(), name
            n_lcp += 1
        if name + "/lr" in z.files:
            pats, offs, lr = z[name + "/patterns"], z[name + "/offsets"], z[name + "/lr"]
            for q in range(lr.shape[0]):
                got = oracle.bwt_exact_search(c["c"], c["o"], c["sigma"], pats[offs[q]: offs[q + 1]])
                assert got == (int(lr[q, 0]), int(lr[q, 1])), (name, q)
            n_search += lr.shape[0]
    assert n_lcp >= 8 and n_search > 200


def test_fasta_packing_golden(golden_fasta):
    """oracle_fasta_pack against the reference's load_fasta_records on its own test files and edge cases"""
    from oracle import pyoracle
    for name, c in golden_fasta.items():
        bad, packed, recs = pyoracle.fasta_pack(c["file"])
        assert bad == (c["err"] == 2), name
        if not bad:
            assert packed == c["packed"] and len(recs) == c["records"], name
    assert golden_fasta["ref/ref.fa"]["records"] == 5 and golden_fasta["ref/malformed.fa"]["err"] == 2  # fasta_test.c:53,76


def test_production_genomes_golden(golden_genomes):
    """the read-mapper's own genomes (tools/readmappers/data/genomes/hg38-1000.fa, hg38-10000.fa; 50 000 and 500 000
    bases): the oracle's FASTA packing, remap, suffix array and C / O / RO tables against what the unmodified
    reference's load_fasta_records + build_complete_table produced"""
    from conftest import check_against_sha
    from oracle import pyoracle
    for fname, g in golden_genomes.items():
        bad, packed, recs = pyoracle.fasta_pack(g["file"])
        assert not bad and g["err"] == 0 and packed == g["packed"] and len(recs) == g["records"] == 1, fname
        for (name, seq), want in zip(recs[::-1], g["recs"]):  # iteration order = reverse file order
            assert name == want["name"], fname
            sym, sigma, _ = oracle.remap(seq)
            assert sigma == want["sigma"] == 5 and (sym == want["sym"]).all(), fname
            sa = oracle.sa_is(sym, sigma)
            check_against_sha(sa, want, "sa", fname)
            assert (oracle.c_table(sym, sigma) == want["c"]).all(), fname
            check_against_sha(oracle.o_table(sym, sa, sigma), want, "o", fname)
            rsym = sym[::-1].copy()
            check_against_sha(oracle.o_table(rsym, oracle.sa_is(rsym, sigma), sigma), want, "ro", fname)


def test_naive_agrees():
    rng = np.random.default_rng(5)
    for sigma in (2, 4, 17):
        for n in (0, 1, 5, 64, 700):
            x = rng.integers(1, sigma, size=n, dtype=np.uint8)
            assert (oracle.sa_naive(x) == oracle.sa_is_strict(x, sigma)).all()


def test_shortcut_quirk():
    # alphabet_size == n + 1 with repeated symbols
    x = np.array([1, 1, 2, 3], dtype=np.uint8)
    got = oracle.sa_is(x, 5)                                        # what sa_is.c:423-428 yields:
    assert got[:4].tolist() == [4, 1, 2, 3]                         # SA[x[i]] = i, last slot never written
    assert not oracle.check_sa(x, got)
    assert oracle.sa_is_strict(x, 5).tolist() == [4, 0, 1, 2, 3]   # the suffix array


def test_synth_streams_agree():
    for sigma in (5, 256):
        assert (oracle.synth(10000, sigma, 42) == synth(10000, sigma, 42)).all()
    assert (synth(100, 5, 7, start=50) == synth(150, 5, 7)[50:]).all()


@pytest.mark.skipif(not oracle.have_ref(), reason="reference library not built here")
def test_fasta_against_reference_build():
    from oracle import pyoracle
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref not built (no reference checkout)")
    ref = pyoracle._Ref()
    rng = np.random.default_rng(3)
    letters = np.frombuffer(b">> \t\n\n\rACGTNacgt xy", dtype=np.uint8)
    for k in range(400):
        data = bytes(rng.choice(letters, size=int(rng.integers(0, 120))))
        err, recs = ref.fasta(data)
        bad, _, mine = pyoracle.fasta_pack(data)
        assert bad == (err == 2), data
        if not bad:
            assert recs[::-1] == mine, data


def test_against_reference_build():
    ref = oracle.ref()
    rng = np.random.default_rng(11)
    for sigma in (2, 5, 20, 128, 256):
        for n in (1, 2, 7, 100, 5000):
            if sigma == n + 1:
                continue
            x = rng.integers(1, sigma, size=n, dtype=np.uint8)
            assert (ref.sa_is(x, sigma) == oracle.sa_is(x, sigma)).all()
    raw = rng.integers(97, 103, size=3000, dtype=np.uint8)
    t = ref.build_complete_table(raw, True)
    sym, sigma, _ = oracle.remap(raw)
    sa = oracle.sa_is(sym, sigma)
    assert (sa == t["sa"]).all()
    assert (oracle.c_table(sym, sigma) == t["c"]).all()
    assert (oracle.o_table(sym, sa, sigma) == t["o"]).all()


def test_big_fixture_is_well_formed():
    """tests/golden/golden_big.npz (make_golden_big.py: the unmodified reference's sa_is_mem_construction on
    synth(2^28 | 2^30, 5 | 256, 42)): every case holds the hashes and samples the GPU suite compares at full size, and its
    counts are those of the text the seed generates (checked here on the text's first 2^20 symbols' generator and on sums)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_big.npz"))
    for log2n in (28, 30):
        for sigma in (5, 256):
            key = f"n{log2n}/s{sigma}"
            n = 1 << log2n
            assert z[key + "/sa_sha256"].size == 32 and z[key + "/bwt_sha256"].size == 32
            assert z[key + "/sa_chunk_sha256"].shape == (-(-(n + 1) // (1 << 26)), 32)
            sampled = z[key + "/sa_sampled"]
            assert sampled.size == (n >> 20) + 2 and sampled[0] == n  # sa[0] is the sentinel suffix (sa_is.c:463)
            assert sampled[-1] == sampled[-2] and len(set(sampled.tolist())) == sampled.size - 1  # (N - 1 is a multiple of 2^20: the last entry twice)
            counts = z[key + "/counts"]
            assert counts.size == sigma and counts[0] == 1 and int(counts.sum()) == n + 1
            assert (counts[1:] > 0).all() and int(z[key + "/seed"][0]) == 42
            assert float(z[key + "/seconds"][0]) > 10  # the reference's own wall time on the build container
