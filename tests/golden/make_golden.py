"""Generates tests/golden/golden.npz by running the UNMODIFIED reference
(oracle/_ref/libstralg_ref.so, built from /root/reference by oracle/Makefile) in
the build container.  The fixture holds inputs and the reference's outputs only.

    python tests/golden/make_golden.py

Cases:
  * the strings of the reference's own tests for this path
    (tests/stralg/suffix_array_test.c:11-32, bwt_test.c:16-70,
     match_test.c:682-696, serialise_test.c:15, test-data/*.txt, bioinf ref.fa)
  * seeded random strings, sigma in {2,3,5,21,128,256}, n up to 64 Ki
  * structured strings (runs, periodic, Fibonacci, monotone)
Also written: golden_next.npz (inverse, LCP, exact-search intervals), golden_fasta.npz (FASTA loader, index files)
and golden_genomes.npz (the production caller's genomes tools/readmappers/data/genomes/hg38-1000.fa / hg38-10000.fa:
file, packed image, and per record C, the SHA-256 and sampled rows of SA, O, RO, and the SHA-256 of the reference's index file).
For every case: remapped symbols, alphabet_size, SA (sa_is_construction; checked
equal to sa_is_mem / skew / qsort inside this script), and for sigma <= 128 and
n <= 4096 the C, O and RO tables of build_complete_table.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF_TESTS = "/root/reference/tests"


def main():
    oracle.build(ref=True)
    ref = oracle.ref()
    cases = {}

    def add(name, raw, tables=True, pre_remapped_sigma=None):
        raw = np.frombuffer(bytes(raw), dtype=np.uint8) if isinstance(raw, (bytes, bytearray)) else np.asarray(raw, np.uint8)
        if pre_remapped_sigma is None:
            sym, sigma = ref.remap_string(raw)
        else:
            sym, sigma = raw, pre_remapped_sigma
        sa = ref.sa_is(sym, sigma)
        assert (ref.sa_is_mem(sym, sigma) == sa).all(), name
        if sym.size and sigma > 2:
            assert (ref.skew(sym) == sa).all(), name
        if sym.size <= 5000:
            assert (ref.qsort(sym) == sa).all(), name
        cases[name + "/sym"] = sym
        cases[name + "/sigma"] = np.array([sigma], np.uint32)
        cases[name + "/sa"] = sa
        if tables and pre_remapped_sigma is None and sigma <= 128 and 0 < raw.size <= 4096:
            t = ref.build_complete_table(raw, True)
            assert (t["sa"] == sa).all() and t["sigma"] == sigma
            cases[name + "/raw"] = raw
            cases[name + "/c"] = t["c"]
            cases[name + "/o"] = t["o"]
            cases[name + "/ro"] = t["ro"]

    # the reference's own test strings
    add("ref/ababacabac", b"ababacabac")
    add("ref/mississippi", b"mississippi")
    add("ref/serialise", b"acgtadtadadfasdfing")
    for i, s in enumerate([b"acacacg", b"gacacacag", b"acacacag", b"acagcaca", b"acatgaca", b"acgc", b"ccgc",
                           b"aaaaaaaaa"]):
        add(f"ref/match{i}", s)
    add("ref/modest-proposal", open(f"{REF_TESTS}/stralg/test-data/modest-proposal.txt", "rb").read().replace(b"\0", b""))
    add("ref/repetitive", open(f"{REF_TESTS}/stralg/test-data/repetitive-string.txt", "rb").read().strip())
    seqs, cur = [], []
    for line in open(f"{REF_TESTS}/bioinf/test-data/ref.fa", "rb").read().splitlines():
        if line.startswith(b">"):
            if cur:
                seqs.append(b"".join(cur))
            cur = []
        elif line.strip():
            cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    for i, s in enumerate(seqs):
        add(f"ref/fasta{i}", s)

    # seeded random strings, already in remapped form (symbols 1..sigma-1)
    rng = np.random.default_rng(20261003)
    for sigma in (2, 3, 5, 21, 128, 256):
        for n in (0, 1, 2, 3, 10, 1000):
            if sigma == n + 1:
                continue  # sort_SA's shortcut needs distinct symbols
            add(f"rand/s{sigma}/n{n}", rng.integers(1, sigma, size=n, dtype=np.uint8), pre_remapped_sigma=sigma)
    for sigma in (5, 256):
        add(f"rand/s{sigma}/n65536", oracle.synth(65536, sigma, 42), pre_remapped_sigma=sigma)

    # structured strings
    add("struct/all-a", np.full(3000, ord("a"), np.uint8))
    add("struct/ab", np.tile(np.frombuffer(b"ab", np.uint8), 2000))
    add("struct/aab", np.tile(np.frombuffer(b"aab", np.uint8), 1300))
    add("struct/runs", np.repeat(rng.integers(97, 101, size=120, dtype=np.uint8), 31))
    add("struct/akb-akb", np.frombuffer(b"a" * 1500 + b"b" + b"a" * 1500 + b"b", np.uint8))
    add("struct/b-ak", np.frombuffer(b"b" + b"a" * 3000, np.uint8))
    add("struct/increasing", np.arange(1, 121, dtype=np.uint8))
    add("struct/decreasing", np.arange(120, 0, -1, dtype=np.uint8))
    a, b = b"a", b"ab"
    while len(b) < 4000:
        a, b = b, b + a
    add("struct/fibonacci", np.frombuffer(b, np.uint8))
    add("struct/periodic", np.tile(rng.integers(97, 101, size=50, dtype=np.uint8), 80))
    add("struct/tile-runs", np.concatenate([np.full(4096, 2, np.uint8), np.full(4100, 2, np.uint8), [1],
                                            np.full(8200, 3, np.uint8), [4]]).astype(np.uint8),
        pre_remapped_sigma=5)

    # "next" rows: inverse + LCP (suffix_array.c:53-85) and exact BWT search intervals
    # (bwt.c:164-199) from the reference, for a subset of the cases above
    nxt = {}
    for name in ("ref/ababacabac", "ref/mississippi", "ref/serialise", "ref/modest-proposal", "ref/repetitive",
                 "struct/fibonacci", "struct/periodic", "struct/all-a", "rand/s5/n65536", "rand/s21/n1000"):
        sym, sigma = cases[name + "/sym"], int(cases[name + "/sigma"][0])
        sa, inv, lcp = ref.lcp(sym, sigma)
        assert (sa == cases[name + "/sa"]).all()
        nxt[name + "/inverse"] = inv
        nxt[name + "/lcp"] = lcp
    prng = np.random.default_rng(7)
    for name in ("ref/mississippi", "ref/serialise", "ref/fasta0", "ref/fasta3", "struct/periodic", "ref/repetitive"):
        raw = bytes(cases[name + "/raw"])
        pats = [raw[i:i + L] for L in (1, 2, 3, 5, 9, 17) for i in prng.integers(0, max(1, len(raw) - L), size=6)
                if 0 < len(raw[i:i + L])]
        letters = sorted(set(raw))
        pats += [bytes(prng.choice(letters, size=L).astype(np.uint8)) for L in (2, 4, 7, 12) for _ in range(6)]
        pats += [raw, raw + raw[:1]]  # the whole text, and a pattern longer than the text
        res = ref.exact_search(raw, pats)
        flat = np.concatenate([r[2] for r in res]).astype(np.uint8)
        offs = np.concatenate(([0], np.cumsum([r[2].size for r in res]))).astype(np.uint32)
        nxt[name + "/patterns"] = flat
        nxt[name + "/offsets"] = offs
        nxt[name + "/lr"] = np.array([[r[0], r[1]] for r in res], dtype=np.uint32)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_next.npz"), **nxt)
    print(f"{len(nxt)} arrays -> golden_next.npz")

    # ---- FASTA ingest: the reference's load_fasta_records on its own test data
    # (tests/bioinf/test-data/ref.fa, malformed.fa: fasta_test.c:13-79) and on small edge cases
    fa = {}
    fasta_inputs = {
        "ref/ref.fa": open(os.path.join(REF_TESTS, "bioinf", "test-data", "ref.fa"), "rb").read(),
        "ref/malformed.fa": open(os.path.join(REF_TESTS, "bioinf", "test-data", "malformed.fa"), "rb").read(),
        "edge/empty": b"", "edge/only-gt": b">", "edge/header-only": b">a\n", "edge/no-final-newline": b">a\nAC",
        "edge/no-gt-first-line": b"ACGT\nAC\n", "edge/spaces-and-gt-in-header": b"> x y >z\tw\nAC GT\n",
        "edge/gt-mid-line": b">a\nAC>b\nGT\n", "edge/crlf": b">a\r\nAC\r\nGT\r\n>b\r\nTT",
        "edge/empty-sequences": b">a\n>b\n\n\n>c\n \t\n", "edge/nul-inside": b">a\nAC\0GT\n>b\nTT\n",
        "edge/ends-in-header": b">a\nACGT\n>b", "edge/blank-lines": b"\n\n>a\n\nAC\n\n\nGT\n\n",
    }
    frng = np.random.default_rng(11)
    letters = np.frombuffer(b">> \t\n\n\rACGTNacgt xy", dtype=np.uint8)
    for k in range(24):
        fasta_inputs[f"rand/{k}"] = bytes(frng.choice(letters, size=int(frng.integers(0, 200))))
    lines = []
    for k in range(7):  # a well-formed multi-record file with 60-column lines
        seq = bytes(frng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(frng.integers(1, 5000))))
        lines.append(b">chr%d some description\n" % k + b"\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + b"\n")
    fasta_inputs["struct/seven-records"] = b"".join(lines)
    for name, data in fasta_inputs.items():
        err, recs = ref.fasta(data)
        fa[name + "/file"] = np.frombuffer(data, dtype=np.uint8)
        fa[name + "/err"] = np.array([err], dtype=np.int32)  # stralg/error.h: 0 NO_ERROR, 2 MALFORMED_FILE
        # the packed image in file order (the iterator yields the records in reverse, fasta.c:131-134)
        fa[name + "/packed"] = np.frombuffer(b"".join(n + b"\0" + q + b"\0" for n, q in recs[::-1]), dtype=np.uint8)
        fa[name + "/records"] = np.array([len(recs)], dtype=np.uint32)
    # ---- index serialisation (section 8f row 1): the reference's write_complete_bwt_info byte stream
    for name in ("ref/mississippi", "ref/serialise", "ref/fasta0", "struct/periodic"):
        raw = bytes(cases[name + "/raw"])
        fa["serial/" + name.replace("/", "-") + "/raw"] = np.frombuffer(raw, dtype=np.uint8)
        fa["serial/" + name.replace("/", "-") + "/with_reverse"] = np.frombuffer(ref.serialise(raw, True), dtype=np.uint8)
        fa["serial/" + name.replace("/", "-") + "/forward_only"] = np.frombuffer(ref.serialise(raw, False), dtype=np.uint8)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_fasta.npz"), **fa)
    print(f"{len(fasta_inputs)} FASTA cases -> golden_fasta.npz")

    # ---- the production caller's own inputs (tools/readmappers/bwt_readmapper/bwt_readmapper.c:54-62 is run by the
    # reference's evaluation scripts on tools/readmappers/data/genomes/hg38-*.fa): the data files, what
    # load_fasta_records makes of them, and per record everything build_complete_table + write_complete_bwt_info hand over
    import hashlib
    gen = {}
    GENOMES = "/root/reference/tools/readmappers/data/genomes"
    for fname in ("hg38-1000.fa", "hg38-10000.fa"):
        data = open(os.path.join(GENOMES, fname), "rb").read()
        err, recs = ref.fasta(data)
        assert err == 0 and len(recs) == 1
        gen[fname + "/file"] = np.frombuffer(data, dtype=np.uint8)
        gen[fname + "/err"] = np.array([err], dtype=np.int32)
        gen[fname + "/packed"] = np.frombuffer(b"".join(n + b"\0" + q + b"\0" for n, q in recs[::-1]), dtype=np.uint8)
        gen[fname + "/records"] = np.array([len(recs)], dtype=np.uint32)
        for k, (name, seq) in enumerate(recs):  # iteration order
            t = ref.build_complete_table(seq, True)
            sym, sigma = ref.remap_string(np.frombuffer(seq, dtype=np.uint8))
            assert sigma == t["sigma"] and (ref.sa_is(sym, sigma) == t["sa"]).all()
            key = f"{fname}/rec{k}"
            gen[key + "/name"] = np.frombuffer(name, dtype=np.uint8)
            gen[key + "/sym"] = sym
            gen[key + "/sigma"] = np.array([sigma], np.uint32)
            gen[key + "/c"] = t["c"]
            # 50 000 and 500 000 bases: the arrays as SHA-256 of their little-endian bytes, plus every 997th row in full
            for field in ("sa", "o", "ro"):
                a = np.ascontiguousarray(t[field], dtype=np.uint32)
                gen[f"{key}/{field}_sha256"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
                gen[f"{key}/{field}_rows"] = a[::997].copy()
            for rev, tag in ((True, "with_reverse"), (False, "forward_only")):
                blob = ref.serialise(seq, rev)
                gen[f"{key}/serial_{tag}_sha256"] = np.frombuffer(hashlib.sha256(blob).digest(), dtype=np.uint8)
                gen[f"{key}/serial_{tag}_len"] = np.array([len(blob)], dtype=np.uint64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_genomes.npz"), **gen)
    print(f"{len(gen)} arrays -> golden_genomes.npz")

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz")
    np.savez_compressed(out, **cases)
    names = sorted({k.rsplit("/", 1)[0] for k in cases})
    print(f"{len(names)} cases -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
