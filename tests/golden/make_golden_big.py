"""Generates tests/golden/golden_big.npz: BASELINE.json configs[1]-[3] pinned to the UNMODIFIED reference's own
output at their full sizes ("SHA-256 + sampled entries for larger").

    python tests/golden/make_golden_big.py [28:5 30:5 28:256 30:256]

For every (log2 n, alphabet_size): the text is oracle.synth(2^log2n, alphabet_size, 42) - the same splitmix64
stream the GPU tests and bench.py generate on the device - and the suffix array is the reference's
sa_is_mem_construction (sa_is_mem.c:471-494; 8.2 bytes per symbol, the constructor SURVEY.md 8c names for these
sizes: sa_is_construction itself would need ~58 GiB at 2^30) from oracle/_ref/libstralg_ref.so, run in the build
container (about a minute at 2^28, ten at 2^30, one core).  Stored per case:
    sa_sha256    SHA-256 of the N = n+1 entries as little-endian u32
    sa_sampled   every 2^20-th entry (sa[0], sa[2^20], ...) and the last one
    sa_chunk_sha256   SHA-256 of every chunk of 2^26 entries (so a mismatch can be located)
    counts       occurrences of every symbol in the text + sentinel = the last O row (bwt.c:50-57 after the last
                 position) and, exclusive-prefix-summed, the C table (bwt.c:22-31); for alphabet_size <= 128
    bwt_sha256   SHA-256 of bwt[i] = text[sa[i]-1] (0 for sa[i] = 0), bwt.c:13-20, from the reference's array
    seconds      the reference's wall time on this container (one core), for BASELINE.md's table
The fixture holds inputs' seeds and the reference's outputs only.
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_big.npz")
CHUNK = 1 << 26
SEED = 42


def one(ref, log2n, sigma):
    n = 1 << log2n
    text = oracle.synth(n, sigma, SEED)
    t0 = time.time()
    sa = ref.sa_is_mem(text, sigma)
    secs = time.time() - t0
    assert sa.size == n + 1 and sa[0] == n
    out = {}
    h = hashlib.sha256()
    chunks = []
    hb = hashlib.sha256()
    for s in range(0, sa.size, CHUNK):
        piece = np.ascontiguousarray(sa[s:s + CHUNK], dtype="<u4")
        b = piece.tobytes()
        h.update(b)
        chunks.append(np.frombuffer(hashlib.sha256(b).digest(), dtype=np.uint8))
        # bwt[i] = text[sa[i]-1], the sentinel 0 where sa[i] = 0 (bwt.c:13-20)
        idx = piece.astype(np.int64) - 1
        bw = np.where(idx >= 0, text[np.maximum(idx, 0)], 0).astype(np.uint8)
        hb.update(bw.tobytes())
    key = f"n{log2n}/s{sigma}"
    out[key + "/sa_sha256"] = np.frombuffer(h.digest(), dtype=np.uint8)
    out[key + "/sa_chunk_sha256"] = np.stack(chunks)
    out[key + "/bwt_sha256"] = np.frombuffer(hb.digest(), dtype=np.uint8)
    out[key + "/sa_sampled"] = np.concatenate([sa[:: 1 << 20], sa[-1:]]).astype(np.uint32)
    counts = np.bincount(text, minlength=sigma).astype(np.uint64)
    counts[0] += 1  # the sentinel
    out[key + "/counts"] = counts
    out[key + "/seconds"] = np.array([secs])
    out[key + "/seed"] = np.array([SEED], dtype=np.uint64)
    print(f"{key}: reference sa_is_mem_construction {secs:.1f} s = {(n + 1) / secs / 1e6:.2f} Msuffixes/s, "
          f"sha256 {h.hexdigest()[:16]}...", flush=True)
    return out


def main():
    oracle.build(ref=True)
    ref = oracle.ref()
    wanted = sys.argv[1:] or ["28:5", "28:256", "30:5", "30:256"]
    have = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for w in wanted:
        log2n, sigma = (int(v) for v in w.split(":"))
        have.update(one(ref, log2n, sigma))
        np.savez_compressed(OUT, **have)
    print(f"{len(have)} arrays -> {OUT}")


if __name__ == "__main__":
    main()
