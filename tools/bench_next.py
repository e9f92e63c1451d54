"""timings of the "next" rows: inverse + LCP, batched exact BWT search, with the oracle beside
them on a bounded sample (GPU box)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stralg_amd, oracle
ctx = stralg_amd.Context(0)
log2n, sigma = 28, 5
n = 1 << log2n; N = n + 1
text = torch.empty(n, dtype=torch.uint8, device="cuda"); ctx.synth_dev(text, n, sigma, 42)
sa = torch.empty(N, dtype=torch.int32, device="cuda"); bwt = torch.empty(N, dtype=torch.uint8, device="cuda")
ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
inv = torch.empty(N, dtype=torch.int32, device="cuda"); lcp = torch.empty(N, dtype=torch.int32, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.sa_lcp_dev(text, sa, N, inv, lcp); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"inverse + LCP, 2^{log2n} DNA: {dt*1e3:.1f} ms = {N/dt/1e6:.0f} Mpositions/s; max lcp {int(lcp.max())}")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.sa_inverse_dev(sa, N, inv); torch.cuda.synchronize(); di = time.perf_counter() - t0
# algorithmic bytes: the inverse reads sa and writes inv (8 B a position); Kasai reads inv, sa[inv - 1], the text around two
# suffixes and writes lcp[inv] (one 64-byte sector each for the three random accesses + 12 B streamed)
print(json.dumps({"row": "inverse (suffix_array.c:53-60)", "N": N, "ms": round(di * 1e3, 2), "moved_GB": round(40 * N / 1e9, 2),
                  "moved_GBps": round(40 * N / di / 1e9), "alg_GB": round(8 * N / 1e9, 2),
                  "GBps": round(8 * N / di / 1e9), "frac_of_8TBps": round(8 * N / di / 8e12, 3)}))
print(json.dumps({"row": "LCP (suffix_array.c:62-85), after the inverse", "N": N, "ms": round((dt - di) * 1e3, 2),
                  "alg_GB": round((12 + 3 * 64) * N / 1e9, 2), "GBps": round((12 + 3 * 64) * N / (dt - di) / 1e9),
                  "frac_of_8TBps": round((12 + 3 * 64) * N / (dt - di) / 8e12, 3),
                  "note": "algorithmic bytes as Kasai's loop has them: three random accesses a position (sa[j-1], the suffix's text, lcp[j]) booked as 64-byte sectors; round 5 runs it through Phi (two permutation scatters, one random access a position, a parallel first look at 32 symbols)"}))
ns = 1 << 24
xs = stralg_amd.synth(ns, sigma, 42); sas = oracle.sa_is(xs, sigma)
t0 = time.perf_counter(); oracle.lcp(xs, sas); dt = time.perf_counter() - t0
print(f"oracle compute_lcp on 2^24 symbols: {dt*1e3:.0f} ms = {ns/dt/1e6:.1f} Mpositions/s (1 core)")
c = torch.zeros(sigma, dtype=torch.int32, device="cuda"); o = torch.empty((N + 1) * sigma, dtype=torch.int32, device="cuda")
ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c, o)
q, m = 10_000_000, 30
starts = torch.randint(0, n - m, (q,), device="cuda")
idx = (starts[:, None] + torch.arange(m, device="cuda")[None, :]).reshape(-1)
pats = text[idx].contiguous(); offs = (torch.arange(q + 1, device="cuda") * m).to(torch.int32)
l = torch.zeros(q, dtype=torch.int32, device="cuda"); r = torch.zeros_like(l)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.bwt_exact_search_dev(c, o, N, sigma, pats, offs, q, l, r); torch.cuda.synchronize(); dt = time.perf_counter() - t0
hits = int(((r - l) > 0).sum())
print(f"exact search, {q} patterns x {m} symbols over 2^{log2n}: {dt*1e3:.1f} ms = {q/dt/1e6:.1f} Mpatterns/s ({q*m*2/dt/1e9:.2f} G table look-ups/s), {hits} found")
print(json.dumps({"row": "batched exact search (bwt.c:164-199)", "patterns": q, "symbols": m, "ms": round(dt * 1e3, 2),
                  "alg_GB": round(q * m * 2 * 64 / 1e9, 2), "GBps": round(q * m * 2 * 64 / dt / 1e9),
                  "frac_of_8TBps": round(q * m * 2 * 64 / dt / 8e12, 3), "note": "two O-table look-ups a symbol, a 64-byte sector each"}))
# FASTA ingest + remap on the device: a 1 GiB image (8 records, 60-column lines) from raw bytes to remapped symbols
del o, pats, idx
rng = np.random.default_rng(1)
rec_n = (1 << 27) - 4096
seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=rec_n)
full = rec_n // 60
body = np.concatenate([seq[: full * 60].reshape(full, 60), np.full((full, 1), 10, dtype=np.uint8)], axis=1).tobytes() + seq[full * 60:].tobytes() + b"\n"
data = b"".join(b">chr%d\n" % k + body for k in range(8))
d_file = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
d_packed = torch.empty(len(data) + 1, dtype=torch.uint8, device="cuda")
d_term = torch.empty(64, dtype=torch.int32, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); plen, nrec = ctx.fasta_pack_dev(d_file, len(data), d_packed, d_term, 64); torch.cuda.synchronize(); dt = time.perf_counter() - t0
term = d_term[: 2 * nrec].cpu().numpy()
s0 = int(term[0]) + 1; rn = int(term[1]) - int(term[0]) - 1
d_sym = torch.empty(rn + 1, dtype=torch.uint8, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t1 = time.perf_counter(); sg, _ = ctx.remap_dev(d_packed[s0:], rn, d_sym); torch.cuda.synchronize(); dr = time.perf_counter() - t1
print(json.dumps({"row": "FASTA pack (fasta.c:92-135)", "file_GB": round(len(data) / 1e9, 2), "ms": round(dt * 1e3, 2),
                  "alg_GB": round(3 * len(data) / 1e9, 2), "GBps": round(3 * len(data) / dt / 1e9), "frac_of_8TBps": round(3 * len(data) / dt / 8e12, 3),
                  "note": "two reads of the image (scan + counts for either entry state; write) and one write of the packed image "
                          "(round 3: three reads, 4.37 GB, 2.8 ms)"}))
print(json.dumps({"row": "remap of a record (remap.c:8-31,102-114)", "symbols": rn, "ms": round(dr * 1e3, 3),
                  "alg_GB": round(3 * rn / 1e9, 2), "GBps": round(3 * rn / dr / 1e9), "frac_of_8TBps": round(3 * rn / dr / 8e12, 3)}))
print(f"FASTA pack, {len(data)/2**30:.2f} GiB image, {nrec} records: {dt*1e3:.1f} ms = {len(data)/dt/1e9:.0f} GB/s of file; remap of one {rn/2**20:.0f} Mi record (sigma {sg}): {dr*1e3:.2f} ms")
t0 = time.perf_counter(); oracle.pyoracle.fasta_pack(data[: 1 << 26]); dt = time.perf_counter() - t0
print(f"oracle fasta packing on the first 64 MiB: {dt*1e3:.0f} ms = {(1<<26)/dt/1e9:.2f} GB/s (1 core)")
