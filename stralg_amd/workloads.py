"""Synthetic records for bench.py and the tests, generated on the device that will sort them
(torch ops; the same code runs on the CPU for the GPU-less tests).  All of them are remapped
texts as stralg's constructors take them: symbols in [1, alphabet_size), no 0.

  dna          uniform i.i.d. A C G T (BASELINE.json configs[1..2]; the splitmix64 stream of
               oracle_synth / sx_synth_dev, performance/suffix_array_construction.c:24-34's shape)
  bytes        uniform i.i.d. 1..255 (configs[3])
  genome_like  biased base composition, a family of diverged interspersed repeats (~10 % of the
               text), exact duplications of 100 ... 50 000 symbols, microsatellites and poly-A runs
  n_runs       DNA with four runs of N (the gaps of a reference assembly): 2.8 %, 1.7 %, 0.28 % and
               0.005 % of the text
  text_like    Zipf-distributed words of a 20 000-word vocabulary separated by blanks (28 symbols)
  pangenome    16 haplotypes of one random genome of n / 16 symbols, one after the other, every symbol of a copy
               replaced with probability 0.001 (what a collection of assemblies of one species looks like to an index:
               every suffix shares some thousand symbols with 15 others)
  periodic     a Fibonacci string over two symbols: every LMS substring repeats, the worst case for
               anything that tells suffixes apart by prefixes (what equal_LMS has to name,
               stralg/sa_is.c:265-292)
"""

WORKLOADS = ("dna", "bytes", "genome_like", "n_runs", "text_like", "pangenome", "periodic")


def _gen(dev, seed):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    return g


def genome_like(n, seed, dev):
    import torch
    g = _gen(dev, seed)
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    cdf = torch.tensor([0.3, 0.5, 0.7], device=dev)  # A 30 %, C 20 %, G 20 %, T 30 %
    chunk = 1 << 27
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        u = torch.rand(e - s, device=dev, generator=g)
        x[s:e] = (1 + torch.bucketize(u, cdf)).to(torch.uint8)
    span = 300
    if n > 4 * span:
        # one repeat family: copies of a 300-symbol element, each with 8 % of its symbols redrawn
        k = max(1, n // 3000)
        alu = torch.randint(1, 5, (span,), device=dev, generator=g).to(torch.uint8)
        pos = torch.randint(0, n - span - 1, (k,), device=dev, generator=g)
        for a in range(0, k, 1 << 16):
            b = min(k, a + (1 << 16))
            copies = alu.repeat(b - a, 1)
            mut = torch.rand((b - a, span), device=dev, generator=g) < 0.08
            rnd = torch.randint(1, 5, (b - a, span), device=dev, generator=g).to(torch.uint8)
            copies = torch.where(mut, rnd, copies)
            idx = pos[a:b, None] + torch.arange(span, device=dev)[None, :]
            x[idx.reshape(-1)] = copies.reshape(-1)
        # exact duplications
        for length, count in ((100, 2000), (1000, 300), (6000, 40), (50000, 2)):
            if n <= 4 * length:
                continue
            count = max(1, min(count, n // (8 * length)))
            ab = torch.randint(0, n - length - 1, (count, 2), device=dev, generator=g).cpu().tolist()
            for a, b in ab:
                x[b:b + length] = x[a:a + length].clone()
        # microsatellites and poly-A: a unit of 1 ... 4 symbols repeated over 20 ... 199 positions
        k = max(1, n // 20000)
        pos = torch.randint(0, n - 256, (k,), device=dev, generator=g)
        units = torch.randint(1, 5, (k, 4), device=dev, generator=g)
        period = torch.randint(1, 5, (k,), device=dev, generator=g)
        length = torch.randint(20, 200, (k,), device=dev, generator=g)
        col = torch.arange(200, device=dev)[None, :]
        val = torch.gather(units, 1, col % period[:, None]).to(torch.uint8)
        keep = (col < length[:, None]).reshape(-1)
        idx = (pos[:, None] + col).reshape(-1)
        x[idx[keep]] = val.reshape(-1)[keep]
    return x, 5


def n_runs(ctx, n, seed, dev):
    import torch
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.synth_dev(x, n, 5, seed)
    x[x == 4] = 5  # 1, 2, 3, 5 = A C G T; 4 = N
    for at, share in ((0.093, 0.028), (0.37, 0.0028), (0.65, 0.0168), (0.84, 0.000047)):
        s, length = int(at * n), max(1, int(share * n))
        x[s:s + length] = 4
    return x, 6


def text_like(n, seed, dev):
    import torch
    g = _gen(dev, seed)
    words = 20000
    wlen = torch.randint(2, 11, (words,), device=dev, generator=g)
    woff = torch.cumsum(wlen, 0) - wlen
    letters = torch.randint(2, 28, (int(wlen.sum()),), device=dev, generator=g).to(torch.uint8)
    weight = 1.0 / torch.arange(1, words + 1, device=dev, dtype=torch.float64) ** 1.3
    cdf = torch.cumsum(weight / weight.sum(), 0)
    k = n // 4 + 16  # words drawn: at least 3 symbols each with the blank
    ids = torch.bucketize(torch.rand(k, device=dev, generator=g, dtype=torch.float64), cdf).clamp(max=words - 1)
    span = wlen[ids] + 1
    start = torch.cumsum(span, 0) - span
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    chunk = 1 << 26
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        i = torch.arange(s, e, device=dev)
        w = torch.bucketize(i, start, right=True) - 1
        j = i - start[w]
        wid = ids[w]
        inside = j < wlen[wid]
        sym = letters[(woff[wid] + j).clamp(max=letters.numel() - 1)]
        x[s:e] = torch.where(inside, sym, torch.ones_like(sym))
    return x, 28


def pangenome(ctx, n, seed, dev, copies=16, divergence=0.001):
    import torch
    g = _gen(dev, seed)
    base_len = max(1, n // copies)
    base = torch.empty(base_len, dtype=torch.uint8, device=dev)
    ctx.synth_dev(base, base_len, 5, seed)
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    for s in range(0, n, base_len):
        e = min(n, s + base_len)
        part = base[:e - s].clone()
        changed = torch.rand(e - s, device=dev, generator=g) < divergence
        other = torch.randint(1, 5, (e - s,), device=dev, generator=g).to(torch.uint8)
        x[s:e] = torch.where(changed, other, part)
    return x, 5


def periodic(n, dev):
    import torch
    a = torch.tensor([2], dtype=torch.uint8, device=dev)
    b = torch.tensor([2, 1], dtype=torch.uint8, device=dev)
    while b.numel() < n:
        a, b = b, torch.cat((b, a))
    return b[:n].clone(), 3


def make_text(ctx, workload, n, sigma, seed, dev):
    """(text: torch uint8 [n] on dev, alphabet_size)"""
    import torch
    if workload in ("dna", "bytes", "uniform"):
        sigma = 5 if workload == "dna" else (256 if workload == "bytes" else sigma)
        x = torch.empty(n, dtype=torch.uint8, device=dev)
        ctx.synth_dev(x, n, sigma, seed)
        return x, sigma
    if workload == "genome_like":
        return genome_like(n, seed, dev)
    if workload == "n_runs":
        return n_runs(ctx, n, seed, dev)
    if workload == "text_like":
        return text_like(n, seed, dev)
    if workload == "pangenome":
        return pangenome(ctx, n, seed, dev)
    if workload == "periodic":
        return periodic(n, dev)
    raise ValueError(f"unknown workload {workload!r}")


# ---- FASTA records (BASELINE.json configs[4]) ----------------------------------------------------

def fasta_image(text, name, columns=60):
    """the bytes of a FASTA file holding one record: '>name\\n', then the sequence (symbols 1..4 as A C G T) in
    lines of `columns` letters (torch uint8 on text's device)"""
    import torch
    dev = text.device
    lut = torch.tensor([ord("N"), ord("A"), ord("C"), ord("G"), ord("T"), ord("N")], dtype=torch.uint8, device=dev)
    n = text.numel()
    rows, rest = divmod(n, columns)
    head = torch.tensor(list(b">" + name.encode() + b"\n"), dtype=torch.uint8, device=dev)
    parts = [head]
    if rows:
        body = torch.empty((rows, columns + 1), dtype=torch.uint8, device=dev)
        body[:, :columns] = lut[text[: rows * columns].long().clamp(max=5)].view(rows, columns)
        body[:, columns] = ord("\n")
        parts.append(body.view(-1))
    if rest:
        parts.append(lut[text[rows * columns:].long().clamp(max=5)])
        parts.append(torch.tensor([ord("\n")], dtype=torch.uint8, device=dev))
    return torch.cat(parts)
