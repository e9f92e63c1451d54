"""Batch farm over GPUs: independent records, one process (or host thread) per GPU,
no collective on the data path (the per-record loop is
tools/readmappers/bwt_readmapper/bwt_readmapper.c:54-62).

torch.distributed is used only to bracket timed regions (barrier) and to reduce
the bookkeeping scalars (max time over ranks, total suffixes)."""
import time


def lpt_assign(lengths, world):
    """Longest-processing-time-first assignment of records to ranks.
    Returns a list (per rank) of record indices; ties keep file order."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return out


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except ImportError:
        pass
    return None


def fence(cuda=True):
    """barrier + device sync on both sides, as bench.py's timing contract requires."""
    dist = _dist()
    if cuda:
        import torch
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if cuda:
        import torch
        torch.cuda.synchronize()


def reduce_scalars(elapsed, units, device=None):
    """(max elapsed over ranks, sum of units over ranks)."""
    dist = _dist()
    if dist is None:
        return elapsed, units
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(u.item()))


def gather_ints(value):
    """[value of rank 0, value of rank 1, ...] (bookkeeping only)"""
    dist = _dist()
    if dist is None:
        return [int(value)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, int(value))
    return [int(v) for v in out]


def timed(step, steps, warmup, cuda=True):
    """warmup untimed steps, then exactly `steps` timed ones between fences."""
    for _ in range(warmup):
        step()
    fence(cuda)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence(cuda)
    return time.perf_counter() - t0


class FastaRecordJob:
    """One record of the loop at tools/readmappers/bwt_readmapper/bwt_readmapper.c:54-62, from the bytes
    of a FASTA file to the record's suffix array, BWT and C/O tables, everything on one device:

        file image (host) --H2D--> image in HBM -> sx_fasta_pack_dev (bioinf/fasta.c:92-135)
        -> sx_remap_dev (remap.c:8-31,102-114) -> sx_sa_bwt_build_dev (sa_is.c:466-509)
        -> sx_bwt_tables_from_bwt_dev (bwt.c:35-65)

    `dev` is a torch device: cuda:k with the product library, cpu with the CPU execution harness
    of the kernels (tests).  The buffers are allocated once and reused by every call."""

    def __init__(self, ctx, image_host, dev, tables=True):
        import torch
        self.ctx, self.dev, self.tables = ctx, dev, tables
        self.h_file = image_host  # torch uint8 on the CPU (pinned when a GPU is used)
        flen = int(image_host.numel())
        self.file_len = flen
        self.d_file = torch.empty(flen, dtype=torch.uint8, device=dev)
        self.d_packed = torch.empty(flen + 1, dtype=torch.uint8, device=dev)
        self.d_term = torch.zeros(16, dtype=torch.int32, device=dev)
        self.d_text = torch.empty(flen + 1, dtype=torch.uint8, device=dev)
        self.n = self.sigma = None
        self.sa = self.bwt = self.c = self.o = None

    def upload(self):
        """the H2D copy of the file image (what load_fasta_records' fread is to the reference)"""
        self.d_file.copy_(self.h_file)
        if self.dev.type == "cuda":
            import torch
            torch.cuda.synchronize(self.dev)

    def build(self):
        """image in HBM -> tables; returns the number of suffixes built (n + 1)"""
        import torch
        ctx = self.ctx
        _, nrec = ctx.fasta_pack_dev(self.d_file, self.file_len, self.d_packed, self.d_term, 16)
        if nrec != 1:
            raise RuntimeError(f"expected one FASTA record, found {nrec}")
        term = self.d_term[:2].cpu().tolist()
        seq0, n = term[0] + 1, term[1] - term[0] - 1
        sigma, _ = ctx.remap_dev(self.d_packed[seq0:], n, self.d_text)
        if self.n != n or self.sigma != sigma:
            N = n + 1
            self.n, self.sigma = n, sigma
            self.sa = torch.empty(N, dtype=torch.int32, device=self.dev)
            self.bwt = torch.empty(N, dtype=torch.uint8, device=self.dev)
            self.c = torch.zeros(sigma, dtype=torch.int32, device=self.dev)
            self.o = torch.empty((N + 1) * sigma, dtype=torch.int32, device=self.dev) if self.tables else None
        ctx.sa_bwt_build_dev(self.d_text, n, sigma, self.sa, self.bwt)
        if self.tables:
            ctx.bwt_tables_from_bwt_dev(self.bwt, n + 1, sigma, self.c, self.o)
        return n + 1
