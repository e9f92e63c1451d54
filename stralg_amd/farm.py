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


# The group that carries the timing barrier and the scalar reductions, and the device its tensors live on: RCCL
# ("nccl") with tensors on the rank's GPU, or gloo with CPU tensors (init_collectives decides, every rank alike).
_COLL = {"group": None, "device": None, "backend": None}


def init_collectives(rank, world, dev=None, prefer="nccl", attempt_timeout_s=180):
    """Process groups of an N > 1 run.  The default group is gloo (CPU, TCP on 127.0.0.1 / MASTER_ADDR): it always comes
    up and is the control plane.  With prefer == "nccl" and a GPU, an RCCL group is created beside it and tried once (an
    all-reduce of ones that must sum to `world`); whether EVERY rank's attempt succeeded is agreed over gloo, and only
    then do the barrier and the reductions of the timed region go over RCCL -- otherwise all ranks stay on gloo and the
    reason is reported.  Nothing on the data path is collective either way.  Returns a dict for the bench line:
    collective_backend, nccl_error, n_ranks_seen.

    Whether to attempt RCCL at all is agreed over gloo first (new_group is itself collective: ranks that disagree about
    having a GPU would wait for each other for ever).  An attempt that RAISES falls back to gloo; an attempt that HANGS
    (RCCL's usual failure mode on a broken fabric) is ended by torch's watchdog after attempt_timeout_s, which aborts the
    process: the run then dies with the watchdog's message rather than going on over gloo -- STRALG_BENCH_COLLECTIVES=gloo
    (bench.py: prefer="gloo") is the way round a fabric that hangs."""
    import datetime
    import os
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=1800))
    info = {"collective_backend": "gloo", "nccl_error": None}
    _COLL.update(group=None, device=None, backend="gloo")
    forced = os.environ.get("STRALG_BENCH_FORCE_NCCL_ATTEMPT") == "1"  # (tests: the fall-back branch on a box without GPUs)
    mine = prefer == "nccl" and ((dev is not None and dev.type == "cuda") or forced)
    everyone = torch.tensor([1 if mine else 0], dtype=torch.int32)
    dist.all_reduce(everyone, op=dist.ReduceOp.MIN)  # (gloo)
    if mine and int(everyone.item()) != 1:
        info["nccl_error"] = "not attempted: another rank has no GPU device"
    elif mine:
        ok, err, g = 1, None, None
        try:
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=attempt_timeout_s))
            t = torch.ones(1, dtype=torch.float64, device=dev)
            dist.all_reduce(t, group=g)
            if dev is not None and dev.type == "cuda":
                torch.cuda.synchronize(dev)
            if int(round(float(t.item()))) != world:
                raise RuntimeError(f"RCCL all-reduce of ones gave {float(t.item())}, expected {world}")
        except Exception as e:  # noqa: BLE001 -- whatever RCCL or the runtime raises here: the run goes on over gloo
            ok, err = 0, f"{type(e).__name__}: {e}".replace("\n", " ")[:400]
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # (default group: gloo)
        if int(flag.item()) == 1:
            _COLL.update(group=g, device=dev, backend="nccl")
            info["collective_backend"] = "nccl"
        else:
            info["nccl_error"] = err or "the RCCL attempt failed on another rank"
            if g is not None:  # its watchdog and proxy threads would go on polling beside the timed steps
                try:
                    dist.destroy_process_group(g)
                except Exception:  # noqa: BLE001
                    pass
    elif prefer == "nccl":
        info["nccl_error"] = "not attempted: no GPU device on this rank"
    ones = torch.ones(1, dtype=torch.float64, device=_COLL["device"])
    dist.all_reduce(ones, group=_COLL["group"])
    info["n_ranks_seen"] = int(round(float(ones.item())))
    return info


def collective_backend():
    return _COLL["backend"]


def gather_objects(obj):
    """[rank 0's object, rank 1's, ...] over the gloo control plane (bookkeeping only)"""
    dist = _dist()
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def _barrier():
    dist = _dist()
    if dist is None:
        return
    import torch
    t = torch.zeros(1, dtype=torch.int32, device=_COLL["device"])
    dist.all_reduce(t, group=_COLL["group"])
    if _COLL["device"] is not None and _COLL["device"].type == "cuda":
        torch.cuda.synchronize(_COLL["device"])


def fence(cuda=True):
    """barrier + device sync on both sides, as bench.py's timing contract requires."""
    if cuda:
        import torch
        torch.cuda.synchronize()
    _barrier()
    if cuda:
        import torch
        torch.cuda.synchronize()


def reduce_scalars(elapsed, units, device=None):
    """(max elapsed over ranks, sum of units over ranks), over the group init_collectives chose (`device` is kept for
    callers that set up torch.distributed themselves: the tests' plain gloo groups)."""
    dist = _dist()
    if dist is None:
        return elapsed, units
    import torch
    if _COLL["backend"] is not None:
        device = _COLL["device"]
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_COLL["group"])
    dist.all_reduce(u, op=dist.ReduceOp.SUM, group=_COLL["group"])
    return float(t.item()), int(round(u.item()))


def gather_ints(value):
    """[value of rank 0, value of rank 1, ...] (bookkeeping only)"""
    return [int(v) for v in gather_objects(int(value))]


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
        include_reverse (what bwt_readmapper.c:57 asks build_complete_table for, bwt.c:147-158):
        -> sx_reverse_dev -> sx_sa_bwt_build_dev on the reversed string -> sx_bwt_tables_from_bwt_dev = the RO table
           (the reverse suffix array and C table are temporaries, as in the reference)

    `dev` is a torch device: cuda:k with the product library, cpu with the CPU execution harness
    of the kernels (tests).  The buffers are allocated once and reused by every call."""

    def __init__(self, ctx, image_host, dev, tables=True, include_reverse=False):
        import torch
        self.ctx, self.dev, self.tables = ctx, dev, tables
        self.include_reverse = bool(include_reverse and tables)
        self.h_file = image_host  # torch uint8 on the CPU (pinned when a GPU is used)
        flen = int(image_host.numel())
        self.file_len = flen
        self.d_file = torch.empty(flen, dtype=torch.uint8, device=dev)
        self.d_packed = torch.empty(flen + 1, dtype=torch.uint8, device=dev)
        self.d_term = torch.zeros(16, dtype=torch.int32, device=dev)
        self.d_text = torch.empty(flen + 1, dtype=torch.uint8, device=dev)
        self.n = self.sigma = None
        self.sa = self.bwt = self.c = self.o = None
        self.d_rev = self.rsa = self.rbwt = self.rc = self.ro = None

    def upload(self):
        """the H2D copy of the file image (what load_fasta_records' fread is to the reference)"""
        self.d_file.copy_(self.h_file)
        if self.dev.type == "cuda":
            import torch
            torch.cuda.synchronize(self.dev)

    def set_include_reverse(self, on):
        self.include_reverse = bool(on and self.tables)
        if not self.include_reverse:
            self.d_rev = self.rsa = self.rbwt = self.rc = self.ro = None

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
            self.d_rev = None
        ctx.sa_bwt_build_dev(self.d_text, n, sigma, self.sa, self.bwt)
        if self.tables:
            ctx.bwt_tables_from_bwt_dev(self.bwt, n + 1, sigma, self.c, self.o)
        if self.include_reverse:
            N = n + 1
            if self.d_rev is None:
                self.d_rev = torch.empty(N, dtype=torch.uint8, device=self.dev)
                self.rsa = torch.empty(N, dtype=torch.int32, device=self.dev)
                self.rbwt = torch.empty(N, dtype=torch.uint8, device=self.dev)
                self.rc = torch.zeros(sigma, dtype=torch.int32, device=self.dev)
                self.ro = torch.empty((N + 1) * sigma, dtype=torch.int32, device=self.dev)
            ctx.reverse_dev(self.d_text, n, self.d_rev)
            ctx.sa_bwt_build_dev(self.d_rev, n, sigma, self.rsa, self.rbwt)
            ctx.bwt_tables_from_bwt_dev(self.rbwt, N, sigma, self.rc, self.ro)
        return n + 1
