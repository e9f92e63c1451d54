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
