"""The N > 1 path on CPU: two gloo ranks farm independent records (LPT by length),
each rank builds its own records (here on the CPU execution harness of the kernels),
no data-path collective; only the timing barrier and the scalar reductions are
collective.  Mirrors what bench.py --gpus N does on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_assignment():
    from stralg_amd.farm import lpt_assign
    assert lpt_assign([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]
    a = lpt_assign([10, 1, 1, 1, 7, 3], 2)
    assert sorted(sum(a, [])) == list(range(6))
    loads = [sum([10, 1, 1, 1, 7, 3][i] for i in r) for r in a]
    assert max(loads) - min(loads) <= 3
    assert lpt_assign([], 3) == [[], [], []]
    assert lpt_assign([4], 3) == [[0], [], []]


def _worker(rank, world, port, emu_lib, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle
    from stralg_amd.api import Context
    from stralg_amd.farm import lpt_assign, reduce_scalars, timed
    from stralg_amd.synth import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = [3000, 500, 2500, 1200, 800]
    mine = lpt_assign(lengths, world)[rank]
    ctx = Context(0, lib_path=emu_lib)
    results = {}

    def step():
        for i in mine:
            x = synth(lengths[i], 5, 100 + i)
            results[i] = (x, ctx.sa_build(x, 5))

    elapsed = timed(step, steps=1, warmup=0, cuda=False)
    for i, (x, sa) in results.items():
        assert (sa == oracle.sa_is(x, 5)).all()
    units = sum(lengths[i] + 1 for i in mine)
    t, total = reduce_scalars(elapsed, units)
    assert total == sum(lengths) + len(lengths)
    assert t >= elapsed - 1e-9
    with open(os.path.join(out_dir, f"rank{rank}.ok"), "w") as f:
        f.write(",".join(map(str, mine)))
    dist.destroy_process_group()
    ctx.close()


def test_two_rank_farm(emu_ctx, tmp_path):
    import torch.multiprocessing as mp
    from conftest import EMU_LIB
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, EMU_LIB, str(tmp_path)), nprocs=2, join=True)
    got = sorted(sum((open(tmp_path / f"rank{r}.ok").read().split(",") for r in range(2)), []))
    assert got == ["0", "1", "2", "3", "4"]
