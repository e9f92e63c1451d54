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


def test_c_lpt_assignment_matches_python():
    """stralg_amd_lpt_assign (the C farm's dealer) gives the assignment of farm.lpt_assign"""
    import ctypes as C
    from stralg_amd import _lib
    from stralg_amd.farm import lpt_assign
    lib = _lib.load()
    lib.stralg_amd_lpt_assign.argtypes = [C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.POINTER(C.c_int)]
    lib.stralg_amd_lpt_assign.restype = C.c_int
    rng = np.random.default_rng(11)
    for count, lanes in ((0, 3), (1, 1), (1, 4), (8, 8), (9, 2), (40, 3), (100, 8)):
        lengths = [int(v) for v in rng.integers(1, 1000, size=count)]
        if count > 4:
            lengths[3] = lengths[1]  # ties keep the given order
        arr = (C.c_size_t * max(1, count))(*lengths)
        out = (C.c_int * max(1, count))()
        assert lib.stralg_amd_lpt_assign(arr, count, lanes, out) == 0
        want = lpt_assign(lengths, lanes)
        got = [[k for k in range(count) if out[k] == lane] for lane in range(lanes)]
        assert [sorted(g) for g in got] == [sorted(w) for w in want], (count, lanes)
    assert lib.stralg_amd_lpt_assign(None, 0, 0, None) == -1


def test_c_farm_workers_per_device(monkeypatch):
    """stralg_amd_farm_workers_per_device: short records get up to four workers a device (never more than records a
    device), long ones a single worker; $STRALG_AMD_FARM_WORKERS overrides"""
    import ctypes as C
    from stralg_amd import _lib
    lib = _lib.load()
    fn = lib.stralg_amd_farm_workers_per_device
    fn.argtypes = [C.POINTER(C.c_size_t), C.c_size_t, C.c_int]
    fn.restype = C.c_int

    def workers(lengths, devices):
        arr = (C.c_size_t * max(1, len(lengths)))(*lengths)
        return fn(arr, len(lengths), devices)

    monkeypatch.delenv("STRALG_AMD_FARM_WORKERS", raising=False)
    assert workers([1000] * 64, 1) == 4 and workers([1 << 24] * 64, 8) == 4
    assert workers([(1 << 24) + 1] * 64, 1) == 2 and workers([1 << 26] * 64, 1) == 2
    assert workers([1 << 30] * 64, 8) == 1 and workers([100, 1 << 27], 1) == 1
    assert workers([1000] * 3, 1) == 3 and workers([1000] * 8, 8) == 1 and workers([1000] * 9, 8) == 2
    assert workers([], 4) == 1 and workers([5], 0) == 1
    monkeypatch.setenv("STRALG_AMD_FARM_WORKERS", "8")
    assert workers([1 << 30] * 64, 1) == 8 and workers([1000] * 3, 1) == 3
    monkeypatch.setenv("STRALG_AMD_FARM_WORKERS", "0")
    assert workers([1000] * 64, 1) == 1


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


def test_bench_launches_its_own_ranks_on_fasta_records(emu_ctx):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent starts two fresh ranks, each
    builds its own FASTA record (BASELINE.json configs[4]: image -> pack -> remap -> suffix array + BWT -> C/O
    tables, bwt_readmapper.c:54-62) over the CPU execution harness, the results are verified, and rank 0's line
    carries n_gpus = 2 with the kernel-only and the ingest-inclusive rate."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["STRALG_BENCH_EMU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "5003", "--steps", "1",
                          "--warmup", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["scaling"] == "weak" and doc["unit"] == "Msuffixes/s"
    assert doc["collective_backend"] == "gloo" and doc["n_ranks_seen"] == 2 and doc["collectives"]["nccl_error"] is None
    assert doc["verified"] is True and len(doc["verified_checks"]) == 4
    assert doc["config"]["n"] == 5003 and "FASTA" in doc["config"]["workload"]
    fr = doc["fasta_record"]
    assert fr["kernel_only_Msuffixes_per_s"] > 0 and fr["ingest_inclusive_Msuffixes_per_s"] > 0
    assert doc["build_stats"]["n"] == 5003
    # the egress leg: every rank's index leaves its device at once (streamed index file, then malloc'd host tables)
    assert fr["numa_node_per_rank"] == [-1, -1] or len(fr["numa_node_per_rank"]) == 2
    assert fr["index_bytes_per_record"] == 4 * 5004 + 4 * 5 + 4 * 5 * 5005
    for key in ("stream_ms_per_record", "egress_inclusive_Msuffixes_per_s", "d2h_GBps_per_rank", "d2h_GBps_all_ranks",
                "host_tables_ms_per_record", "host_tables_Msuffixes_per_s"):
        assert key in fr and fr[key] >= 0, key
    assert fr["stream_ms_per_record"] > 0 and fr["egress_inclusive_Msuffixes_per_s"] > 0
    # what bwt_readmapper.c:57 asks for -- build_complete_table(seq, true): the reverse direction on the device too, verified
    ro = fr["with_ro"]
    assert ro["verified"] is True and ro["kernel_only_ms_per_record"] > 0 and ro["ingest_inclusive_Msuffixes_per_s"] > 0
    assert fr["egress_with_ro"]["index_bytes_per_record"] == fr["index_bytes_per_record"] + 4 * 5 * 5005
    assert fr["egress_with_ro"]["stream_ms_per_record"] > 0


def test_bench_falls_back_to_gloo_when_rccl_does_not_come_up(emu_ctx):
    """farm.init_collectives: gloo is the control plane; an RCCL group is tried beside it and used only when the attempt
    succeeds on every rank.  Here (no GPU: the attempt is forced and must fail) both ranks agree on gloo, the run goes
    on, and the line says which backend carried the barrier, why, how many ranks it saw and who they were."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(STRALG_BENCH_EMU="1", STRALG_BENCH_BACKEND="nccl", STRALG_BENCH_FORCE_NCCL_ATTEMPT="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "1500", "--steps", "1",
                          "--warmup", "0", "--no-egress"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    doc = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert doc["n_gpus"] == 2 and doc["verified"] is True
    assert doc["collective_backend"] == "gloo" and doc["n_ranks_seen"] == 2
    co = doc["collectives"]
    assert co["backend"] == "gloo" and co["nccl_error"] and co["control_plane"] == "gloo"
    assert [r["rank"] for r in co["ranks"]] == [0, 1]


def test_bench_under_torch_distributed_run(emu_ctx):
    """the driver's own launch form for N > 1 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W: RANK / LOCAL_RANK / WORLD_SIZE come from the environment,
    rank 0 prints the one line (here over the CPU execution harness, so the collectives stay on gloo)"""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(STRALG_BENCH_EMU="1", STRALG_BENCH_LOG2N="12")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["steps"] == 2 and doc["warmup"] == 1 and doc["scaling"] == "weak"
    assert doc["verified"] is True and doc["n_ranks_seen"] == 2
    assert [r["local_rank"] for r in doc["collectives"]["ranks"]] == [0, 1]
    assert doc["config"]["n"] == 4096 and "FASTA" in doc["config"]["workload"]


def test_bench_default_run_carries_the_other_configs(emu_ctx):
    """the default (one GPU, DNA) line also measures BASELINE.json's other single-GPU configurations after the timed
    region, each verified, and names the algorithm that ran (here at 2^12 symbols over the CPU execution harness)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["STRALG_BENCH_EMU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log2n", "12", "--steps", "1", "--warmup", "0",
                          "--no-e2e", "--no-cpu", "--other-steps", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    doc = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    # (a record this short is sorted directly -- SX_FLAG_SMALL_DIRECT_MAX -- and the line says so; at BASELINE's sizes it reads "SA-IS + ...")
    assert doc["verified"] is True and doc["metric"].startswith("Msuffixes/s (direct prefix sort + BWT C/O tables") and doc["build_stats"]["lms_path"] == 3
    oc = doc["other_configs"]
    assert set(oc) == {"dna_1024B", "bytes_4096B", "bytes_4096B_induced", "genome_like_4096B", "fibonacci_4096B"}
    for name, c in oc.items():
        assert c.get("verified") is True, (name, c)
        assert c["ms_per_step"] > 0 and "roofline_frac" in c and "lms_path" in c
    assert oc["bytes_4096B"]["lms_path"] == 3 and "direct prefix sort" in oc["bytes_4096B"]["algorithm"]
    assert oc["bytes_4096B_induced"]["lms_path"] in (1, 2) and oc["bytes_4096B_induced"]["induce_rounds"] > 100


def test_bench_line_carries_round5_legs(emu_ctx):
    """round 5's additions to the line, over the CPU execution harness at 2^12 symbols: the measured-ceiling keys beside the
    roofline, the row at the reference's published size through the host C API (arrays compared with the CPU baseline's), the
    host-buffer leg, and -- FASTA line -- SURVEY 8d's CPU row for configs[4] (one CPU process per record)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["STRALG_BENCH_EMU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log2n", "12", "--steps", "6", "--warmup", "0",
                          "--e2e-log2n", "12", "--cpu-log2n", "12", "--no-other-configs"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    doc = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert "peak_measured" in doc["roofline"] and doc["roofline"]["peak"] == 8000.0
    # the dominant class carries its events in every fourth step of the timed region (steps 0 and 4 of six), and only there
    rf = doc["roofline"]
    assert (rf["timed_steps"], rf["of_steps"]) == (2, 6) and rf["launches"] > 0 and rf["launches"] % 2 == 0, rf  # (the harness's events carry no times)
    assert rf["launches"] // 2 == doc["kernels"][rf["kernel"]]["launches_per_step"], (rf, doc["kernels"][rf["kernel"]])
    rs = doc["cpu_baseline"]["reference_scale"]
    assert [r["n"] for r in rs["rows"]] == [49000, 65536] and all(r["arrays_identical"] for r in rs["rows"]), rs
    assert rs["published"]["ms"] == 4.63 and all(r["gpu_call_ms"] > 0 and r["cpu_ms"] > 0 for r in rs["rows"])
    e2e = doc["end_to_end"]["2^12"]
    assert e2e["build_complete_table_ms"] > 0 and e2e["with_ro_ms"] > 0 and "readmapper_loop" in e2e
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "fasta", "--n", "5003", "--steps", "1", "--warmup", "0",
                          "--cpu-log2n", "12", "--no-e2e", "--no-egress"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    doc = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    row = doc["fasta_record"]["cpu_cfg5"]
    assert row["records"] == 8 and row["aggregate_Msuffixes_per_s"] > 0 and len(row["seconds_per_process"]) == 8, row


def test_bench_reports_a_failed_verification(emu_ctx, tmp_path):
    """the verifier is not a rubber stamp: a wrong suffix array is refused"""
    import torch
    from stralg_amd import verify
    import oracle
    from stralg_amd.synth import synth
    x = synth(3000, 5, 3)
    sa = oracle.sa_is(x, 5).astype(np.int64)
    t = torch.from_numpy(x)
    good = torch.from_numpy(sa.astype(np.int32))
    assert verify.verify_sa_on_device(t, good, 3000)
    bad = good.clone()
    bad[[10, 11]] = bad[[11, 10]]
    with pytest.raises(AssertionError):
        verify.verify_sa_on_device(t, bad, 3000)
    dup = good.clone()
    dup[5] = dup[6]
    with pytest.raises(AssertionError):
        verify.verify_sa_on_device(t, dup, 3000)
    bw = torch.from_numpy(oracle.bwt(x, sa.astype(np.uint32)))
    c = torch.from_numpy(oracle.c_table(x, 5).astype(np.int32))
    o = torch.from_numpy(oracle.o_table(x, sa.astype(np.uint32), 5).astype(np.int32).reshape(-1))
    assert len(verify.verify_build_on_device(t, 3000, 5, good, bw, c, o)) == 3
    o2 = o.clone()
    o2[5 * 1000 + 2] += 1
    with pytest.raises(AssertionError):
        verify.verify_build_on_device(t, 3000, 5, good, bw, c, o2)
