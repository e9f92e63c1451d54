#!/usr/bin/env python3
"""bench.py -- Msuffixes/s of suffix-array + BWT C/O-table construction on MI355X.

A "step" is one pass of the hot path over one record: sa_is_construction's device
path (sx_sa_build_dev) followed by the C/O-table build (sx_bwt_tables_dev) on a
synthetic DNA record (sigma = 4 letters + sentinel) already resident in HBM.
Workload at N=1: BASELINE.json configs[2], "SA-IS + BWT C/O-table build on 1 GiB
random DNA".  With --gpus N every rank builds its own independent record (the
per-record loop of bwt_readmapper.c:54-62 farmed one per GPU): weak scaling, no
collective on the data path; torch.distributed is used only for the timing
barrier and the max-over-ranks reduction.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md, chip-level parameters)


def cpu_baseline(log2_sample, sigma, seed):
    """Oracle (single thread) on a bounded sample of the same stream: SA + C/O tables."""
    import numpy as np
    import oracle
    from stralg_amd.synth import synth
    n = 1 << log2_sample
    x = synth(n, sigma, seed)
    t0 = time.perf_counter()
    sa = oracle.sa_is(x, sigma)
    t1 = time.perf_counter()
    levels = oracle.last_levels()
    # C/O tables on a smaller sample: the O table takes (n+2) * sigma * 4 bytes of host memory
    o_n = min(n, 1 << 22)
    xs = x[:o_n]
    sas = sa if o_n == n else oracle.sa_is(xs, sigma)
    t2 = time.perf_counter()
    oracle.c_table(xs, sigma)
    oracle.o_table(xs, sas, sigma)
    t3 = time.perf_counter()
    port = (n + 1) / (t1 - t0) / 1e6
    out = {
        "value": round(port, 3),
        "unit": "Msuffixes/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle_sa_is on the first 2^{log2_sample} symbols of the same stream ({t1 - t0:.1f} s); "
                  f"C/O tables on 2^{o_n.bit_length() - 1} symbols ({t3 - t2:.2f} s = "
                  f"{(o_n + 1) / (t3 - t2) / 1e6:.1f} Mpositions/s)",
        "host_cpus": os.cpu_count(),
        "levels": levels[:6],
    }
    from oracle import pyoracle
    if pyoracle.have_ref():
        # the unmodified reference (oracle/_ref, built from the reference's sources by oracle/Makefile): its own
        # sa_is_construction on the same sample is the baseline proper; the port's rate stays in the text
        ref = pyoracle._Ref()
        t4 = time.perf_counter()
        sa_ref = ref.sa_is(x, sigma)
        t5 = time.perf_counter()
        if not (sa_ref == sa).all():
            raise RuntimeError("reference and oracle disagree on the baseline sample")
        out["value"] = round((n + 1) / (t5 - t4) / 1e6, 3)
        out["kind"] = "reference"
        out["sample"] = (f"the reference's sa_is_construction on the first 2^{log2_sample} symbols of the same stream "
                         f"({t5 - t4:.1f} s; the oracle port: {port:.1f} Msuffixes/s); " + out["sample"].split("; ", 1)[1])
    return out


def pmc_traffic(log2n, sigma, tables, klass):
    """HBM bytes per launch of the dominant kernel class from the committed rocprofv3 PMC passes of this
    same command (profiles/pmc_traffic.json, made by tools/profile.sh + tools/pmc_to_json.py); PMC
    counters cannot be read from inside the process, so this is null for workloads not profiled."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        c = doc[f"log2n={log2n} sigma={sigma} tables={int(tables)}"]["classes"][klass]
        return round(c["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2n", type=int, default=int(os.environ.get("STRALG_BENCH_LOG2N", "30")))
    ap.add_argument("--sigma", type=int, default=5)
    ap.add_argument("--no-tables", action="store_true", help="suffix array only")
    ap.add_argument("--cpu-log2n", type=int, default=int(os.environ.get("STRALG_BENCH_CPU_LOG2N", "25")))
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import stralg_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; STRALG_BENCH_BACKEND=gloo + STRALG_BENCH_SHARE_GPU=1 let the N > 1 path be
    # exercised on a single-GPU box (tests): every rank then uses cuda:0 and the scalars travel on the CPU
    backend = os.environ.get("STRALG_BENCH_BACKEND", "nccl")
    if os.environ.get("STRALG_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = stralg_amd.Context(local_rank)

    n = 1 << args.log2n
    N = n + 1
    sigma = args.sigma
    tables = (not args.no_tables) and sigma <= 128
    seed = 42 + rank
    text = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.synth_dev(text, n, sigma, seed)
    sa = torch.empty(N, dtype=torch.int32, device=dev)
    c_tab = torch.zeros(sigma, dtype=torch.int32, device=dev)
    o_tab = torch.empty((N + 1) * sigma, dtype=torch.int32, device=dev) if tables else None
    bwt = torch.empty(N, dtype=torch.uint8, device=dev) if tables else None

    def step():
        # build_complete_table's device work (stralg/bwt.c:143,154): suffix array, then C and O.
        # The induced-sort passes hand the BWT over with the suffix array (sx_sa_bwt_build_dev).
        if tables:
            ctx.sa_bwt_build_dev(text, n, sigma, sa, bwt)
            ctx.bwt_tables_from_bwt_dev(bwt, N, sigma, c_tab, o_tab)
        else:
            ctx.sa_build_dev(text, n, sigma, sa)

    from stralg_amd import farm

    def profiled_step():
        """one untimed step with HIP events around every launch: the per-class table, and which class dominates"""
        ctx.profile_reset()
        ctx.profile_only(None)
        ctx.profile_enable(True)
        step()
        torch.cuda.synchronize()
        ctx.profile_enable(False)
        return ctx.profile_read()

    for _ in range(args.warmup):
        step()
    # one more untimed step, with events around every launch: the per-class table and the dominant class
    table = profiled_step()
    dom = max(table, key=lambda k: table[k]["ms"])
    # Timed region: events only around the dominant kernel's launches (the roofline figure is measured live, on
    # the library's own stream); two event records around each of a step's ~300 launches would cost ~5 % of it.
    ctx.profile_reset()
    ctx.profile_only(dom)
    ctx.profile_enable(True)
    # barrier + torch.cuda.synchronize() on both sides of exactly `steps` steps
    elapsed = farm.timed(step, args.steps, 0, cuda=True)
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    ctx.profile_only(None)
    stats = ctx.last_stats()
    # max time over ranks, total suffixes over ranks (the only collectives; none on the data path)
    elapsed, total_units = farm.reduce_scalars(elapsed, args.steps * N, device=dev if backend == "nccl" else None)

    if rank == 0:
        value = total_units / elapsed / 1e6
        # dominant kernel class (by summed HIP-event time of the profiled step), measured in the timed steps
        d = prof[dom]
        achieved = d["alg_bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        size_label = f"{n >> 30} GiB" if n >= (1 << 30) and n % (1 << 30) == 0 else (f"{n >> 20} MiB" if n >= (1 << 20) else f"{n} B")
        alpha_label = "DNA" if sigma == 5 else f"sigma={sigma}"
        out = {
            "metric": f"Msuffixes/s ({'SA-IS + BWT C/O tables' if tables else 'SA-IS'}, {size_label} {alpha_label})",
            "value": round(value, 3),
            "unit": "Msuffixes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 text / u32 indices",
            "data": "synthetic",
            "config": {
                "workload": f"sa_is_construction + init_bwt_table (C, O) on 2^{args.log2n} random symbols, "
                            f"sigma={sigma - 1}+sentinel, one independent record per GPU"
                            if tables else
                            f"sa_is_construction on 2^{args.log2n} random symbols, alphabet_size={sigma}",
                "n": n, "alphabet_size": sigma, "records_per_gpu": 1, "parallelism": f"batch x{world}, no collectives",
                "seed": 42,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(args.log2n, sigma, tables, dom),
                "launches": d["launches"],
                "avg_ms": round(d["ms"] / max(1, d["launches"]), 4),
            },
            # per-class times of ONE untimed step with events around every launch (run between warm-up and timing)
            "kernels": {k: {"ms_per_step": round(v["ms"], 3), "launches_per_step": v["launches"],
                            "GBps": round(v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0}
                        for k, v in table.items() if v["launches"]},
            "build_stats": stats,
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_log2n, sigma, 42)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
